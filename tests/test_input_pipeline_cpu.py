"""CPU: the restatement of the loader transform (oracle/input_ref.c) and the host half of the device pipeline.

cv2 is not installable here, so cv2.resize itself cannot be run: the oracle restates OpenCV's published 8-bit
algorithm (oracle/input_ref.c header) -- parity UNPINNED against cv2, pinned here against independent properties:
  * f = 1 is the identity, crop / pad / mirror bookkeeping equals plain numpy slicing;
  * the fixed-point INTER_LINEAR result is within 1 grey level of float bilinear interpolation with the same
    half-pixel geometry (torch, align_corners=False, scale_factor given) wherever both are defined;
  * INTER_NEAREST equals floor(d / f) indexing;
  * the random draws are taken in the reference's order from the reference's generators (datasets.py:158,198-199,206).
"""
import ctypes
import random

import numpy as np
import pytest
import torch

from oracle import cref
from structure_knowledge_distillation_amd import _lib
from structure_knowledge_distillation_amd.dataset import datasets as D

MEAN = (104.00698793, 116.66876762, 122.67891434)


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def run(lib, img, lab, params, ch, cw, channels_last=0, to=lambda t: t):
    B, H0, W0 = img.shape[:3]
    p = np.asarray(params, dtype=np.float64).reshape(B, 6)
    f = to(torch.from_numpy(np.ascontiguousarray(p[:, 0])))
    ints = [to(torch.from_numpy(np.ascontiguousarray(p[:, k].astype(np.int32)))) for k in range(1, 6)]
    out = to(torch.empty(B, ch, cw, 3) if channels_last else torch.empty(B, 3, ch, cw))
    ol = to(torch.empty(B, ch, cw, dtype=torch.int64)) if lab is not None else None
    lut = to(torch.from_numpy(D.trainid_lut()))
    mean = (ctypes.c_float * 3)(*MEAN)
    keep = (to(img), to(lab) if lab is not None else None)
    assert lib.skd_cs_transform(B, H0, W0, P(keep[0]), P(keep[1]), P(lut), P(f), *[P(t) for t in ints], ch, cw, mean, 255,
                                P(out), channels_last, P(ol), None)
    return out, ol


@pytest.fixture(scope="module")
def ref():
    return cref.load(_lib.SIGNATURES)


def test_lut_matches_the_reference_table():
    lut = D.trainid_lut()
    assert [int(lut[k]) for k in (7, 8, 11, 26, 33)] == [0, 1, 2, 13, 18] and int(lut[0]) == 255 and int(lut[29]) == 255
    assert int(lut[200]) == 200 and int(lut[255]) == 255          # ids outside the table are kept (datasets.py:162-171)
    assert sum(1 for v in D.ID_TO_TRAINID.values() if v != 255) == 19


def test_identity_crop_pad_mirror(ref):
    g = torch.Generator().manual_seed(0)
    B, H0, W0, ch, cw = 3, 37, 53, 48, 40            # crop_h > H0: bottom padding
    img = torch.randint(0, 256, (B, H0, W0, 3), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, 34, (B, H0, W0), generator=g, dtype=torch.uint8)
    params = [(1.0, H0, W0, 0, 5, 1), (1.0, H0, W0, 0, 13, -1), (1.0, H0, W0, 0, 0, -1)]
    out, ol = run(ref, img, lab, params, ch, cw)
    m = np.asarray(MEAN, dtype=np.float32)
    lut = D.trainid_lut()
    for b, (_, _, _, ho, wo, fl) in enumerate(params):
        im = np.zeros((max(H0, ch), W0, 3), np.float32)
        im[:H0] = img[b].numpy().astype(np.float32) - m
        la = np.full((max(H0, ch), W0), 255, np.uint8)
        la[:H0] = lut[lab[b].numpy()]
        im, la = im[ho:ho + ch, wo:wo + cw].transpose(2, 0, 1), la[ho:ho + ch, wo:wo + cw]
        if fl < 0:
            im, la = im[:, :, ::-1], la[:, ::-1]
        assert np.array_equal(out[b].numpy(), im) and np.array_equal(ol[b].numpy(), la.astype(np.int64))
    out_cl, _ = run(ref, img, lab, params, ch, cw, channels_last=1)
    assert torch.equal(out_cl.permute(0, 3, 1, 2), out)


@pytest.mark.parametrize("f", [0.7, 0.8, 1.3, 2.1])
def test_fixed_point_resize_tracks_float_bilinear(ref, f):
    g = torch.Generator().manual_seed(int(f * 10))
    H0, W0 = 41, 67
    img = torch.randint(0, 256, (1, H0, W0, 3), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, 34, (1, H0, W0), generator=g, dtype=torch.uint8)
    dh, dw = int(round(H0 * f)), int(round(W0 * f))
    out, ol = run(ref, img, lab, [(f, dh, dw, 0, 0, 1)], dh, dw)
    got = out[0] + torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)          # back to grey levels
    want = torch.nn.functional.interpolate(img.permute(0, 3, 1, 2).float(), scale_factor=f, mode="bilinear", align_corners=False,
                                           recompute_scale_factor=False)[0]
    h, w = min(dh, want.shape[1]), min(dw, want.shape[2])
    assert float((got[:, :h, :w] - want[:, :h, :w]).abs().max()) <= 1.0 + 1e-3
    assert float((got - got.round()).abs().max()) < 1e-4                            # integer grey levels minus a float32 mean
    ys = np.minimum(np.floor(np.arange(dh) * (1.0 / f)).astype(int), H0 - 1)
    xs = np.minimum(np.floor(np.arange(dw) * (1.0 / f)).astype(int), W0 - 1)
    assert np.array_equal(ol[0].numpy(), D.trainid_lut()[lab[0].numpy()][ys][:, xs].astype(np.int64))


def test_random_draws_follow_the_reference_order():
    random.seed(5)
    np.random.seed(6)
    got = D.draw_sample_params(1024, 2048, 512, 512)
    random.seed(5)
    np.random.seed(6)
    f = 0.7 + random.randint(0, 14) / 10.0                                    # datasets.py:158
    dh, dw = int(round(1024 * f)), int(round(2048 * f))
    h_off = random.randint(0, max(dh, 512) - 512)                             # :198
    w_off = random.randint(0, max(dw, 512) - 512)                             # :199
    flip = np.random.choice(2) * 2 - 1                                        # :206
    assert got == (f, dh, dw, h_off, w_off, int(flip))
    assert D.draw_sample_params(300, 400, 512, 512, scale=False, mirror=False) == (1.0, 300, 400, 0, 0, 1)


def test_collate_runs_in_dataloader_workers(ref, tmp_path, monkeypatch):
    """``DataLoader(ds, collate_fn=ds.collate, num_workers>0)`` -- the documented recipe -- runs
    collate_fn in forked worker processes, where nothing may touch the GPU.  collate is host-only now (stack + random
    draws -> RawBatch); the device transform runs in the consuming process (NetModel.set_input / RawBatch.to_device).
    cv2 is absent here: a stub ``imread`` decodes seeded arrays (the workers inherit it through fork)."""
    import sys
    import types
    H0, W0 = 64, 96
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_COLOR, cv2.IMREAD_GRAYSCALE = 1, 0

    def imread(path, flag):
        g = np.random.RandomState(abs(hash(path)) % (2 ** 31))
        return g.randint(0, 256, (H0, W0, 3)).astype(np.uint8) if flag == 1 else g.randint(0, 34, (H0, W0)).astype(np.uint8)

    cv2.imread = imread
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    lst = tmp_path / "train.lst"
    lst.write_text("".join("img/%d.png lab/%d.png\n" % (i, i) for i in range(6)))
    ds = D.CSDataSet(str(tmp_path), str(lst), crop_size=(48, 40), mean=MEAN, device="cpu")
    loader = torch.utils.data.DataLoader(ds, batch_size=3, collate_fn=ds.collate, num_workers=2, shuffle=False)
    _lib.install_test_backend(ref)
    try:
        seen = 0
        for data in loader:
            raw, none, size, names = data
            assert isinstance(raw, D.RawBatch) and none is None and len(names) == 3 and tuple(size.shape) == (3, 3)
            assert raw.images.dtype == torch.uint8 and tuple(raw.images.shape) == (3, H0, W0, 3) and not raw.images.is_cuda
            assert len(raw.params) == 3 and all(len(p) == 6 for p in raw.params)
            img, lab = raw.to_device("cpu")
            want_img, want_lab = D.CSTrainTransform((48, 40), MEAN, device="cpu")(raw.images, raw.labels, raw.params)
            assert torch.equal(img, want_img) and torch.equal(lab, want_lab)
            assert tuple(img.shape) == (3, 3, 48, 40) and lab.dtype == torch.int64
            seen += 1
        assert seen == 2
        # NetModel.set_input's branch: anything with to_device + params is transformed in the calling process
        from structure_knowledge_distillation_amd.networks.kd_model import NetModel
        holder = types.SimpleNamespace(args=types.SimpleNamespace(device=torch.device("cpu")))
        NetModel.set_input(holder, data)
        assert torch.equal(holder.images, img) and torch.equal(holder.labels, lab)
    finally:
        _lib.install_test_backend(None)
