"""Pins the plain-C restatement (oracle/libskd_ref.so) to the torch restatements that
tests/test_oracle_vs_reference.py pins to the reference's own Python: same closed forms, fp64."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from oracle import abn_torch, cref, step_torch as O
from structure_knowledge_distillation_amd import _lib


@pytest.fixture(scope="module")
def ref():
    return cref.load(_lib.SIGNATURES)


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("act", ["none", "leaky_relu", "elu"])
@pytest.mark.parametrize("shape", [(2, 3, 5), (4, 6, 37), (1, 2, 300)])
def test_abn_c_vs_autograd(ref, act, shape):
    N, C, S = shape
    code = {"none": 0, "leaky_relu": 1, "elu": 2}[act]
    g = torch.Generator().manual_seed(S)
    x = torch.randn(N, C, S, generator=g) * 2 + 1
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    w[0] = -w[0].abs()
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    xo = x.double().requires_grad_(True)
    wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
    rmo, rvo = rm.double(), rv.double()
    zo = abn_torch.abn_autograd(xo, wo, bo, rmo, rvo, True, 0.1, 1e-5, act, 0.01)
    dz = torch.randn(N, C, S, generator=g)
    zo.backward(dz.double())
    z, m, v = x.clone(), torch.empty(C), torch.empty(C)
    ws = torch.empty(max(1, ref.skd_abn_workspace_floats(N, C, S)))
    assert ref.skd_abn_forward_train(N, C, S, P(z), P(w), P(b), P(rm), P(rv), P(m), P(v), 0.1, 1e-5, code, 0.01, P(ws), None)
    assert rel(z, zo.detach()) < 1e-5 and rel(rm, rmo) < 1e-6 and rel(rv, rvo) < 1e-6
    dx, dw, db, e, ey = torch.empty_like(x), torch.zeros(C), torch.zeros(C), torch.empty(C), torch.empty(C)
    assert ref.skd_abn_backward(N, C, S, P(z), P(dz), P(v), P(w), P(b), P(e), P(ey), P(dx), P(dw), P(db), 1e-5, code, 0.01, 1, P(ws), None)
    assert rel(dx, xo.grad) < 2e-4 and rel(dw, wo.grad) < 2e-4 and rel(db, bo.grad) < 2e-4
    # and the hand-written backward formula restated in torch agrees with autograd
    fdx, fdw, fdb, _, _ = abn_torch.abn_backward_formula(zo.detach(), dz.double(), abn_torch.batch_stats(x.double())[1],
                                                         wo.detach(), bo.detach(), True, 1e-5, act, 0.01)
    assert rel(fdx, xo.grad) < 1e-9 and rel(fdw, wo.grad) < 1e-9 and rel(fdb, bo.grad) < 1e-9


def test_pixelwise_c_vs_torch(ref):
    g = torch.Generator().manual_seed(1)
    s, t = torch.randn(2, 19, 5, 7, generator=g) * 3, torch.randn(2, 19, 5, 7, generator=g) * 3
    so = s.double().requires_grad_(True)
    L = O.criterion_pixel_wise([so], [t.double()])
    L.backward()
    loss, grad = torch.empty(1), torch.empty_like(s)
    assert ref.skd_pixelwise_loss(2, 19, 35, P(s), P(t), P(loss), P(grad), P(torch.empty(1)), None)
    assert rel(loss, L.detach().reshape(1)) < 1e-6 and rel(grad, so.grad) < 1e-5


@pytest.mark.parametrize("H,W,scale", [(33, 33, 0.5), (65, 65, 0.5), (65, 65, 0.125), (46, 61, 0.5), (9, 9, 0.12)])
def test_pairwise_c_vs_torch(ref, H, W, scale):
    g = torch.Generator().manual_seed(H)
    B, Cs, Ct = 2, 6, 10
    fs, ft = torch.randn(B, Cs, H, W, generator=g), torch.randn(B, Ct, H, W, generator=g)
    fso = fs.double().requires_grad_(True)
    preds = lambda f: [None, None, f, None, None, None, None]
    L = O.criterion_pair_wise(preds(fso), preds(ft.double()), scale, -5)
    (0.5 * L).backward()
    kh, kw = O.pair_wise_pool_window(ft.shape, scale)
    kh, kw = max(kh, 1), max(kw, 1)
    OH, OW = -(-H // kh), -(-W // kw)
    M = OH * OW
    ps, pt = torch.empty(B, Cs, M), torch.empty(B, Ct, M)
    idx = torch.empty(B, Cs, M, dtype=torch.int32)
    assert ref.skd_maxpool_argmax(B * Cs, H, W, kh, kw, P(fs), P(ps), P(idx), None)
    assert ref.skd_maxpool_argmax(B * Ct, H, W, kh, kw, P(ft), P(pt), None, None)
    tp, ti = F.max_pool2d(fs, (kh, kw), (kh, kw), 0, ceil_mode=True, return_indices=True)
    assert torch.equal(ti.reshape(B, Cs, M).int(), idx) and torch.equal(tp.reshape(B, Cs, M), ps)
    ldm, ldc = ref.skd_pairwise_ldm(M), 128
    fhs, fht = torch.empty(B, Cs, ldm), torch.empty(B, Ct, ldm)
    fst, nrm = torch.empty(B, ldm, ldc), torch.empty(B, M)
    assert ref.skd_channel_l2_normalise(B, Cs, M, P(ps), P(fhs), ldm, P(fst), ldc, P(nrm), None)
    assert ref.skd_channel_l2_normalise(B, Ct, M, P(pt), P(fht), ldm, None, 0, None, None)
    G, loss = torch.empty(B, ldm, ldm), torch.empty(1)
    assert ref.skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, P(fhs), P(fht), P(G), P(loss), P(torch.empty(1)), None)
    assert rel(loss, L.detach().reshape(1)) < 1e-5
    dp, dx = torch.empty(B, Cs, ldm), torch.empty(B, Cs, H, W)
    assert ref.skd_pairwise_backward(B, Cs, M, ldm, P(fhs), P(G), P(nrm), P(torch.tensor([0.5])), P(dp), None, None)
    assert ref.skd_maxunpool_scatter(B * Cs, H, W, kh, kw, P(dp), ldm, P(idx), P(dx), None)
    assert rel(dx, fso.grad) < 1e-4


def test_spectral_c_vs_torch(ref):
    g = torch.Generator().manual_seed(2)
    W = torch.randn(12, 40, generator=g) * 0.1
    u, v = torch.randn(12, generator=g), torch.randn(40, generator=g)
    Pd = {"weight_bar": W.double().reshape(12, 10, 2, 2).requires_grad_(True), "weight_u": (u / u.norm()).double(),
          "weight_v": (v / v.norm()).double()}
    uc, vc = (u / u.norm()).clone(), (v / v.norm()).clone()
    w_t = O.spectral_weight(Pd, "")
    gw = torch.randn(12, 40, generator=g)
    (w_t.reshape(12, 40) * gw.double()).sum().backward()
    sig, wo = torch.empty(1), torch.empty(12, 40)
    assert ref.skd_spectral_norm_forward(12, 40, P(W), P(uc), P(vc), P(sig), P(wo), P(torch.empty(1)), None)
    assert rel(wo, w_t.detach().reshape(12, 40)) < 1e-5 and rel(uc, Pd["weight_u"]) < 1e-5 and rel(vc, Pd["weight_v"]) < 1e-5
    gwb = torch.empty(12, 40)
    assert ref.skd_spectral_norm_backward(12, 40, P(W), P(uc), P(vc), P(sig), P(gw), P(gwb), P(torch.empty(1)), None)
    assert rel(gwb, Pd["weight_bar"].grad.reshape(12, 40)) < 1e-4


@pytest.mark.parametrize("geom", [(2, 19, 9, 9, 65, 65), (1, 5, 7, 4, 20, 33), (2, 11, 6, 8, 41, 57), (1, 3, 1, 1, 4, 4), (1, 4, 5, 5, 5, 5)])
def test_ce_dsn_c_vs_torch(ref, geom):
    B, C, h, w, H, W = geom
    g = torch.Generator().manual_seed(H)
    lm, ld = torch.randn(B, C, h, w, generator=g) * 3, torch.randn(B, C, h, w, generator=g) * 3
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[0, : max(1, H // 8)] = 255
    lmo, ldo = lm.double().requires_grad_(True), ld.double().requires_grad_(True)
    L = O.criterion_dsn([lmo, ldo], y)
    L.backward()
    loss, gm, gd = torch.empty(1), torch.empty_like(lm), torch.empty_like(ld)
    assert ref.skd_ce_dsn_forward(B, C, h, w, H, W, P(lm), P(ld), P(y), 255, 0.4, P(loss), P(gm), P(gd), P(torch.empty(8)), None)
    assert rel(loss, L.detach().reshape(1)) < 1e-6
    assert rel(gm, lmo.grad) < 2e-5 and rel(gd, ldo.grad) < 2e-5
    # single head, no gradients
    assert ref.skd_ce_dsn_forward(B, C, h, w, H, W, P(lm), None, P(y), 255, 0.4, P(loss), None, None, P(torch.empty(8)), None)
    want = F.cross_entropy(F.interpolate(lm.double(), size=(H, W), mode="bilinear", align_corners=True), y, ignore_index=255)
    assert rel(loss, want.reshape(1)) < 1e-6


@pytest.mark.parametrize("geom", [(2, 5, 65, 65), (1, 3, 33, 33), (2, 4, 46, 61), (1, 2, 7, 9)])
def test_ppm_c_vs_torch(ref, geom):
    """AdaptiveAvgPool2d(1,2,3,6), bilinear(align_corners) up-sampling + cat, and their backward."""
    B, C, H, W = geom
    sizes = (1, 2, 3, 6)
    arr = (ctypes.c_int * 4)(*sizes)
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, C, H, W, generator=g)
    xo = x.double().requires_grad_(True)
    want = [F.adaptive_avg_pool2d(xo, s) for s in sizes]
    total = ref.skd_ppm_pooled_floats(B * C, 4, arr)
    assert total == B * C * 50
    pooled = torch.empty(total)
    assert ref.skd_ppm_pool(B * C, H, W, 4, arr, P(x), P(pooled), None)
    off, gflat = 0, []
    for s, w_ in zip(sizes, want):
        n = B * C * s * s
        assert rel(pooled[off:off + n].view(B, C, s, s), w_.detach()) < 1e-6
        gflat.append(torch.randn(B, C, s, s, generator=g))
        off += n
    sum((w_ * gg.double()).sum() for w_, gg in zip(want, gflat)).backward()
    dx = torch.empty_like(x)
    assert ref.skd_ppm_pool_backward(B * C, H, W, 4, arr, P(torch.cat([t.reshape(-1) for t in gflat])), P(dx), None)
    assert rel(dx, xo.grad) < 1e-6
    # concat
    Cout = 3
    priors = [torch.randn(B, Cout, s, s, generator=g) for s in sizes]
    po = [p.double().requires_grad_(True) for p in priors]
    fo = x.double().requires_grad_(True)
    cat_o = torch.cat([F.interpolate(p, size=(H, W), mode="bilinear", align_corners=True) for p in po] + [fo], 1)
    cat = torch.empty(B, 4 * Cout + C, H, W)
    ptrs = (ctypes.c_void_p * 4)(*[p.data_ptr() for p in priors])
    assert ref.skd_ppm_concat(B, Cout, C, H, W, 4, arr, ptrs, P(x), P(cat), None)
    assert rel(cat, cat_o.detach()) < 1e-6
    gc = torch.randn(B, 4 * Cout + C, H, W, generator=g)
    (cat_o * gc.double()).sum().backward()
    gp = [torch.empty_like(p) for p in priors]
    gptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in gp])
    assert ref.skd_ppm_concat_backward(B, Cout, C, H, W, 4, arr, P(gc), gptrs, None)
    for a, b in zip(gp, po):
        assert rel(a, b.grad) < 1e-5
    assert rel(gc[:, 4 * Cout:], fo.grad) == 0.0


@pytest.mark.parametrize("geom", [(2, 19, 9, 17, 65, 129), (1, 5, 7, 4, 20, 33), (1, 3, 1, 1, 4, 4), (2, 11, 6, 6, 6, 6)])
def test_seg_confusion_c_vs_torch(ref, geom):
    """upsample (align_corners) + argmax + confusion matrix, networks/evaluate.py:106-113, 136-154, 186-198."""
    B, C, h, w, H, W = geom
    g = torch.Generator().manual_seed(H * W)
    lg = torch.randn(B, C, h, w, generator=g) * 5
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[0, : max(1, H // 8)] = 255
    pred = torch.empty(B, H, W, dtype=torch.uint8)
    conf = torch.zeros(C, C, dtype=torch.int64)
    assert ref.skd_seg_confusion(B, C, h, w, H, W, P(lg), P(y), 255, P(pred), P(conf), None)
    up = F.interpolate(lg.double(), size=(H, W), mode="bilinear", align_corners=True)
    want_pred = up.argmax(1)
    flips = int((want_pred != pred.long()).sum())
    assert flips <= max(1, B * H * W // 100000), flips          # fp32 vs fp64 interpolation: near-ties only
    valid = y != 255
    idx = (y[valid] * C + pred.long()[valid])
    want_conf = torch.bincount(idx, minlength=C * C).reshape(C, C)      # get_confusion_matrix, evaluate.py:144-152
    assert torch.equal(conf, want_conf)
    assert int(conf.sum()) == int(valid.sum())
    # accumulates (+=) and supports prediction-only calls
    assert ref.skd_seg_confusion(B, C, h, w, H, W, P(lg), P(y), 255, None, P(conf), None)
    assert torch.equal(conf, 2 * want_conf)
    pred2 = torch.empty_like(pred)
    assert ref.skd_seg_confusion(B, C, h, w, H, W, P(lg), None, 255, P(pred2), None, None)
    assert torch.equal(pred2, pred)
