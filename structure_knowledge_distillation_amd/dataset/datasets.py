"""Cityscapes training loader with the per-sample transform on the GPU.

Reference: dataset/datasets.py:121-210 (``CSDataSet``).  There every sample costs one CPU core a cv2.resize of a
1024 x 2048 image by up to 2.1x, a float conversion, padding, a crop and a flip; at the ~100 images/s per GPU of the
distillation step (800 images/s on a node) that loader is the bottleneck.  Here the host keeps what must stay on the
host -- reading / decoding the PNGs and drawing the random numbers, in the reference's order and from the same
generators (``random`` and ``np.random``) -- and the whole arithmetic runs as ONE kernel per batch
(csrc/input_pipeline.hip, ``skd_cs_transform``) straight from the decoded uint8 images: bit-exact labels and image
values (tests/test_input_pipeline_gpu.py against oracle/input_ref.c), output already in the channels-last layout the
networks run in.

    ds = CSDataSet(root, list_path, crop_size=(512, 512), mean=IMG_MEAN)          # decodes on the host (needs cv2)
    tf = CSTrainTransform(crop_size=(512, 512), mean=IMG_MEAN, device="cuda")
    images, labels = tf(raw_images_u8, raw_labels_u8)                              # (B,3,h,w) fp32, (B,h,w) int64
"""
import os.path as osp
import random

import numpy as np
import torch

from .. import _lib

# datasets.py:143-148
ID_TO_TRAINID = {-1: 255, 0: 255, 1: 255, 2: 255, 3: 255, 4: 255, 5: 255, 6: 255, 7: 0, 8: 1, 9: 255, 10: 255, 11: 2, 12: 3,
                 13: 4, 14: 255, 15: 255, 16: 255, 17: 5, 18: 255, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10, 24: 11, 25: 12,
                 26: 13, 27: 14, 28: 15, 29: 255, 30: 255, 31: 16, 32: 17, 33: 18}


def trainid_lut(ignore_label=255):
    """256-entry look-up table of ``id2trainId`` (datasets.py:162-171): ids outside the table keep their value."""
    lut = np.arange(256, dtype=np.uint8)
    for k, v in ID_TO_TRAINID.items():
        if 0 <= k < 256:
            lut[k] = ignore_label if v == 255 else v
    return lut


def draw_sample_params(src_h, src_w, crop_h, crop_w, scale=True, mirror=True):
    """The random draws of ONE ``CSDataSet.__getitem__`` call, in its order and from its generators:
    ``random.randint(0, 14)`` for the scale (:158), ``random.randint`` for h_off then w_off on the padded size
    (:198-199), ``np.random.choice(2)`` for the mirror (:206).  Returns (f, dst_h, dst_w, h_off, w_off, flip)."""
    f = 0.7 + random.randint(0, 14) / 10.0 if scale else 1.0
    # cv2.resize(None, fx=f, fy=f): dsize = (cvRound(w * f), cvRound(h * f)); Python's round() is the same half-to-even
    dst_h, dst_w = int(round(src_h * f)), int(round(src_w * f))
    pad_h, pad_w = max(dst_h, crop_h), max(dst_w, crop_w)
    h_off = random.randint(0, pad_h - crop_h)
    w_off = random.randint(0, pad_w - crop_w)
    flip = int(np.random.choice(2) * 2 - 1) if mirror else 1
    return f, dst_h, dst_w, h_off, w_off, flip


class CSTrainTransform:
    """Batch form of the transform of ``CSDataSet.__getitem__`` (datasets.py:173-210) on the device."""

    def __init__(self, crop_size=(321, 321), mean=(128, 128, 128), scale=True, mirror=True, ignore_label=255,
                 device="cuda", channels_last=True):
        self.crop_h, self.crop_w = crop_size
        self.mean = np.asarray(mean, dtype=np.float32).reshape(3)
        self.scale, self.is_mirror, self.ignore_label = scale, mirror, ignore_label
        self.device = torch.device(device)
        self.channels_last = channels_last
        self._lut = torch.from_numpy(trainid_lut(ignore_label)).to(self.device)

    def __call__(self, images, labels=None, params=None):
        """images (B, H0, W0, 3) uint8 BGR, labels (B, H0, W0) uint8 raw ids (host or device tensors / arrays).
        ``params``: optional list of per-sample (f, dst_h, dst_w, h_off, w_off, flip) -- drawn here when absent."""
        images = torch.as_tensor(images)
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError("images must be (B, H, W, 3) uint8 (decoded BGR, as cv2.imread returns them)")
        B, H0, W0 = images.shape[0], images.shape[1], images.shape[2]
        if labels is not None:
            labels = torch.as_tensor(labels)
            if labels.dtype != torch.uint8 or tuple(labels.shape) != (B, H0, W0):
                raise ValueError("labels must be (B, H, W) uint8 raw ids")
        if params is None:
            params = [draw_sample_params(H0, W0, self.crop_h, self.crop_w, self.scale, self.is_mirror) for _ in range(B)]
        dev = self.device
        images = images.to(dev, non_blocking=True).contiguous()
        labels = labels.to(dev, non_blocking=True).contiguous() if labels is not None else None
        _lib.require_device(images, labels)
        p = np.asarray(params, dtype=np.float64).reshape(B, 6)
        f = torch.from_numpy(np.ascontiguousarray(p[:, 0])).to(dev)
        ints = torch.from_numpy(np.ascontiguousarray(p[:, 1:].T.astype(np.int32))).to(dev)     # rows: dst_h, dst_w, h_off, w_off, flip
        if self.channels_last:
            out = torch.empty((B, self.crop_h, self.crop_w, 3), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
        else:
            out = torch.empty((B, 3, self.crop_h, self.crop_w), dtype=torch.float32, device=dev)
        lab = torch.empty((B, self.crop_h, self.crop_w), dtype=torch.int64, device=dev) if labels is not None else None
        import ctypes
        mean = (ctypes.c_float * 3)(*[float(v) for v in self.mean])
        _lib.check(_lib.get().skd_cs_transform(B, H0, W0, images.data_ptr(), _lib.ptr(labels), self._lut.data_ptr(), f.data_ptr(),
                                               ints[0].data_ptr(), ints[1].data_ptr(), ints[2].data_ptr(), ints[3].data_ptr(),
                                               ints[4].data_ptr(), self.crop_h, self.crop_w, mean, int(self.ignore_label),
                                               out.data_ptr(), 1 if self.channels_last else 0, _lib.ptr(lab),
                                               _lib.stream_of(images)), "skd_cs_transform")
        return (out, lab) if labels is not None else out


_TRANSFORMS = {}


def _transform_for(cfg, device):
    """One CSTrainTransform per (configuration, device) and process, built on first use -- i.e. in the process that calls
    it, never in a loader worker."""
    key = (cfg, str(torch.device(device)))
    tf = _TRANSFORMS.get(key)
    if tf is None:
        crop, mean, scale, mirror, ignore = cfg
        tf = _TRANSFORMS[key] = CSTrainTransform(crop, mean, scale, mirror, ignore, device)
    return tf


class RawBatch:
    """What ``CSDataSet.collate`` hands from a DataLoader WORKER to the training process: the decoded uint8 images and
    raw-id labels of a batch stacked on the host, the per-sample random draws (taken in the worker, where the reference
    takes them: ``__getitem__`` runs there, datasets.py:158,198-199,206), and the transform's configuration.  Holds no
    device tensor, so it pickles through the worker queues under fork and spawn alike; ``to_device`` -- called by
    ``NetModel.set_input`` in the process that owns the GPU -- runs the one-kernel transform."""

    def __init__(self, images, labels, params, cfg):
        self.images, self.labels, self.params, self.cfg = images, labels, params, cfg

    def pin_memory(self):                       # DataLoader(pin_memory=True) calls this on custom batch objects
        self.images, self.labels = self.images.pin_memory(), self.labels.pin_memory()
        return self

    def __len__(self):
        return self.images.shape[0]

    def to_device(self, device):
        """-> (images (B,3,h,w) fp32 channels-last, labels (B,h,w) int64) on ``device``."""
        return _transform_for(self.cfg, device)(self.images, self.labels, self.params)


class CSDataSet(torch.utils.data.Dataset):
    """Same constructor as the reference's ``CSDataSet`` (datasets.py:122-150).  ``__getitem__`` only READS: it returns
    the decoded uint8 image (H, W, 3) BGR, the raw-id label (H, W), ``size`` and ``name``; the transform -- everything
    the reference does between imread and the return (:176-210) -- is applied per batch on the device.  Decoding needs
    cv2, like the reference; nothing else here does.

        loader = DataLoader(ds, batch_size=b, collate_fn=ds.collate, num_workers=4, pin_memory=True)
        for data in loader: model.set_input(data)      # data[0] is a RawBatch; set_input runs the device transform

    ``collate`` is host-only (it runs in the forked / spawned workers, which must not touch the GPU)."""

    def __init__(self, root, list_path, max_iters=None, crop_size=(321, 321), mean=(128, 128, 128), scale=True, mirror=True,
                 ignore_label=255, device="cuda"):
        self.root, self.list_path = root, list_path
        self.crop_h, self.crop_w = crop_size
        self.scale, self.ignore_label, self.mean, self.is_mirror = scale, ignore_label, mean, mirror
        self.img_ids = [i_id.strip().split() for i_id in open(list_path)]
        if max_iters is not None:
            self.img_ids = self.img_ids * int(np.ceil(float(max_iters) / len(self.img_ids)))
        self.files = []
        for image_path, label_path in self.img_ids:
            name = osp.splitext(osp.basename(label_path))[0]
            self.files.append({"img": osp.join(self.root, image_path), "label": osp.join(self.root, label_path), "name": name})
        self.id_to_trainid = dict(ID_TO_TRAINID)
        self.device = device
        self._cfg = ((int(self.crop_h), int(self.crop_w)), tuple(float(m) for m in np.asarray(mean).reshape(3)), bool(scale),
                     bool(mirror), int(ignore_label))

    @property
    def transform(self):
        """The device transform of this dataset's configuration (built on first use, in the calling process)."""
        return _transform_for(self._cfg, self.device)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        import cv2     # host-side decode, exactly the reference's two imread calls (datasets.py:175-176)
        datafiles = self.files[index]
        image = cv2.imread(datafiles["img"], cv2.IMREAD_COLOR)
        label = cv2.imread(datafiles["label"], cv2.IMREAD_GRAYSCALE)
        return torch.from_numpy(image), torch.from_numpy(label), np.array(image.shape), datafiles["name"]

    def collate(self, batch):
        """``collate_fn`` for the DataLoader -- HOST ONLY: stacks the raw samples and draws the per-sample random numbers.
        Returns the reference's batch tuple (images, labels, size, name) with ``images`` a ``RawBatch`` (and ``labels``
        None: they travel inside it); ``NetModel.set_input`` turns it into device tensors."""
        images = torch.stack([b[0] for b in batch])
        labels = torch.stack([b[1] for b in batch])
        H0, W0 = images.shape[1], images.shape[2]
        params = [draw_sample_params(H0, W0, self.crop_h, self.crop_w, self.scale, self.is_mirror) for _ in batch]
        return RawBatch(images, labels, params, self._cfg), None, np.stack([b[2] for b in batch]), [b[3] for b in batch]
