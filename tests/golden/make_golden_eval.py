"""Generates tests/golden/reference_eval.pt by running the REFERENCE's own networks/evaluate.py (imported from
/root/reference through oracle/ref_import.load_reference_evaluate: cv2 / torchvision stubbed, nothing copied) on seeded
inputs.  Only runnable in the build container; the fixture travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py

Pins: ``get_confusion_matrix`` (evaluate.py:136-154) on seeded label / prediction vectors, and the
whole ``evaluate_main(..., whole=True)`` recipe (evaluate.py:106-113, 156-206: forward, bilinear align_corners upsample to
1024 x 2048, argmax, crop to ``size``, ignore mask, confusion, IoU) with a small seeded network standing in for the
student, so that the product's helpers, fused kernel (csrc/evaluate.hip) and evaluate_main are checked against outputs of
the reference's code rather than against the product's own helpers.
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import abn_torch, ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_eval.pt")
CONFUSION_CASES = [(101, 5000, 19, 19), (102, 4096, 19, 11), (103, 777, 7, 7), (104, 1, 19, 1)]   # (seed, n, classes, labels actually used)


def confusion_case(seed, n, classes, used):
    """Seeded (gt, pred) vectors; ``used`` < classes leaves the top of the bincount empty (evaluate.py:150 guard)."""
    g = np.random.RandomState(seed)
    return g.randint(0, used, size=n).astype(np.int64), g.randint(0, used, size=n).astype(np.uint8)


class FakeStudent(torch.nn.Module):
    """3 -> 19 channels, 8 x 8 stride-8 convolution with seeded weights: (1, 19, 128, 256) logits from a 1024 x 2048 image,
    returned as a list like Res_pspnet.forward (evaluate.py:109-110 takes element 0)."""

    def __init__(self, seed=7):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.conv = torch.nn.Conv2d(3, 19, 8, 8)
        with torch.no_grad():
            self.conv.weight.copy_(torch.randn(19, 3, 8, 8, generator=g) * 0.02)
            self.conv.bias.copy_(torch.randn(19, generator=g) * 0.5)

    def forward(self, x):
        y = self.conv(x)
        return [y, y]


def eval_batches(seed=9):
    """Two (image, label, size, name) batches as CSDataSet yields them for the validation list (datasets.py:121-210)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    # size == the full tile for both: evaluate.py:194-197 crops the LABEL to size but not the prediction, so the reference
    # itself only runs when they coincide (every Cityscapes validation image is 1024 x 2048)
    for i, (h, w) in enumerate(((1024, 2048), (1024, 2048))):
        image = torch.randn(1, 3, 1024, 2048, generator=g) * 57.0
        label = torch.randint(0, 19, (1, 1024, 2048), generator=g).float()
        label[0, 10 + 100 * i:40 + 300 * i, :300 + 500 * i] = 255
        out.append((image, label, torch.tensor([[h, w, 3]]), ["img%d" % i]))
    return out


def main():
    E = ref_import.load_reference_evaluate(abn_torch)
    G = {"confusion": [], "cases": CONFUSION_CASES}
    for seed, n, classes, used in CONFUSION_CASES:
        gt, pred = confusion_case(seed, n, classes, used)
        G["confusion"].append(torch.from_numpy(E.get_confusion_matrix(gt, pred, classes)))
    per_image = []
    orig = E.get_confusion_matrix
    E.get_confusion_matrix = lambda *a: per_image.append(orig(*a)) or per_image[-1]
    cwd = os.getcwd()
    try:
        with tempfile.TemporaryDirectory() as d, ref_import.evaluate_shims():
            os.chdir(d)                                   # evaluate.py:174-175,191 writes outputs/<name>.png
            mean_iu, iu = E.evaluate_main(FakeStudent(), eval_batches(), "0", "512,512", 19, True)
    finally:
        os.chdir(cwd)
        E.get_confusion_matrix = orig
    G["evaluate_main"] = {"net_seed": 7, "batch_seed": 9, "mean_IU": float(mean_iu), "IU_array": torch.from_numpy(np.asarray(iu)),
                          "confusion_per_image": [torch.from_numpy(c) for c in per_image]}
    torch.save(G, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; mean IU", mean_iu)


if __name__ == "__main__":
    main()
