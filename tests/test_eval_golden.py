"""Evaluation path pinned to outputs of the REFERENCE's own networks/evaluate.py:
tests/golden/reference_eval.pt was written by tests/golden/make_golden_eval.py, which imports the reference's module from
/root/reference and runs ``get_confusion_matrix`` (evaluate.py:136-154) and the whole ``evaluate_main(whole=True)`` recipe
(evaluate.py:106-113,156-206) on seeded inputs.  Checked against it: the product's host helpers (bit-exact), and
``evaluate_main`` with the fused upsample + argmax + confusion kernel -- through the C-ABI double on the CPU, through
csrc/evaluate.hip on the GPU.  The logits come from a convolution evaluated by a different library on each side
(MKL-DNN when the fixture was made; MIOpen / rocBLAS on the GPU), so a handful of near-tie argmax decisions among the
2 x 2M pixels may flip: the per-image confusion matrices must agree on all but 1e-5 of the pixels."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import abn_torch, cref, ref_import
from structure_knowledge_distillation_amd import _lib
from structure_knowledge_distillation_amd.networks import evaluate as E

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_eval", os.path.join(GOLDEN_DIR, "make_golden_eval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _gold():
    return torch.load(os.path.join(GOLDEN_DIR, "reference_eval.pt"), weights_only=False)


def test_confusion_matrix_and_iou_helpers_vs_reference_fixture():
    gen, gold = _gen(), _gold()
    assert [tuple(c) for c in gold["cases"]] == [tuple(c) for c in gen.CONFUSION_CASES]
    for (seed, n, classes, used), want in zip(gen.CONFUSION_CASES, gold["confusion"]):
        gt, pred = gen.confusion_case(seed, n, classes, used)
        got = E.get_confusion_matrix(gt, pred, classes)
        assert got.shape == (classes, classes) and np.array_equal(got, want.numpy()), (seed, n, classes)
    ev = gold["evaluate_main"]
    total = sum(c.numpy() for c in ev["confusion_per_image"])
    mean_iu, iu = E.iou_from_confusion(total)                               # evaluate.py:200-205
    assert abs(mean_iu - ev["mean_IU"]) <= 1e-15 and np.array_equal(np.asarray(iu), ev["IU_array"].numpy())


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")
def test_confusion_matrix_vs_live_reference_module():
    R = ref_import.load_reference_evaluate(abn_torch)
    g = np.random.RandomState(5)
    for classes, used in ((19, 19), (19, 3), (5, 5)):
        gt, pred = g.randint(0, used, 3001), g.randint(0, used, 3001).astype(np.uint8)
        assert np.array_equal(E.get_confusion_matrix(gt, pred, classes), R.get_confusion_matrix(gt, pred, classes))


def _run_evaluate_main(device):
    gen, gold = _gen(), _gold()
    ev = gold["evaluate_main"]
    net = gen.FakeStudent(ev["net_seed"]).to(device)
    per_image = []
    for batch in gen.eval_batches(ev["batch_seed"]):
        # one image at a time so that each image's confusion matrix can be compared with the reference's
        m, iu = E.evaluate_main(net, [batch], "0", "512,512", 19, whole=True)
        per_image.append((m, np.asarray(iu)))
    mean_iu, iu = E.evaluate_main(net, gen.eval_batches(ev["batch_seed"]), "0", "512,512", 19, whole=True)
    want_total = sum(c.numpy() for c in ev["confusion_per_image"])
    pixels = want_total.sum()
    # IoU moves by at most (flipped pixels) / (class support): 1e-5 of the pixels over ~1/19 of them per class
    assert abs(mean_iu - ev["mean_IU"]) <= 5e-4 * ev["mean_IU"] + 1e-6, (mean_iu, ev["mean_IU"])
    assert np.abs(np.asarray(iu) - ev["IU_array"].numpy()).max() <= 2e-5
    for (m, iu_i), want in zip(per_image, ev["confusion_per_image"]):
        wm, wiu = E.iou_from_confusion(want.numpy())
        assert np.abs(iu_i - wiu).max() <= 2e-5 and abs(m - wm) <= 2e-5
    return mean_iu, pixels


def test_evaluate_main_vs_reference_fixture_through_c_double():
    _lib.install_test_backend(cref.load(_lib.SIGNATURES))
    try:
        _run_evaluate_main(torch.device("cpu"))
    finally:
        _lib.install_test_backend(None)


@pytest.mark.gpu
def test_evaluate_main_vs_reference_fixture_on_gpu():
    """csrc/evaluate.hip (fused upsample + argmax + confusion) inside evaluate_main against the reference's own
    evaluate_main output, and the raw confusion counts of the kernel against the reference's per-image matrices."""
    from structure_knowledge_distillation_amd import functional as SF
    gen, gold = _gen(), _gold()
    dev = torch.device("cuda", 0)
    _run_evaluate_main(dev)
    ev = gold["evaluate_main"]
    net = gen.FakeStudent(ev["net_seed"]).to(dev)
    for batch, want in zip(gen.eval_batches(ev["batch_seed"]), ev["confusion_per_image"]):
        with torch.no_grad():
            logits = net(batch[0].to(dev))[0]
        cm = torch.zeros(19, 19, dtype=torch.int64, device=dev)
        SF.seg_confusion(logits, batch[1].long().to(dev), 255, cm, want_pred=False)
        diff = np.abs(cm.cpu().numpy() - want.numpy()).sum() / 2          # a flipped pixel moves one count
        assert cm.sum().item() == want.sum().item(), "same number of scored pixels (ignore mask, evaluate.py:195-197)"
        print("confusion vs the reference's: %d of %d scored pixels differ" % (diff, want.sum().item()))
        assert diff <= 1e-5 * want.sum().item() + 2
