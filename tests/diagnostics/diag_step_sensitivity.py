"""CPU diagnostic (fp64 oracle only, no GPU): how smooth are the STUDENT's gradients in the input images?

    python tests/diagnostics/diag_step_sensitivity.py [seeds=4]

The B = 2, 512 x 512 step of tests/golden/gpu_suite_oracle.pt["full_step_ho*"] is evaluated in fp64, and again with the images
multiplied by (1 + 1e-6 * N(0, 1)) -- a perturbation of the size of ONE fp32 rounding of the input.  What comes back is the
conditioning of the function the GPU tests compare against a record: the floor of the ONE gradient bound (5e-3 of the tensor norm,
tests/test_step_gpu.py) has to sit above it whatever the kernels do.  Printed per configuration (Pi + Pa; Pi + Pa + Ho): the change
of each loss and the worst relative change of a gradient tensor, per perturbation seed; then ONE perturbation pattern at the
amplitudes 1e-9 ... 1e-5, which separates conditioning (change proportional to the amplitude) from kinks (ReLU / leaky-ReLU slopes,
max-pool and pair-wise-pool arg-max decisions that flip: a change that does not shrink with the amplitude).  Result kept in
profiles/r04t_step_sensitivity.txt: linear with an amplification of ~30 up to 1e-8, then 1.9e-3 at 1e-7 -- below the resolution of
fp32 -- 5.6e-3 at 1e-6 and 1.7e-2 at 1e-5 (median over the 92 gradient tensors): the student's gradient is piecewise smooth with kinks
so dense that any fp32 evaluation sits 2e-3 ... 6e-3 of a tensor's norm from the fp64 record, whatever the kernels do.
"""
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import step_torch as O  # noqa: E402

spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tests", "golden", "make_golden_gpu_suite.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


def step(images, labels, alpha, ho):
    cfg = O.StepConfig(ho=ho, weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    PS, PT, PD = gen.init_nets("full_step", torch.float64)
    o = O.distillation_step(PS, PT, PD if ho else None, images.double(), labels, cfg, {"G": {}, "D": {}}, alpha.double(),
                            lr_g=gen.LR_G, lr_d=gen.LR_D)
    return o


def main(nseeds):
    images, labels, alpha = gen.full_step_inputs(0)
    print("pyramid margins of this batch (min |y| in front of the leaky ReLU, stages 1 2 3 6):",
          ["%.1e" % m for m in gen.pyramid_margins(gen.init_nets("full_step")[0], images)])
    for ho in (False, True):
        t0 = time.time()
        base = step(images, labels, alpha, ho)
        print("%s: reference evaluation %.0f s" % ("Pi + Pa + Ho" if ho else "Pi + Pa", time.time() - t0), flush=True)
        keys = [k for k in gen.LOSS_KEYS if ho or k != "D_loss"]
        for seed in range(nseeds):
            noise = torch.randn(images.shape, generator=torch.Generator().manual_seed(1000 + seed), dtype=torch.float64)
            o = step(images.double() * (1 + 1e-6 * noise), labels, alpha, ho)
            dl = {k: abs(o[k] - base[k]) / (abs(base[k]) + 1e-300) for k in keys}
            rels = sorted(((float((o["grads_S"][k] - g).norm() / (g.norm() + 1e-300)), k) for k, g in base["grads_S"].items()
                           if g is not None and float(g.norm()) > 1e-12), reverse=True)
            med = rels[len(rels) // 2][0]
            print("  images x (1 + 1e-6 N(0,1)), seed %d: losses %s; student gradients: median change %.1e, worst three %s" % (
                seed, {k: "%.0e" % v for k, v in dl.items()}, med, [(k, "%.1e" % e) for e, k in rels[:3]]), flush=True)


def amplitude_sweep():
    images, labels, alpha = gen.full_step_inputs(0)
    base = step(images, labels, alpha, False)
    noise = torch.randn(images.shape, generator=torch.Generator().manual_seed(1000), dtype=torch.float64)
    print("Pi + Pa, one perturbation pattern, amplitude sweep (fp64 throughout):")
    for eps in (1e-9, 1e-8, 1e-7, 1e-6, 1e-5):
        o = step(images.double() * (1 + eps * noise), labels, alpha, False)
        rels = sorted(float((o["grads_S"][k] - g).norm() / (g.norm() + 1e-300)) for k, g in base["grads_S"].items()
                      if g is not None and float(g.norm()) > 1e-12)
        print("  images x (1 + %.0e N(0,1)): student gradients change by median %.2e, max %.2e, min %.2e of the tensor norm" % (
            eps, rels[len(rels) // 2], rels[-1], rels[0]), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    amplitude_sweep()
