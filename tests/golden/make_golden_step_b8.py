"""Generates tests/golden/step_b8_oracle.pt: ONE full distillation step at the BENCHMARKED configuration
(BASELINE configs[2]: batch 8, 512x512, Pi + Pa + Ho, wgan-gp) from the CPU oracle (oracle/step_torch.py, itself
pinned to the reference's own Python by tests/test_oracle_vs_reference.py and tests/golden/reference_vectors.pt),
in fp64 AND in fp32, so that the GPU test (tests/test_step_gpu.py::test_full_step_b8_vs_golden) costs seconds of
GPU-box time instead of minutes of idle GPU while 128 host cores run the oracle.

    python tests/golden/make_golden_step_b8.py          (about 15 min on 8 cores, ~25 GB of host memory)

Nothing large is stored: weights come from oracle.step_torch.*_init(seed) (seeded torch CPU RNG, same torch build
here and on the GPU box; the fixture carries checksums so a drifted RNG is detected, not trusted), inputs from
synthetic_batch(seed).  Per tensor the fixture holds: the fp64 value's norm / sum / strided sample, and the norm of
(fp32-oracle - fp64-oracle) = the CPU-fp32 error the SURVEY.md section 8c gradient bound is stated against.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import step_torch as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_b8_oracle.pt")
SEEDS = {"student": 141, "teacher": 142, "D": 143, "batch": 144, "alpha": 145}
B, H, W = 8, 512, 512
NSAMPLE = 1024


def sample_idx(numel, n=NSAMPLE):
    step = max(1, numel // n)
    return step, min(n, -(-numel // step))


def rec(t64, t32=None, n=NSAMPLE):
    f = t64.detach().double().reshape(-1)
    step, cnt = sample_idx(f.numel(), n)
    r = {"shape": list(t64.shape), "step": step, "sample": f[::step][:cnt].clone(), "norm": float(f.norm()),
         "sum": float(f.sum())}
    if t32 is not None:
        r["base"] = float((t32.detach().double().reshape(-1) - f).norm())      # CPU fp32 oracle vs fp64 oracle
    return r


def init(dtype=torch.float32):
    """fp32 weights (what the GPU holds); the fp64 oracle run uses the SAME values widened to fp64."""
    PS = O.pspnet_init(O.STUDENT, 19, seed=SEEDS["student"])
    PT = O.pspnet_init(O.TEACHER, 19, seed=SEEDS["teacher"])
    PD = O.discriminator_init(seed=SEEDS["D"])
    PD["attn1.gamma"].fill_(0.25)      # make the attention branches live (gamma is 0 at init)
    PD["attn2.gamma"].fill_(-0.5)
    cast = lambda P: {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in P.items()}
    return cast(PS), cast(PT), cast(PD)


def checksum(P):
    return {k: float(v.double().sum()) for k, v in list(P.items())[:3] + list(P.items())[-3:]}


def main():
    cfg = O.StepConfig(weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)   # run_train_val.sh overrides; dropout off
    images, labels = O.synthetic_batch(B, H, W, seed=SEEDS["batch"])
    alpha = torch.rand(B, 1, 1, 1, generator=torch.Generator().manual_seed(SEEDS["alpha"]))
    outs, params = {}, {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        t0 = time.time()
        PS, PT, PD = init(dt)
        if name == "f64":
            sums = {"student": checksum(PS), "teacher": checksum(PT), "D": checksum(PD)}
        outs[name] = O.distillation_step(PS, PT, PD, images.to(dt), labels, cfg, {"G": {}, "D": {}}, alpha.to(dt))
        params[name] = (PS, PD)
        print("oracle %s step: %.0f s" % (name, time.time() - t0), flush=True)
    o64, o32 = outs["f64"], outs["f32"]
    G = {"seeds": SEEDS, "shape": (B, H, W), "cfg": {"weight_decay": 5e-4, "lambda_pa": 0.5, "dropout_p": 0.0},
         "checksums": sums,
         "losses64": {k: o64[k] for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss", "D_loss")},
         "losses32": {k: o32[k] for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss", "D_loss")},
         "preds_S": [rec(a, b) for a, b in zip(o64["preds_S"], o32["preds_S"])],
         "preds_T": [rec(a, b) for a, b in zip(o64["preds_T"][:3], o32["preds_T"][:3])],
         "grads_S": {k: rec(g, o32["grads_S"][k]) for k, g in o64["grads_S"].items() if g is not None},
         "grads_D": {k: rec(g, o32["grads_D"][k]) for k, g in o64["grads_D"].items() if g is not None},
         "running": {k: rec(v, params["f32"][0][k]) for k, v in params["f64"][0].items() if "running" in k},
         "student_after": {k: rec(params["f64"][0][k], params["f32"][0][k]) for k in O.learnable_keys(params["f64"][0])},
         "D_after": {k: rec(params["f64"][1][k], params["f32"][1][k]) for k in params["f64"][1]
                     if params["f64"][1][k].is_floating_point()}}
    torch.save(G, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    print({k: (o64[k], o32[k]) for k in G["losses64"]})


if __name__ == "__main__":
    main()
