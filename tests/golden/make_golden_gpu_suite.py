"""Generates tests/golden/gpu_suite_oracle.pt: every CPU-oracle result the `-m gpu` tests used to compute INSIDE the
test (the driver's GPU box has a noisy 128-thread host; the fp64 + fp32 oracle runs of the full
ResNet101 + ResNet18 step took 400 of the suite's 500 s there and 900+ s on a busy box).  The GPU tests now only run the
HIP path and compare against these records; nothing under oracle/ executes a network forward / backward in a GPU test.

    python tests/golden/make_golden_gpu_suite.py [section ...]      (about 10 min on 8 cores; sections below)

Same conventions as make_golden_step_b8.py: nothing large is stored -- weights come from oracle.step_torch.*_init(seed),
inputs from seeded generators, the fixture carries weight checksums (a drifted torch RNG is detected, not trusted); per
tensor the fixture holds the fp64 oracle's norm / strided sample and ``base`` = ||fp32 oracle - fp64 oracle||, the
conditioning yard-stick of the ONE gradient bound (tests/test_step_gpu.py).

Sections (each pins what the named test used to compute live):
  full_step_ho0 / full_step_ho1   test_full_step_vs_oracle[False / True]   B=2, 512x512, one / two consecutive steps
  config1_pa / config1_pi         test_step_config1_*                      B=2, 256x256 (BASELINE configs[0] shape)
  networks_forward                test_networks_forward_vs_oracle          student (train) + teacher (eval), 161x129 input
  eval_full                       test_evaluate_main_full_size_student_on_gpu   1024x2048 student forward -> confusion matrix
  sharded2                        tests/test_distributed_gpu.py            two shards of B=2, Pi+Pa+Ho, sharded semantics
  sharded8                        tests/test_distributed_gpu.py            EIGHT shards of two images, Pi+Pa+Ho (configs[3]'s world size)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import step_torch as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_suite_oracle.pt")
NSAMPLE = 512
LOSS_KEYS = ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss", "D_loss")
SEEDS = {
    "full_step": {"student": 241, "teacher": 242, "D": 243, "batch": (0, 1), "alpha": (70, 71)},
    "config1": {"student": 41, "teacher": 42, "batch": 43},                    # = reference_vectors.pt["step_config1_pa"]
    "networks_forward": {"student": 251, "teacher": 252, "x": 253},
    "eval_full": {"student": 261, "stats": 262, "data": 263},
    "sharded2": {"student": 271, "teacher": 272, "D": 273, "batch": 3, "alpha": 17},
    # batch seed chosen by pyramid_margins() below: with seed 13 ONE unit of the pyramid's 1 x 1 stage (16 samples x 128 channels, each
    # unit feeding a whole image) sat at y = -8.9e-6 in front of its leaky ReLU; the GPU's y was +4.5e-6, the slope flipped 0.01 -> 1
    # and the recorded gradients of that stage (and, through it, of the backbone) were 10 % away from a CORRECT product
    "sharded8": {"student": 281, "teacher": 282, "D": 283, "batch": 109, "alpha": 27},
}
PYRAMID_MARGIN = (5e-4, 1e-4, 5e-5, 1e-5)      # min |y| in front of the leaky ReLU of the pyramid stages (1, 2, 3, 6) a recorded step must keep
NUM_STEPS, POWER, LR_G, LR_D = 40000, 0.9, 1e-2, 4e-4                         # default_args() / train_options.py


def lr_poly(base, it):
    return base * ((1 - float(it) / NUM_STEPS) ** POWER)                       # kd_model.py:110-112


def sample_idx(numel, n=NSAMPLE):
    step = max(1, numel // n)
    return step, min(n, -(-numel // step))


def rec(t64, t32=None, n=NSAMPLE):
    f = t64.detach().double().reshape(-1)
    step, cnt = sample_idx(f.numel(), n)
    r = {"shape": list(t64.shape), "step": step, "sample": f[::step][:cnt].clone(), "norm": float(f.norm())}
    if t32 is not None:
        r["base"] = float((t32.detach().double().reshape(-1) - f).norm())      # CPU fp32 oracle vs fp64 oracle
    return r


def cast(P, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in P.items()}


def checksum(P):
    return {k: float(v.double().sum()) for k, v in list(P.items())[:3] + list(P.items())[-3:]}


def init_nets(which, dtype=torch.float32, with_d=True):
    """fp32 weights (what the GPU holds); the fp64 oracle run uses the SAME values widened to fp64."""
    s = SEEDS[which]
    PS = O.pspnet_init(O.STUDENT, 19, seed=s["student"])
    PT = O.pspnet_init(O.TEACHER, 19, seed=s["teacher"])
    PD = None
    if with_d and "D" in s:
        PD = O.discriminator_init(seed=s["D"])
        PD["attn1.gamma"].fill_(0.25)      # make the attention branches live (gamma is 0 at init)
        PD["attn2.gamma"].fill_(-0.5)
        PD = cast(PD, dtype)
    return cast(PS, dtype), cast(PT, dtype), PD


def full_step_inputs(step):
    s = SEEDS["full_step"]
    images, labels = O.synthetic_batch(2, 512, 512, seed=s["batch"][step])
    alpha = torch.rand(2, 1, 1, 1, generator=torch.Generator().manual_seed(s["alpha"][step]))
    return images, labels, alpha


def gen_full_step(ho):
    cfg = O.StepConfig(ho=ho, weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    runs, params = {}, {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        PS, PT, PD = init_nets("full_step", dt)
        if name == "f64":
            sums = {"student": checksum(PS), "teacher": checksum(PT), "D": checksum(PD)}
        st, outs = {"G": {}, "D": {}}, []
        for step in range(2 if ho else 1):
            images, labels, alpha = full_step_inputs(step)
            o = O.distillation_step(PS, PT, PD if ho else None, images.to(dt), labels, cfg, st, alpha.to(dt),
                                    lr_g=lr_poly(LR_G, step), lr_d=lr_poly(LR_D, step))
            o["running"] = {k: v.clone() for k, v in PS.items() if "running" in k}
            outs.append(o)
        runs[name], params[name] = outs, PS
    steps = []
    for o64, o32 in zip(runs["f64"], runs["f32"]):
        steps.append({
            "losses64": {k: o64[k] for k in LOSS_KEYS}, "losses32": {k: o32[k] for k in LOSS_KEYS},
            "preds_S": [rec(a, b) for a, b in zip(o64["preds_S"], o32["preds_S"])],
            "grads_S": {k: rec(g, o32["grads_S"][k]) for k, g in o64["grads_S"].items() if g is not None},
            "running": {k: rec(v, o32["running"][k]) for k, v in o64["running"].items()}})
    return {"cfg": {"ho": ho, "weight_decay": 5e-4, "lambda_pa": 0.5}, "checksums": sums, "steps": steps,
            "student_after": {k: rec(params["f64"][k], params["f32"][k]) for k in O.learnable_keys(params["f64"])}}


def config1_weights():
    """(PS, PT) fp32-rounded float64 init of seeds 41 / 42 and the batch of seed 43 -- exactly what make_golden.py fed the
    reference's own modules for reference_vectors.pt['step_config1_pa']."""
    s = SEEDS["config1"]
    PS = O.pspnet_init(O.STUDENT, 19, seed=s["student"], dtype=torch.float64)
    PT = O.pspnet_init(O.TEACHER, 19, seed=s["teacher"], dtype=torch.float64)
    x, y = O.synthetic_batch(2, 256, 256, seed=s["batch"], dtype=torch.float64)
    return cast(PS, torch.float32), cast(PT, torch.float32), x.float(), y


def gen_config1(pa):
    cfg = O.StepConfig(pi=True, pa=pa, ho=False, lambda_pa=0.5, weight_decay=5e-4, dropout_p=0.0)
    outs, params = {}, {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        PS, PT, x, y = config1_weights()
        PS, PT = cast(PS, dt), cast(PT, dt)            # both oracles start from the fp32-rounded weights the GPU holds
        outs[name] = O.distillation_step(PS, PT, None, x.to(dt), y, cfg)
        params[name] = PS
    o64, o32 = outs["f64"], outs["f32"]
    return {"pa": pa, "losses64": {k: o64[k] for k in LOSS_KEYS[:4]},
            "grads_S": {k: rec(g, o32["grads_S"][k]) for k, g in o64["grads_S"].items() if g is not None},
            "student_after": {k: rec(params["f64"][k], params["f32"][k]) for k in O.learnable_keys(params["f64"])}}


def networks_forward_inputs():
    s = SEEDS["networks_forward"]
    PS = O.pspnet_init(O.STUDENT, 19, seed=s["student"])
    PT = O.pspnet_init(O.TEACHER, 19, seed=s["teacher"])
    x = torch.randn(2, 3, 161, 129, generator=torch.Generator().manual_seed(s["x"])) * 57
    return PS, PT, x


def gen_networks_forward():
    PS, PT, x = networks_forward_inputs()
    PS64, PT64 = cast(PS, torch.float64), cast(PT, torch.float64)
    with torch.no_grad():
        s_out = O.pspnet_forward(PS64, x.double(), O.STUDENT, True, dropout_p=0.0)
        t_out = O.pspnet_forward(PT64, x.double(), O.TEACHER, False)
    return {"checksums": {"student": checksum(PS), "teacher": checksum(PT)},
            "student": [rec(t, n=2048) for t in s_out], "teacher": [rec(t, n=2048) for t in t_out],
            "running": {k: rec(v) for k, v in PS64.items() if "running" in k}}


def eval_full_inputs():
    s = SEEDS["eval_full"]
    P = O.pspnet_init(O.STUDENT, 19, seed=s["student"])
    g = torch.Generator().manual_seed(s["stats"])
    for k, v in P.items():                                  # trained-looking statistics: spread logits, fewer argmax ties
        if k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif k.endswith("running_mean"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    H, W = 1024, 2048
    g = torch.Generator().manual_seed(s["data"])
    image = torch.randn(1, 3, H, W, generator=g) * 57
    label = torch.randint(0, 19, (1, H, W), generator=g)
    label[0, :7] = 255
    size = torch.tensor([[H - 10, W - 3, 3]])
    return P, image, label, size


def gen_eval_full():
    """networks/evaluate.py:106-113,186-198 (whole=True) in numpy on the fp64 oracle forward."""
    P, image, label, size = eval_full_inputs()
    H, W = image.shape[2:]
    with torch.no_grad():
        logits = O.pspnet_forward(cast(P, torch.float64), image.double(), O.STUDENT, False)[0]
        up = torch.nn.functional.interpolate(logits, size=(H, W), mode="bilinear", align_corners=True)
    pred = up[0].permute(1, 2, 0).numpy().argmax(2).astype(np.uint8)              # evaluate.py:112, 186
    h, w = int(size[0, 0]), int(size[0, 1])
    gt = label[0].numpy()[:h, :w]
    keep = gt != 255
    idx = gt[keep].astype(np.int64) * 19 + pred[:h, :w][keep].astype(np.int64)    # evaluate.py:136-154 (bincount of gt*C+pred)
    cm = np.bincount(idx, minlength=19 * 19).reshape(19, 19)
    return {"checksums": checksum(P), "confusion": torch.from_numpy(cm.copy()), "pixels": int(keep.sum())}


def sharded2_inputs():
    s = SEEDS["sharded2"]
    x, y = O.synthetic_batch(4, 512, 512, seed=s["batch"])
    alpha = torch.rand(4, 1, 1, 1, generator=torch.Generator().manual_seed(s["alpha"]))
    return x, y, alpha, [slice(0, 2), slice(2, 4)]


def gen_sharded2():
    x, y, alpha, shards = sharded2_inputs()
    cfg = O.StepConfig(weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    outs, after = {}, {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        PS, PT, PD = init_nets("sharded2", dt)
        if name == "f64":
            sums = {"student": checksum(PS), "teacher": checksum(PT), "D": checksum(PD)}
        outs[name] = O.distillation_step_sharded(PS, PT, PD, x.to(dt), y, cfg, shards, [alpha[sl].to(dt) for sl in shards])
        after[name] = (PS, PD)
    o64, o32 = outs["f64"], outs["f32"]
    PS64, PD64 = after["f64"]
    for k in PD64:
        if k.endswith(("weight_u", "weight_v")):
            assert torch.equal(o64["PD_shards"][0][k], o64["PD_shards"][1][k]), k      # replicas hold identical u, v
    return {"cfg": {"weight_decay": 5e-4, "lambda_pa": 0.5}, "checksums": sums,
            "shard_losses": o64["shards"],
            "grads_S": {k: rec(g, o32["grads_S"][k]) for k, g in o64["grads_S"].items() if g is not None},
            "grads_D": {k: rec(g, o32["grads_D"][k]) for k, g in o64["grads_D"].items() if g is not None},
            "running": {k: rec(v) for k, v in PS64.items() if "running" in k},
            "d_bn_running": [{k: v.clone() for k, v in P.items() if k.startswith("preprocess_additional.running")}
                             for P in o64["PD_shards"]],
            "d_uv": {k: v.clone() for k, v in PD64.items() if k.endswith(("weight_u", "weight_v"))}}


def sharded8_inputs():
    s = SEEDS["sharded8"]
    # two images per shard: batch 2 (and 8) are the batch sizes the shipped MIOpen find-db was tuned for -- with one image per rank
    # every rank JIT-compiles ~50 untuned convolution kernels first (measured: 105 s for the test instead of ~30)
    x, y = O.synthetic_batch(16, 512, 512, seed=s["batch"])
    alpha = torch.rand(16, 1, 1, 1, generator=torch.Generator().manual_seed(s["alpha"]))
    return x, y, alpha, [slice(2 * r, 2 * r + 2) for r in range(8)]


def pyramid_margins(PS, x):
    """min |y| over the units of the four pyramid-stage InPlace-ABNs (pspnet_combine.py:94-98) for this batch: the distance of
    the closest unit from the leaky ReLU's kink.  The 1 x 1 stage normalises B nearly identical pooled vectors (variance ~ eps), so
    rounding differences of 1e-5 in the pooled features arrive as 1e-4 in y -- and one flipped unit there rescales the gradient of
    a whole image channel by 100.  A recorded fixture must not sit on such a discontinuity."""
    import torch.nn.functional as F
    with torch.no_grad():
        x4 = O.pspnet_forward({k: v.clone() for k, v in PS.items()}, x, O.STUDENT, True, 0.0)[3].double()
        out = []
        for i, size in enumerate((1, 2, 3, 6)):
            t = F.conv2d(F.adaptive_avg_pool2d(x4, (size, size)), PS["pspmodule.stages.%d.1.weight" % i].double())
            mu, var = t.mean((0, 2, 3), keepdim=True), t.var((0, 2, 3), unbiased=False, keepdim=True)
            out.append(float(((t - mu) / torch.sqrt(var + 1e-5)).abs().min()))
    return out


def gen_sharded8():
    """BASELINE configs[3]'s world size: eight shards of two images each (global batch 16), sharded semantics
    (utils/parallel.py:155, libs/functions.py:185-209, sagan_models.py:148).  A few minutes on 8 cores, ~45 GB of host memory (fp64).

    Two steps from the same weights and inputs:
      * Pi + Pa + Ho (the configs[3] step).  Its student gradients are NOT a smooth function of the logits: the critic's LeakyReLU
        slopes flip for units within rounding distance of zero (measured in round 4: the fp64 critic's gradient moves by 4e-3 ..
        1.3e-2 between two sets of logits that agree to 1.4e-5, on 5 of the 8 shards) -- so the fixture also records the SMOOTH part
        of d G_loss / d logits (CE + pixel-wise KL, ``dlogits_smooth``); the test adds the fp64 critic's gradient on the rank's OWN
        logits to it and holds the product's d G_loss / d logits to that, per rank.
      * Pi + Pa (``pa``): smooth criteria only -- every averaged student gradient under the ONE bound, at world 8."""
    x, y, alpha, shards = sharded8_inputs()
    margins = pyramid_margins(init_nets("sharded8")[0], x)
    assert all(m >= need for m, need in zip(margins, PYRAMID_MARGIN)), \
        "batch seed %r puts a pyramid-stage unit on the leaky ReLU's kink: min |y| %r (need %r)" % (SEEDS["sharded8"]["batch"], margins, PYRAMID_MARGIN)
    cfg = O.StepConfig(weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    outs, after = {}, {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        PS, PT, PD = init_nets("sharded8", dt)
        if name == "f64":
            sums = {"student": checksum(PS), "teacher": checksum(PT), "D": checksum(PD)}
        outs[name] = O.distillation_step_sharded(PS, PT, PD, x.to(dt), y, cfg, shards, [alpha[sl].to(dt) for sl in shards])
        after[name] = (PS, PD)
    o64, o32 = outs["f64"], outs["f32"]
    PS64, PD64 = after["f64"]
    smooth = []
    for sl in shards:                                     # CE + pixel-wise KL of the shard, differentiated at the oracle's own logits
        s = [o64["preds_S"][0][sl].clone().requires_grad_(True), o64["preds_S"][1][sl]]
        t = [o64["preds_T"][0][sl], o64["preds_T"][1][sl]]
        loss = O.criterion_dsn(s, y[sl]) + cfg.lambda_pi * O.criterion_pixel_wise(s, t)
        smooth.append(rec(torch.autograd.grad(loss, s[0])[0], n=4096))
    fx = {"cfg": {"weight_decay": 5e-4, "lambda_pa": 0.5}, "checksums": sums, "shard_losses": o64["shards"],
          "grads_S": {k: rec(g, o32["grads_S"][k]) for k, g in o64["grads_S"].items() if g is not None},
          "grads_D": {k: rec(g, o32["grads_D"][k]) for k, g in o64["grads_D"].items() if g is not None},
          "running": {k: rec(v) for k, v in PS64.items() if "running" in k},
          "d_uv": {k: v.clone() for k, v in PD64.items() if k.endswith(("weight_u", "weight_v"))},
          "dlogits_smooth": smooth, "pyramid_margins": margins}
    del outs, after, o64, o32
    cfg_pa = O.StepConfig(ho=False, weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    outs, after = {}, {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        PS, PT, _ = init_nets("sharded8", dt, with_d=False)
        outs[name] = O.distillation_step_sharded(PS, PT, None, x.to(dt), y, cfg_pa, shards)
        after[name] = PS
    o64, o32 = outs["f64"], outs["f32"]
    fx["pa"] = {"shard_losses": o64["shards"],
                "grads_S": {k: rec(g, o32["grads_S"][k]) for k, g in o64["grads_S"].items() if g is not None},
                "running": {k: rec(v) for k, v in after["f64"].items() if "running" in k}}
    return fx


SECTIONS = {"sharded8": gen_sharded8, "full_step_ho0": lambda: gen_full_step(False), "full_step_ho1": lambda: gen_full_step(True),
            "config1_pa": lambda: gen_config1(True), "config1_pi": lambda: gen_config1(False),
            "networks_forward": gen_networks_forward, "eval_full": gen_eval_full, "sharded2": gen_sharded2}


def main(argv):
    G = torch.load(OUT, weights_only=False) if os.path.exists(OUT) else {}
    G["seeds"], G["nsample"], G["torch"] = SEEDS, NSAMPLE, torch.__version__
    for name in (argv or list(SECTIONS)):
        t0 = time.time()
        G[name] = SECTIONS[name]()
        print("%-18s %.0f s" % (name, time.time() - t0), flush=True)
        torch.save(G, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


def print_margins(seeds):
    """--margins [batch seed ...]: min |y| of the pyramid stages for candidate batch seeds of the eight-shard section (7 s per seed):
    how the seed in SEEDS["sharded8"] was chosen."""
    PS = init_nets("sharded8")[0]
    for seed in seeds or [SEEDS["sharded8"]["batch"]]:
        x, _ = O.synthetic_batch(16, 512, 512, seed=int(seed))
        m = pyramid_margins(PS, x)
        print("batch seed %s: pyramid margins %s%s" % (seed, ["%.1e" % v for v in m],
                                                      "" if all(a >= b for a, b in zip(m, PYRAMID_MARGIN)) else "   (below PYRAMID_MARGIN)"), flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--margins"]:
        print_margins(sys.argv[2:])
    else:
        main(sys.argv[1:])
