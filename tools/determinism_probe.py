"""GPU tool: is the step bit-reproducible, and what does SKD_DETERMINISTIC=1 cost?

    python tools/determinism_probe.py [batch] > gpurun_out/determinism.jsonl

For each mode (default; deterministic = NetModel under SKD_DETERMINISTIC=1) one NetModel is built from the same seed and the
SAME step (same weights, buffers, inputs, WGAN-GP alpha; Dropout off) is executed three times from a restored state:
per-tensor max relative difference of the student / discriminator gradients between the runs (0 = bit-equal), the tensors
that differ, and the step time over 6 steps.  One JSON line per mode."""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_mode(det, B):
    import torch
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    os.environ["SKD_DETERMINISTIC"] = "1" if det else "0"
    torch.backends.cudnn.enabled = True
    torch.use_deterministic_algorithms(False)
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        model = NetModel(default_args(batch_size=B, device=dev, weight_decay=5e-4, lambda_pa=0.5))
        for m in model.student.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        with torch.no_grad():
            model.D_model.attn1.gamma.fill_(0.25)
            model.D_model.attn2.gamma.fill_(-0.5)
        snap = lambda mod: {k: v.detach().clone() for k, v in mod.state_dict().items()}
        S0, D0 = snap(model.student), snap(model.D_model)
        gen = torch.Generator().manual_seed(0)       # SURVEY.md 8d synthetic inputs (a tool may not import oracle/)
        images = torch.randn(B, 3, 512, 512, generator=gen) * 57.0
        labels = torch.randint(0, 19, (B, 512, 512), generator=gen)
        labels[0, :32] = 255
        alpha = torch.rand(B, 1, 1, 1, generator=torch.Generator().manual_seed(7)).to(dev)
        runs = []
        for rep in range(3):
            model.student.load_state_dict(S0)
            model.D_model.load_state_dict(D0)
            model.G_solver.state.clear()
            model.D_solver.state.clear()
            model.gp_alpha = alpha
            model.set_input((images, labels, None, None))
            model.optimize_parameters()
            torch.cuda.synchronize()
            g = {"S." + k: p.grad.detach().clone() for k, p in model.student.named_parameters()}
            g.update({"D." + k: p.grad.detach().clone() for k, p in model.D_model.named_parameters() if p.grad is not None})
            runs.append((g, [model.G_loss, model.mc_G_loss, model.pi_G_loss, model.pa_G_loss, model.D_loss]))
        worst, differing = 0.0, []
        for k in runs[0][0]:
            a = runs[0][0][k].double()
            d = max(float((runs[r][0][k].double() - a).norm() / (a.norm() + 1e-30)) for r in (1, 2))
            if d > 0:
                differing.append((k, d))
            worst = max(worst, d)
        differing.sort(key=lambda kv: -kv[1])
        for i in range(3):
            model.set_input((images, labels, None, None))
            model.optimize_parameters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(6):
            model.set_input((images, labels, None, None))
            model.optimize_parameters()
            _ = model.G_loss
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
    msgs = sorted({str(w.message)[:160] for w in wlist if "deterministic" in str(w.message).lower()})
    torch.backends.cudnn.enabled = True
    torch.use_deterministic_algorithms(False)
    return {"mode": "deterministic" if det else "default", "batch": B, "ms_per_step": round(ms, 2),
            "images_per_s": round(B / ms * 1e3, 2), "tensors": len(runs[0][0]), "tensors_differing_between_runs": len(differing),
            "worst_rel_diff_between_runs": worst, "top_differing": [(k, float("%.3g" % d)) for k, d in differing[:8]],
            "losses_run0": runs[0][1], "losses_bit_equal": all(runs[r][1] == runs[0][1] for r in (1, 2)),
            "nondeterministic_op_warnings": msgs}


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for det in (False, True):
        print(json.dumps(run_mode(det, B)), flush=True)
