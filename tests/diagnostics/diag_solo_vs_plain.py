"""Diagnostic (not collected): the N > 1 form on a one-rank RCCL communicator (SKD_DIST_SOLO=1) against the plain single-rank step,
state after ONE and after TWO steps, and two plain runs against each other (the run-to-run noise of default-mode MIOpen).
    python tests/diagnostics/diag_solo_vs_plain.py"""
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(out, steps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from structure_knowledge_distillation_amd.utils import parallel as P
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    from oracle import step_torch as O
    P.init_distributed()
    dev = torch.device("cuda", 0)
    torch.manual_seed(10)
    model = NetModel(default_args(batch_size=2, ho=True, device=dev, weight_decay=5e-4, lambda_pa=0.5))
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    x, y = O.synthetic_batch(2, 512, 512, seed=3)
    alpha = torch.rand(2, 1, 1, 1, generator=torch.Generator().manual_seed(17))
    snaps = []
    for step in range(steps):
        model.gp_alpha = alpha.to(dev)
        model.set_input((x, y, None, None))
        model.optimize_parameters()
        torch.cuda.synchronize()
        snaps.append({"state": {k: v.detach().cpu().clone() for k, v in model.student.state_dict().items()},
                      "grads": {k: p.grad.detach().cpu().contiguous().clone() for k, p in model.student.named_parameters() if p.grad is not None}})
    torch.save(snaps, out)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    variants = (("plain_a", {"SKD_DIST_SOLO": "0"}), ("plain_b", {"SKD_DIST_SOLO": "0"}), ("plain_dstream0", {"SKD_DIST_SOLO": "0", "SKD_D_STREAM": "0"}),
                ("plain_nograph", {"SKD_DIST_SOLO": "0", "SKD_TEACHER_GRAPH": "0"}),
                ("solo", {"SKD_DIST_SOLO": "1"}), ("solo_collectives", {"SKD_DIST_SOLO": "1", "SKD_SYNC_IPC": "0"}),
                ("solo_fused", {"SKD_DIST_SOLO": "1", "SKD_ABN_SYNC_FUSED": "1"}), ("solo_dstream0", {"SKD_DIST_SOLO": "1", "SKD_D_STREAM": "0"}),
                ("solo_graph", {"SKD_DIST_SOLO": "1", "SKD_TEACHER_GRAPH": "force"}))
    steps = int(os.environ.get("DIAG_STEPS", "1"))
    with tempfile.TemporaryDirectory() as d:
        runs = {}
        for name, env in variants:
            e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", MIOPEN_LOG_LEVEL="3", **env)
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                e.pop(k, None)
            f = os.path.join(d, name + ".pt")
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", f, str(steps)], env=e, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            runs[name] = torch.load(f)
        for x, _ in variants[1:]:
            y = "plain_a"
            for step in range(steps):
                for what in ("grads", "state"):
                    A, B = runs[x][step][what], runs[y][step][what]
                    ds = sorted(((rel(A[k], B[k]), k) for k in A if A[k].dtype.is_floating_point and A[k].numel() > 1 and A[k].dim() != 1),
                                reverse=True)
                    d1 = sorted(((rel(A[k], B[k]), k) for k in A if A[k].dtype.is_floating_point and A[k].numel() > 1 and A[k].dim() == 1),
                                reverse=True)[:2]
                    print("%s vs %s, after step %d, %s: >1-D worst %s best %s | 1-D worst %s"
                          % (x, y, step, what, [("%.1e" % v, k) for v, k in ds[:2]], [("%.1e" % v, k) for v, k in ds[-2:]],
                             [("%.1e" % v, k) for v, k in d1]))
