"""-m gpu: the world_size-2 data-parallel path with the REAL HIP kernels.  Two processes share cuda:0 and
talk through gloo (RCCL refuses two ranks on one device; the driver's 2/4/8-GPU runs use backend "nccl"
through exactly the same code): cross-rank InPlaceABNSync statistics (both the in-place leaky-ReLU form
and the fused BN+ReLU form), bucketed gradient averaging, replica consistency of a NetModel step."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", MIOPEN_LOG_LEVEL="3")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = globals()[fn_name](rank, world)
        torch.save(out, os.path.join(outdir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), fn_name, d), nprocs=world, join=True)
        return [torch.load(os.path.join(d, "r%d.pt" % r)) for r in range(world)]


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _sync_abn(rank, world):
    from structure_knowledge_distillation_amd import libs
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 65, 65, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 65, 65, generator=g)
    w, b = torch.randn(6, generator=g), torch.randn(6, generator=g)
    out = {}
    for kind in ("leaky", "relu"):
        mod = libs.InPlaceABNSync(6, activation="leaky_relu" if kind == "leaky" else "none").to(dev).train()
        with torch.no_grad():
            mod.weight.copy_(w); mod.bias.copy_(b)
        xs = x[rank * 2:(rank + 1) * 2].to(dev).requires_grad_(True)
        z = mod(xs * 1.0) if kind == "leaky" else mod.forward_relu(xs * 1.0)
        (z * gz[rank * 2:(rank + 1) * 2].to(dev)).sum().backward()
        out[kind] = {"z": z.detach().cpu(), "dx": xs.grad.cpu(), "dw": mod.weight.grad.cpu(), "db": mod.bias.grad.cpu(),
                     "rm": mod.running_mean.cpu(), "rv": mod.running_var.cpu()}
    return out


def test_sync_abn_two_ranks_hip_kernels():
    from oracle import abn_torch
    outs = _run("_sync_abn")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 65, 65, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 65, 65, generator=g)
    w, b = torch.randn(6, generator=g), torch.randn(6, generator=g)
    for kind in ("leaky", "relu"):
        xo = x.double().requires_grad_(True)
        wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
        rm, rv = torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
        zo = abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, "leaky_relu" if kind == "leaky" else "none", 0.01)
        if kind == "relu":
            zo = torch.relu(zo)
        (zo * gz.double()).sum().backward()
        for r in range(2):
            sl = slice(2 * r, 2 * r + 2)
            assert rel(outs[r][kind]["z"], zo[sl]) < 1e-5, kind
            assert rel(outs[r][kind]["dx"], xo.grad[sl]) < 1e-4, kind
            assert rel(outs[r][kind]["rm"], rm) < 1e-5 and rel(outs[r][kind]["rv"], rv) < 1e-5, kind
        assert rel(outs[0][kind]["dw"] + outs[1][kind]["dw"], wo.grad) < 1e-4, kind
        assert rel(outs[0][kind]["db"] + outs[1][kind]["db"], bo.grad) < 1e-4, kind


def _netmodel_step(rank, world):
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    from oracle import step_torch as O
    dev = torch.device("cuda", 0)
    torch.manual_seed(10 + rank)
    model = NetModel(default_args(batch_size=2, ho=False, device=dev, weight_decay=5e-4, lambda_pa=0.5))
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    x, y = O.synthetic_batch(4, 128, 128, seed=3)
    sl = slice(rank * 2, rank * 2 + 2)
    model.set_input((x[sl], y[sl], None, None))
    model.optimize_parameters()
    return {"after": {k: v.detach().cpu() for k, v in model.student.state_dict().items()},
            "losses": (model.mc_G_loss, model.pi_G_loss, model.pa_G_loss)}


def test_netmodel_step_replicas_stay_identical():
    outs = _run("_netmodel_step")
    for k in outs[0]["after"]:
        assert torch.equal(outs[0]["after"][k], outs[1]["after"][k]), "replicas diverged: %s" % k
    assert outs[0]["losses"] != outs[1]["losses"]      # different shards, different local losses


def test_bench_under_torchrun_two_ranks_over_gloo():
    """The exact command line the driver uses for its multi-GPU runs (python -m torch.distributed.run ... bench.py
    --gpus N), with two ranks sharing cuda:0 over gloo (SKD_DIST_BACKEND): rendezvous, replica broadcast, SyncABN
    collectives in every training BN, bucketed gradient all-reduce from the backward hooks, barrier + max-over-ranks
    timing and the single JSON line from rank 0 are all exercised end to end."""
    import json
    import subprocess
    env = dict(os.environ, SKD_DIST_BACKEND="gloo", MIOPEN_LOG_LEVEL="3")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 4 * 1e3 / out["ms_per_step"]) < 1e-2 * out["value"]
    for k, v in out["config"]["losses_last_step"].items():
        assert v == v and abs(v) < 1e6, (k, v)          # finite
