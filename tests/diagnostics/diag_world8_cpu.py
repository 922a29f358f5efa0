"""CPU diagnostic: the world-W step (gloo + the C-ABI double) with B images per rank against the sharded fp64 oracle, every gradient
tensor's error printed in registration order.    python tests/diagnostics/diag_world8_cpu.py [world] [B] [HW] [ho]"""
import os
import socket
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, world, port, B, HW, ho, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import cref, step_torch as O
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    _lib.install_test_backend(cref.load(_lib.SIGNATURES))
    P.init_distributed("gloo")
    torch.manual_seed(20 + rank)
    model = NetModel(default_args(batch_size=B * world, ho=bool(ho), device=torch.device("cpu"), weight_decay=5e-4, lambda_pa=0.5))
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    snap = lambda mod: {k: v.detach().clone() for k, v in mod.state_dict().items()}
    init, teacher = snap(model.student), snap(model.teacher)
    d_init = snap(model.D_model) if ho else None
    x, y = O.synthetic_batch(B * world, HW, HW, seed=5)
    alpha = torch.rand(B * world, 1, 1, 1, generator=torch.Generator().manual_seed(17))
    sl = slice(rank * B, (rank + 1) * B)
    if ho:
        model.gp_alpha = alpha[sl]
    model.set_input((x[sl], y[sl], None, None))
    model.optimize_parameters()
    out = {"grads": {k: p.grad.clone() for k, p in model.student.named_parameters()},
           "losses": {k: getattr(model, k) for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss")}}
    if rank == 0:
        out.update(init=init, teacher=teacher, d_init=d_init)
    torch.save(out, os.path.join(outdir, "r%d.pt" % rank))
    dist.destroy_process_group()


if __name__ == "__main__":
    world, B, HW, ho = (int(a) for a in (sys.argv[1:] + ["8", "2", "128", "0"][len(sys.argv) - 1:]))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(world, port, B, HW, ho, d), nprocs=world, join=True)
        outs = [torch.load(os.path.join(d, "r%d.pt" % r)) for r in range(world)]
    from oracle import step_torch as O
    dbl = lambda P: None if P is None else {k: v.double() if v.is_floating_point() else v.clone() for k, v in P.items()}
    PS, PT, PD = dbl(outs[0]["init"]), dbl(outs[0]["teacher"]), dbl(outs[0]["d_init"])
    x, y = O.synthetic_batch(B * world, HW, HW, seed=5)
    alpha = torch.rand(B * world, 1, 1, 1, generator=torch.Generator().manual_seed(17)).double()
    shards = [slice(r * B, (r + 1) * B) for r in range(world)]
    cfg = O.StepConfig(ho=bool(ho), weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    want = O.distillation_step_sharded(PS, PT, PD, x.double(), y, cfg, shards, [alpha[sl] for sl in shards] if ho else None)
    for r in range(world):
        print("rank", r, {k: "%.2e" % (abs(outs[r]["losses"][k] - v) / (abs(v) + 1e-30)) for k, v in want["shards"][r].items() if k in outs[r]["losses"]})
    for k, g in want["grads_S"].items():
        if g is None or float(g.norm()) < 1e-12:
            continue
        print("    %-44s err/|g| %.2e  |g| %.3e" % (k, float((outs[0]["grads"][k].double() - g).norm() / g.norm()), float(g.norm())))
