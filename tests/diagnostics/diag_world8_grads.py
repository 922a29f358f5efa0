"""GPU diagnostic (round 4): the world-8 step's gradients AT THE STUDENT'S OUTPUTS (logits, deep-supervision logits, post-pyramid
feature) per rank, written to gpurun_out/ so that they can be compared with the sharded oracle's off the GPU box.

    python tests/diagnostics/diag_world8_grads.py [world] [out.pt]
"""
import importlib
import os
import socket
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MIOPEN_LOG_LEVEL="3")
    os.environ.setdefault("SKD_SYNC_TIMEOUT_S", "20")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import test_distributed_gpu as T
        PC = importlib.import_module("structure_knowledge_distillation_amd.networks.pspnet_combine")
        got = {}
        o_fwd = PC.ResNet.forward

        def forward(self, x):
            outs = o_fwd(self, x)
            if self.training and torch.is_grad_enabled():
                for i, name in ((0, "logits"), (1, "dsn"), (2, "feat_psp"), (3, "x4")):
                    got[name] = outs[i].detach().clone()
                    outs[i].register_hook(lambda g, name=name: got.__setitem__("d_" + name, g.detach().clone()))
            return outs

        PC.ResNet.forward = forward
        out = T._netmodel_step_world8(rank, world)
        keep = {"losses": out["losses"]}
        for name in ("logits", "dsn"):
            keep[name] = got[name].contiguous().cpu()
            keep["d_" + name] = got["d_" + name].contiguous().cpu()
        for name in ("feat_psp", "x4"):
            keep[name + "_mean"] = got[name].mean((2, 3)).cpu()
            keep["d_" + name + "_sum"] = got["d_" + name].sum((2, 3)).cpu()
            keep["d_" + name + "_norm"] = float(got["d_" + name].norm())
        keep["d_feat_psp_half"] = got["d_feat_psp"].contiguous().cpu().half()
        keep["grads"] = {k: v for k, v in out["grads"].items() if k.startswith(("pspmodule.stages.0", "head", "dsn"))}
        torch.save(keep, os.path.join(outdir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r04q_world8_grads.pt")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as dd:
        mp.spawn(_worker, args=(world, port, dd), nprocs=world, join=True)
        outs = [torch.load(os.path.join(dd, "r%d.pt" % r)) for r in range(world)]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    torch.save(outs, dst)
    print("wrote", dst, os.path.getsize(dst), "bytes")
