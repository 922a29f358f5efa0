"""CPU rehearsal of the two multi-GPU tests of tests/test_distributed_gpu.py (backend "nccl", one rank per device: skipped on every
1-GPU box this repository has seen, so their own Python had never executed): the same workers and the same assertions with gloo and
the C-ABI double (oracle/cref) on CPU tensors.  Establishes that the tests themselves are sound; RCCL and hipIpcOpenMemHandle across
devices remain for the first multi-GPU box.

    python tests/diagnostics/diag_multi_gpu_cpu_rehearsal.py [world=2] [plumbing|step|both]
"""
import os
import socket
import sys
import tempfile
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, fn_name, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(max(1, 8 // world))
    from oracle import cref
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    _lib.install_test_backend(cref.load(_lib.SIGNATURES))
    P.init_distributed("gloo")
    try:
        import test_distributed_gpu as T
        torch.save(getattr(T, fn_name)(rank, world, torch.device("cpu")), os.path.join(outdir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def run(fn_name, world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    t0 = time.time()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, fn_name, d), nprocs=world, join=True)
        outs = [torch.load(os.path.join(d, "r%d.pt" % r)) for r in range(world)]
    print("%s at world %d: %.0f s" % (fn_name, world, time.time() - t0), flush=True)
    return outs


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    what = sys.argv[2] if len(sys.argv) > 2 else "both"
    import test_distributed_gpu as T
    if what in ("plumbing", "both"):
        T._check_multi_gpu_plumbing(run("_multi_gpu_plumbing", world), world, cap=256)      # the C double reports 256 workgroups
        print("plumbing checks hold")
    if what in ("step", "both"):
        T._check_multi_gpu_step(run("_multi_gpu_step", world), world)
        print("two-step replica checks hold%s" % (" (incl. the recorded sharded oracle's losses)" if world == 2 else ""))
