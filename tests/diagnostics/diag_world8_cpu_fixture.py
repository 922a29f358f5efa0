"""CPU rehearsal of tests/test_distributed_gpu.py::test_netmodel_ho_step_eight_ranks_vs_sharded_oracle: the SAME worker (eight ranks,
two 512 x 512 images each, Pi + Pa + Ho then Pi + Pa), the SAME recorded fixture and the SAME assertions (_check_world8), with the
C-ABI double (oracle/cref: plain C on host pointers, mailboxes in POSIX shared memory) in place of the HIP library and gloo in place of
RCCL.  What it establishes without a GPU: the fixture section, the test's own logic and the whole host path at world 8 and full size.
What it cannot: the HIP kernels (that is what the GPU test is for).  ~5 minutes and ~45 GB on 8 cores.

    python tests/diagnostics/diag_world8_cpu_fixture.py
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


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import cref
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    _lib.install_test_backend(cref.load(_lib.SIGNATURES))
    P.init_distributed("gloo")
    try:
        import test_distributed_gpu as T
        out = T._netmodel_step_world8(rank, world, torch.device("cpu"))
        torch.set_num_threads(1)
        torch.save(out, os.path.join(outdir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    world = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    t0 = time.time()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d), nprocs=world, join=True)
        both = [torch.load(os.path.join(d, "r%d.pt" % r)) for r in range(world)]
    print("eight CPU ranks: %.0f s" % (time.time() - t0), flush=True)
    import test_distributed_gpu as T
    T._check_world8(both, on_gpu=False)
    print("all checks of the world-8 test hold on the CPU double")
