"""TEST INFRASTRUCTURE: ``bench.py`` with the C-ABI double (oracle/libskd_ref.so, host pointers) installed first, so the LAUNCHER
of ``python bench.py --gpus N`` -- self re-execution under torch.distributed.run, rendezvous, warm-up with fallback, max-over-ranks
timing, ONE JSON line from rank 0 -- can be rehearsed on a box without a GPU (tests/test_distributed_cpu.py).  Nothing is measured
here; bench.py itself refuses ``--device cpu`` without an installed double."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.set_num_threads(int(os.environ.get("SKD_TEST_THREADS", "2")))
from oracle import cref  # noqa: E402
from structure_knowledge_distillation_amd import _lib  # noqa: E402

_lib.install_test_backend(cref.load(_lib.SIGNATURES))
os.environ.setdefault("SKD_DIST_BACKEND", "gloo")
os.environ["SKD_BENCH_ENTRY"] = os.path.abspath(__file__)          # bench.self_launch starts the ranks through this wrapper too
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
