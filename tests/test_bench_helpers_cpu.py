"""bench.py's bookkeeping helpers on the CPU: the JSON line the driver parses must not depend on code paths that only a GPU
box ever executes for the first time."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gemm_summary_and_pmc_traffic_lookup():
    b = _bench()
    recs = [(0.2, (33800, 256, 1024, 0)), (0.2, (33800, 256, 1024, 0)), (0.6, (33800, 512, 2048, 0))]
    g = b.summarise_gemm(recs)
    flops = 2.0 * 33800 * 256 * 1024 * 2 + 2.0 * 33800 * 512 * 2048
    assert g["launches"] == 3 and abs(g["total_ms"] - 1.0) < 1e-9
    assert abs(g["achieved_TFLOPs"] - round(flops / 1e-3 / 1e12, 2)) < 1e-9
    assert set(g["per_shape"]) == {"M=33800 K=256 N=1024", "M=33800 K=512 N=2048"}
    assert g["per_shape"]["M=33800 K=256 N=1024"]["launches"] == 2
    assert b.summarise_gemm([]) is None
    pmc = b.gemm_pmc_traffic()                       # from the committed profiles/*_gemm_lab_pmc.json
    assert pmc is not None and 250.0 < pmc["MB"] < 700.0 and pmc["source"].startswith("profiles/")


def test_abn_summary_modes():
    b = _bench()
    recs = [(0.05, (33800, 1024, 1234, 0)), (0.07, (33800, 1024, 1234, 5678))]      # (rows, C, x, residual): 8 B/elem, 12 with a residual
    s = b.summarise(recs, 8, nhwc="apply")
    want = (8 * 33800 * 1024 * 1.0 + 8 * 33800 * 1024 * 1.5) / (0.12e-3) / 1e9
    assert s["launches"] == 2 and abs(s["achieved_GBs"] - round(want, 1)) < 0.11
    t = b.summarise([(0.03, (33800, 256, 1, 2))], 20, nhwc="train")
    assert abs(t["achieved_GBs"] - round(20 * 33800 * 256 / 0.03e-3 / 1e9, 1)) < 0.11
    assert b.summarise([], 8) is None
