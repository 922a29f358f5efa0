"""GPU tool (not product code): MIOpen's EXHAUSTIVE tuning (MIOPEN_FIND_ENFORCE=SEARCH_DB_UPDATE) for chosen convolution problems of
the step, on top of a copy of the shipped user db -- and a warm timing of the same problems under any db directory.

The shipped find-db (tools/miopen_tune.py) was recorded with MIOpen's ordinary find: every applicable solver timed with its DEFAULT
(heuristic) kernel configuration.  The implicit-GEMM assembly solvers that win almost every problem here are tunable (a list of
tile configurations per direction); the exhaustive search times all of them.  Minutes per problem and direction.

    python tools/miopen_search.py search <out_dir> <problem> [<problem> ...]     # out_dir starts as a copy of the shipped db
    python tools/miopen_search.py time   <db_dir>  <problem> [<problem> ...]     # one JSON line per problem
    problem = cin,cout,k,stride,dilation,H,W,dirs      dirs: any of f (forward), b (backward: data + weights)   batch 8, channels-last
"""
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
SHIPPED = os.path.join(ROOT, "structure_knowledge_distillation_amd", "miopen_db")


def parse(spec):
    cin, cout, k, s, d, H, W, dirs = spec.split(",")
    return int(cin), int(cout), int(k), int(s), int(d), int(H), int(W), dirs


def tensors(cin, cout, k, H, W, B=8):
    import torch
    dev = torch.device("cuda", 0)
    x = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    return x, w


def main():
    mode, db = sys.argv[1], os.path.abspath(sys.argv[2])
    probs = [parse(p) for p in sys.argv[3:]]
    if mode == "search":
        if not os.path.isdir(db):
            shutil.copytree(SHIPPED, db)
        os.environ["MIOPEN_FIND_ENFORCE"] = "4"          # SEARCH_DB_UPDATE: tune even when the find-db has the problem, update both dbs
    os.environ["MIOPEN_USER_DB_PATH"] = db
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(db, "cache")
    os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
    import torch
    import torch.nn.functional as F
    from _timing import warm_timed
    torch.backends.cudnn.benchmark = mode == "search"
    for (cin, cout, k, s, d, H, W, dirs) in probs:
        p = d * (k - 1) // 2
        x, w = tensors(cin, cout, k, H, W)
        OH = (H + 2 * p - d * (k - 1) - 1) // s + 1
        OW = (W + 2 * p - d * (k - 1) - 1) // s + 1
        flop = 2.0 * 8 * cout * cin * k * k * OH * OW
        row = {"problem": "%d->%d k%d s%d d%d %dx%d" % (cin, cout, k, s, d, H, W)}
        if "f" in dirs:
            t0 = time.time()
            with torch.no_grad():
                F.conv2d(x, w, None, s, p, d)
            torch.cuda.synchronize()
            if mode == "search":
                row["fwd_search_s"] = round(time.time() - t0, 1)
            else:
                with torch.no_grad():
                    ms = warm_timed(lambda: F.conv2d(x, w, None, s, p, d), 10)
                row["fwd_us"], row["fwd_frac"] = round(ms * 1e3, 1), round(flop / (ms * 1e-3) / 1e12 / 157.3, 3)
        if "b" in dirs:
            xg, wg = x.clone().requires_grad_(cin != 3), w.clone().requires_grad_(True)
            t0 = time.time()
            y = F.conv2d(xg, wg, None, s, p, d)
            gy = torch.randn_like(y)
            ins = [t for t in (xg, wg) if t.requires_grad]
            torch.autograd.grad(y, ins, gy, retain_graph=True)
            torch.cuda.synchronize()
            if mode == "search":
                row["bwd_search_s"] = round(time.time() - t0, 1)
            else:
                ms = warm_timed(lambda: torch.autograd.grad(y, ins, gy, retain_graph=True), 10)
                row["bwd_us"], row["bwd_frac"] = round(ms * 1e3, 1), round(len(ins) * flop / (ms * 1e-3) / 1e12 / 157.3, 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
