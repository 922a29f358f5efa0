"""GPU tool: InPlace-ABN kernels alone on the step's representative shapes, timed with HIP events on the
launch stream.  Used for kernel tuning and as the short command profiled by rocprofv3 (--kernel-trace
--stats, and separate --pmc FETCH_SIZE / WRITE_SIZE passes) for profiles/.

    python tools/abn_microbench.py [reps]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # (N, C, S) at batch 8, 512x512 input
    (8, 256, 4225), (8, 1024, 4225), (8, 512, 4225), (8, 2048, 4225),   # teacher layer3/4 (eval apply)
    (8, 64, 65536), (8, 128, 65536), (8, 64, 16641), (8, 128, 4225),    # student stem / layer1 / layer2
    (8, 256, 16641),                                                     # teacher layer1 (256 x 129^2)
]


def main():
    import torch
    from structure_knowledge_distillation_amd import _lib
    lib = _lib.load()
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for (N, C, S) in SHAPES:
        x = torch.randn(N, C, S, device=dev)
        r = torch.randn(N, C, S, device=dev)
        dz = torch.randn(N, C, S, device=dev)
        dx = torch.empty_like(x)
        w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        m, v = torch.empty(C, device=dev), torch.empty(C, device=dev)
        e, ey = torch.empty(C, device=dev), torch.empty(C, device=dev)
        dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = torch.empty(lib.skd_abn_workspace_floats(N, C, S), device=dev)
        p = lambda t: t.data_ptr()
        calls = {
            "apply_eval(8B)": (8, lambda: lib.skd_abn_apply(N, C, S, p(x), p(rm), p(rv), p(w), p(b), 1e-5, 3, 0.01, st)),
            "apply_nhwc(8B)": (8, lambda: lib.skd_abn_apply_nhwc(N * S, C, p(x), None, p(rm), p(rv), p(w), p(b), 1e-5, 3, 0.01, st)),
            "apply_nhwc_residual(12B)": (12, lambda: lib.skd_abn_apply_nhwc(N * S, C, p(x), p(r), p(rm), p(rv), p(w), p(b), 1e-5, 3, 0.01, st)),
            "apply_residual(12B)": (12, lambda: lib.skd_abn_apply_residual(N, C, S, p(x), p(r), p(rm), p(rv), p(w), p(b), 1e-5, 3, 0.01, st)),
            "stats(4B)": (4, lambda: lib.skd_abn_stats(N, C, S, p(x), p(m), p(v), p(ws), st)),
            "forward_train(12B)": (12, lambda: lib.skd_abn_forward_train(N, C, S, p(x), p(w), p(b), p(rm), p(rv), p(m), p(v), 0.1, 1e-5, 0, 0.01, p(ws), st)),
            "backward(20B)": (20, lambda: lib.skd_abn_backward(N, C, S, p(x), p(dz), p(v), p(w), p(b), p(e), p(ey), p(dx), p(dw), p(db), 1e-5, 0, 0.01, 1, p(ws), st)),
        }
        row = {}
        for name, (bpe, fn) in calls.items():
            for _ in range(3):
                assert fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row[name] = {"us": round(ms * 1e3, 1), "GBs": round(bpe * N * C * S / (ms * 1e-3) / 1e9, 0)}
        out["%dx%dx%d" % (N, C, S)] = row
        print(json.dumps({"%dx%dx%d" % (N, C, S): row}), flush=True)


def cold():
    """HBM-resident behaviour: cycle over enough distinct tensors (> 1 GB in total) that neither the 32 MiB of
    L2 nor the 256 MiB Infinity Cache can hold the working set between two visits of the same tensor."""
    import torch
    from structure_knowledge_distillation_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for (N, C, S) in ((8, 512, 4225), (8, 2048, 4225), (8, 64, 65536)):
        n = max(2, int(1.5e9 // (N * C * S * 4)) + 1)
        xs = [torch.randn(N, C, S, device=dev) for _ in range(n)]
        rs = [torch.randn(N, C, S, device=dev) for _ in range(2)]
        w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        m, v = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(lib.skd_abn_workspace_floats(N, C, S), device=dev)
        p = lambda t: t.data_ptr()
        row = {}
        ys = [torch.empty_like(xs[0]) for _ in range(2)]
        for name, bpe, fn in (
                ("torch copy_(8B)", 8, lambda x: ys[0].copy_(x)),
                ("torch relu_(8B)", 8, lambda x: torch.relu_(x)),
                ("apply_eval(8B)", 8, lambda x: lib.skd_abn_apply(N, C, S, p(x), p(rm), p(rv), p(w), p(b), 1e-5, 3, 0.01, st)),
                ("stats(4B)", 4, lambda x: lib.skd_abn_stats(N, C, S, p(x), p(m), p(v), p(ws), st)),
                ("forward_train(12B)", 12, lambda x: lib.skd_abn_forward_train(N, C, S, p(x), p(w), p(b), None, None, p(m), p(v), 0.1, 1e-5, 0, 0.01, p(ws), st))):
            for x in xs:
                fn(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                for x in xs:
                    fn(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / (2 * n)
            row[name] = {"us": round(ms * 1e3, 1), "GBs": round(bpe * N * C * S / (ms * 1e-3) / 1e9, 0)}
        print(json.dumps({"cold %dx%dx%d (x%d tensors)" % (N, C, S, n): row}), flush=True)
        del xs, rs


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cold":
        cold()
    else:
        main()
