"""GPU probe (not product code): first-call latency and steady-state speed of the fp32 convolutions of
the distillation step under MIOpen immediate mode / MIOpen find mode / PyTorch's rocBLAS im2col path,
and of 1x1 convolutions written as batched GEMMs.  One shape+mode per subprocess; results appended
to gpurun_out/probe_conv.log line by line so partial runs are still useful."""
import json
import os
import subprocess
import sys
import time

SHAPES = {  # name: (Cin, Cout, k, stride, dil, H, W)
    "t_1x1_256_1024_65": (256, 1024, 1, 1, 1, 65, 65),
    "t_1x1_1024_256_65": (1024, 256, 1, 1, 1, 65, 65),
    "t_3x3d2_256_65": (256, 256, 3, 1, 2, 65, 65),
    "s_3x3d4_512_65": (512, 512, 3, 1, 4, 65, 65),
    "stem_3x3_64_64_256": (64, 64, 3, 1, 1, 256, 256),
    "t_ppm_4096_512_65": (4096, 512, 3, 1, 1, 65, 65),
    "l1_3x3_64_129": (64, 64, 3, 1, 1, 129, 129),
}


def child(name, mode, bwd):
    import torch
    import torch.nn.functional as F
    cin, cout, k, s, d, H, W = SHAPES[name]
    if mode == "find":
        torch.backends.cudnn.benchmark = True
    elif mode == "immediate":
        torch.backends.cudnn.benchmark = False
    elif mode in ("rocblas", "gemm"):
        torch.backends.cudnn.enabled = False
    dev = "cuda"
    B = 8
    x = torch.randn(B, cin, H, W, device=dev, requires_grad=bwd)
    w = torch.randn(cout, cin, k, k, device=dev, requires_grad=bwd)
    pad = d * (k // 2)

    def run():
        if mode == "gemm" and k == 1:
            y = torch.matmul(w.view(cout, cin), x.view(B, cin, H * W)).view(B, cout, H, W)
        else:
            y = F.conv2d(x, w, None, s, pad, d)
        if bwd:
            y.backward(torch.ones_like(y))
            x.grad = None
            w.grad = None
        return y
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    flop = 2.0 * B * cout * cin * k * k * (H // s) * (W // s) * (3 if bwd else 1)
    print(json.dumps({"shape": name, "mode": mode, "bwd": bwd, "first_s": round(first, 2), "ms": round(ms, 3),
                      "TFLOPs": round(flop / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], sys.argv[3], sys.argv[4] == "1")
        sys.exit(0)
    out = open("gpurun_out/probe_conv.log", "a")
    budget = float(os.environ.get("PROBE_BUDGET_S", "420"))
    t_start = time.time()
    plan = []
    for name in SHAPES:
        for mode in ("immediate", "rocblas", "gemm", "find"):
            if mode == "gemm" and SHAPES[name][2] != 1:
                continue
            for bwd in (False, True):
                plan.append((name, mode, bwd))
    for name, mode, bwd in plan:
        if time.time() - t_start > budget:
            out.write("budget exhausted\n")
            break
        try:
            r = subprocess.run([sys.executable, __file__, "child", name, mode, "1" if bwd else "0"],
                               capture_output=True, text=True, timeout=120)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR %s %s %s: %s" % (name, mode, bwd, r.stderr[-300:])
        except subprocess.TimeoutExpired:
            line = "TIMEOUT %s %s bwd=%s (>120 s)" % (name, mode, bwd)
        out.write(line + "\n")
        out.flush()
    os.system("du -sh ~/.cache/miopen ~/.config/miopen >> gpurun_out/probe_conv.log 2>&1; find ~/.cache/miopen ~/.config/miopen -type f | head -20 >> gpurun_out/probe_conv.log 2>&1")
