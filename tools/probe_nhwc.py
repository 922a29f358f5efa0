"""GPU probe (not product code): MIOpen find-mode speed of key convolutions with channels_last (NHWC) tensors vs
the NCHW numbers of tools/probe_conv.py -- decides whether an NHWC end-to-end layout (no MIOpen transposes) pays."""
import json
import os
import subprocess
import sys
import time

SHAPES = {
    "t_3x3d2_256_65": (256, 256, 3, 1, 2, 65, 65),
    "s_3x3d4_512_65": (512, 512, 3, 1, 4, 65, 65),
    "stem_3x3_64_64_256": (64, 64, 3, 1, 1, 256, 256),
    "t_1x1_1024_256_65": (1024, 256, 1, 1, 1, 65, 65),
    "l1_3x3_64_129": (64, 64, 3, 1, 1, 129, 129),
}


def child(name, layout, bwd):
    import torch
    import torch.nn.functional as F
    cin, cout, k, s, d, H, W = SHAPES[name]
    torch.backends.cudnn.benchmark = True
    B = 8
    mf = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=mf).requires_grad_(bwd)
    w = torch.randn(cout, cin, k, k, device="cuda").contiguous(memory_format=mf).requires_grad_(bwd)
    pad = d * (k // 2)

    def run():
        y = F.conv2d(x, w, None, s, pad, d)
        if bwd:
            y.backward(torch.ones_like(y))
            x.grad = None
            w.grad = None
        return y
    t0 = time.perf_counter()
    y = run()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    flop = 2.0 * B * cout * cin * k * k * H * W * (3 if bwd else 1)
    print(json.dumps({"shape": name, "layout": layout, "bwd": bwd, "out_is_channels_last": bool(y.is_contiguous(memory_format=torch.channels_last)),
                      "first_s": round(first, 1), "ms": round(ms, 3), "TFLOPs": round(flop / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], sys.argv[3], sys.argv[4] == "1")
        sys.exit(0)
    out = open("gpurun_out/probe_nhwc.log", "a")
    t_start = time.time()
    env = dict(os.environ, PYTORCH_MIOPEN_SUGGEST_NHWC="1", MIOPEN_LOG_LEVEL="3")
    for name in SHAPES:
        for layout in ("nhwc", "nchw"):
            for bwd in (False, True):
                if time.time() - t_start > float(os.environ.get("PROBE_BUDGET_S", "200")):
                    break
                if layout == "nchw" and bwd:
                    continue
                try:
                    r = subprocess.run([sys.executable, __file__, "child", name, layout, "1" if bwd else "0"], env=env,
                                       capture_output=True, text=True, timeout=100)
                    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-300:]
                except subprocess.TimeoutExpired:
                    line = "TIMEOUT %s %s bwd=%s" % (name, layout, bwd)
                out.write(line + "\n")
                out.flush()
