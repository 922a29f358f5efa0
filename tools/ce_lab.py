"""GPU lab (not product code): the fused CE kernel (csrc/ce_dsn.hip) built at 2 / 3 / 4 waves per SIMD, timed on the step's shape.

    python tools/ce_lab.py build      # here (hipcc cross-compiles): tools/_ce_lab/libce_w{2,3,4}.so (git-ignored, travel to the GPU box)
    python tools/ce_lab.py            # on the GPU box: one JSON line per variant
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "_ce_lab")
SRC = os.path.join(ROOT, "structure_knowledge_distillation_amd", "csrc")


def build():
    os.makedirs(LAB, exist_ok=True)
    for w in (2, 3, 4):
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-gpu-rdc",
               "-DSKD_CE_WAVES_PER_SIMD=%d" % w, "-I", os.path.join(ROOT, "include"), "-I", SRC,
               os.path.join(SRC, "ce_dsn.hip"), os.path.join(SRC, "status.hip"), "-o", os.path.join(LAB, "libce_w%d.so" % w)]
        subprocess.run(cmd, check=True)
        print("built", cmd[-1])


def main():
    import torch
    dev = torch.device("cuda", 0)
    B, C, h, w, H, W = 8, 19, 65, 65, 512, 512
    g = torch.Generator().manual_seed(0)
    lm, ld = (torch.randn(B, C, h, w, generator=g) * 3).to(dev), (torch.randn(B, C, h, w, generator=g) * 3).to(dev)
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[0, :32] = 255
    y = y.to(dev)
    ref = None
    for wv in (2, 3, 4):
        lib = ctypes.CDLL(os.path.join(LAB, "libce_w%d.so" % wv))
        lib.skd_ce_dsn_workspace_floats.restype = ctypes.c_int64
        n = lib.skd_ce_dsn_workspace_floats(B, C, h, w, H, W)
        ws = torch.empty(max(8, n), device=dev)
        loss, gm, gd = torch.empty(1, device=dev), torch.empty_like(lm), torch.empty_like(ld)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        call = lambda: lib.skd_ce_dsn_forward(B, C, h, w, H, W, P(lm), P(ld), P(y), 255, ctypes.c_float(0.4), P(loss), P(gm), P(gd), P(ws), None)
        for _ in range(3):
            assert call()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for e0, e1 in ev:
            e0.record(); call(); e1.record()
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        if ref is None:
            ref = (float(loss), gm.clone())
        print(json.dumps({"waves_per_simd": wv, "us_median": round(t[15], 1), "us_min": round(t[0], 1), "loss": float(loss),
                          "same_loss_bits": float(loss) == ref[0], "same_grad_bits": bool(torch.equal(gm, ref[1]))}), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else main()
