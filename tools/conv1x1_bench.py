"""GPU tool: A/B of the three ways to run the frozen teacher's 1x1 convolution + BatchNorm (+ residual) + ReLU
(VERDICT r01 item 5 / SURVEY.md 8f row 2) on the teacher's real shapes at batch 8:
  (a) MIOpen convolution (tuned find-db) + the fused in-place ABN pass (skd_abn_apply_nhwc)       -- round-1 path
  (b) one fp32-MFMA GEMM with the ABN / residual / ReLU epilogue (csrc/conv1x1.hip)
  (c) torch.ops.aten.miopen_convolution_relu / miopen_convolution_add_relu with the BN folded into weight + bias
      (MIOpen's own fusion; probed, may be unsupported for fp32 NHWC on this build)
    python tools/conv1x1_bench.py [reps]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
SHAPES = [  # (Cin, Cout, HW, count per teacher forward, residual?)
    (256, 1024, 65, 23, True), (1024, 256, 65, 22, False), (512, 2048, 65, 3, True), (2048, 512, 65, 2, False),
    (1024, 512, 65, 1, False), (128, 512, 65, 4, True), (512, 128, 65, 3, False), (64, 256, 129, 3, True),
    (512, 256, 65, 1, False), (256, 128, 129, 1, False), (128, 256, 129, 1, False), (1024, 2048, 65, 1, False), (512, 1024, 65, 1, False),
]


def main():
    import torch
    import structure_knowledge_distillation_amd as _skd
    _skd.configure_miopen()
    from structure_knowledge_distillation_amd import _lib, functional as SF
    import importlib
    IA = importlib.import_module("structure_knowledge_distillation_amd.libs.inplace_abn")
    _lib.load()
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    B = 8
    tot = {"conv_plus_abn_ms": 0.0, "fused_gemm_ms": 0.0}
    for cin, cout, hw, count, has_res in SHAPES:
        x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        res = torch.randn(B, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last) if has_res else None
        conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
        rm, rv = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5
        w, b = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        flop = 2.0 * B * hw * hw * cin * cout

        def timed(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        with torch.no_grad():
            t_conv = timed(lambda: conv(x))
            t_a = timed(lambda: IA.abn_eval_fused(conv(x), w, b, rm, rv, 1e-5, "relu", 0.01, res))
            row = {"cin": cin, "cout": cout, "hw": hw, "count": count, "residual": has_res, "gflop": round(flop / 1e9, 1),
                   "conv_only_us": round(t_conv * 1e3, 1), "conv_only_tflops": round(flop / t_conv / 1e9, 1),
                   "conv_plus_abn_us": round(t_a * 1e3, 1)}
            if SF.conv1x1_abn_supported(x, conv):
                t_b = timed(lambda: SF.conv1x1_abn_eval(x, conv.weight, rm, rv, w, b, 1e-5, "relu", 0.01, res))
                row.update({"fused_gemm_us": round(t_b * 1e3, 1), "fused_gemm_tflops": round(flop / t_b / 1e9, 1),
                            "fused_gemm_frac_fp32_mfma": round(flop / t_b / 1e9 / 157.3, 3), "speedup_vs_conv_plus_abn": round(t_a / t_b, 3)})
                tot["conv_plus_abn_ms"] += count * t_a
                tot["fused_gemm_ms"] += count * t_b
            # (d) the BLAS library on the same GEMM (NHWC 1x1 convolution = (B*H*W, Cin) x (Cin, Cout)), BN folded into
            #     weight / bias; with the bias + ReLU epilogue (torch._addmm_activation) where there is no residual
            s = (w.abs() + 1e-5) / torch.sqrt(rv + 1e-5)
            x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
            w2t = (conv.weight.reshape(cout, cin) * s.view(-1, 1)).t().contiguous()      # (Cin, Cout)
            w2 = (conv.weight.reshape(cout, cin) * s.view(-1, 1)).contiguous()           # (Cout, Cin), used as .t()
            bf = b - rm * s
            try:
                row["blas_mm_us"] = round(timed(lambda: torch.mm(x2, w2.t())) * 1e3, 1)
                row["blas_mm_kn_us"] = round(timed(lambda: torch.mm(x2, w2t)) * 1e3, 1)
                if has_res:
                    r2 = res.permute(0, 2, 3, 1).reshape(-1, cout)
                    row["blas_addmm_res_us"] = round(timed(lambda: torch.addmm(r2, x2, w2.t())) * 1e3, 1)
                else:
                    row["blas_bias_relu_us"] = round(timed(lambda: torch._addmm_activation(bf, x2, w2.t())) * 1e3, 1)
                    want = torch.relu(torch.mm(x2.double(), w2.t().double()) + bf.double())
                    got = torch._addmm_activation(bf, x2, w2.t())
                    row["blas_bias_relu_err"] = float((got.double() - want).abs().max() / want.abs().max())
            except Exception as e:      # noqa: BLE001
                row["blas_error"] = str(e).splitlines()[0][:100]
            if os.environ.get("C11_SKIP_MIOPEN", "0") == "1":
                print(json.dumps(row), flush=True)
                continue
            try:   # MIOpen's own fusion with the BN folded into weight / bias
                s = (w.abs() + 1e-5) / torch.sqrt(rv + 1e-5)
                wf = (conv.weight * s.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
                bf = b - rm * s
                if has_res:
                    fn = lambda: torch.ops.aten.miopen_convolution_add_relu(x, wf, res, 1.0, bf, [1, 1], [0, 0], [1, 1], 1)
                else:
                    fn = lambda: torch.ops.aten.miopen_convolution_relu(x, wf, bf, [1, 1], [0, 0], [1, 1], 1)
                t_c = timed(fn)
                row["miopen_fused_us"] = round(t_c * 1e3, 1)
            except Exception as e:      # noqa: BLE001
                row["miopen_fused_us"] = "unsupported: %s" % str(e).splitlines()[0][:80]
        print(json.dumps(row), flush=True)
    print(json.dumps({"per_teacher_forward": {k: round(v, 3) for k, v in tot.items()}}), flush=True)


if __name__ == "__main__":
    main()
