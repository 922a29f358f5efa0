"""GPU tool: per-shape convolution table of the batch-8 512x512 step on this MIOpen build.

    python tools/conv_table.py [reps] > gpurun_out/conv_table.jsonl        (under rocprofv3 --kernel-trace to get solver names)

Collects every nn.Conv2d call of one teacher forward and one student forward (module hooks -> unique problem
configurations with their call counts), then times each configuration in isolation with HIP events exactly as the
step issues it (fp32, channels-last, batch 8, immediate mode against the committed find-db): forward for both
networks, backward-data + backward-weights for the student.  One JSON line per configuration with us, TFLOP/s and
the fraction of the fp32 MFMA peak (157.3 TFLOP/s); a final line totals count x time per direction.
Every configuration is preceded by a marker launch (skd_leaky_relu, grid = index + 1 workgroups) so that
tools/summarise_conv_table.py can attach the kernel names of a rocprofv3 kernel trace to the rows.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import warm_timed  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 157.3


def main():
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    lib = _lib.load()
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    B = 8
    torch.manual_seed(0)
    model = NetModel(default_args(batch_size=B, device=dev, weight_decay=5e-4, lambda_pa=0.5))
    found = {}

    def hook(net):
        def h(mod, inp, out):
            x = inp[0]
            key = (net, mod.in_channels, mod.out_channels, mod.kernel_size[0], mod.stride[0], mod.padding[0], mod.dilation[0],
                   x.shape[2], x.shape[3], mod.bias is not None, x.requires_grad or net == "student")
            found[key] = found.get(key, 0) + 1
        return h

    handles = []
    for net, m in (("teacher", model.teacher), ("student", model.student)):
        for mod in m.modules():
            if isinstance(mod, nn.Conv2d):
                handles.append(mod.register_forward_hook(hook(net)))
    x = (torch.randn(B, 3, 512, 512, device=dev) * 57).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        model.teacher.eval()(x)
    model.student.train()(x)
    for h in handles:
        h.remove()
    if True:      # (the fold is the only form since round 5)
        # the folded PSP bottleneck calls F.conv2d on the feature-map half of the weight directly (no module hook fires):
        # those two problems replace the 4096 -> 512 / 1024 -> 128 rows of the concatenate-then-convolve form
        hw = x.shape[2] // 8 + 1
        found[("teacher", 2048, 512, 3, 1, 1, 1, hw, hw, False, False)] = 1
        found[("student", 512, 128, 3, 1, 1, 1, hw, hw, False, True)] = 1
    del model
    torch.cuda.empty_cache()
    st = torch.cuda.current_stream().cuda_stream
    marker = torch.zeros(256 * (len(found) + 2), device=dev)
    tot = {"teacher_fwd_ms": 0.0, "student_fwd_ms": 0.0, "student_bwd_ms": 0.0, "teacher_fwd_tflop": 0.0, "student_fwd_tflop": 0.0}
    for idx, (key, count) in enumerate(sorted(found.items(), key=lambda kv: (kv[0][0], -kv[0][1] * kv[0][2] * kv[0][3] ** 2 * kv[0][7] * kv[0][8]))):
        net, cin, cout, k, s, p, d, H, W, bias, _ = key
        xin = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        bvec = torch.randn(cout, device=dev) if bias else None
        OH = (H + 2 * p - d * (k - 1) - 1) // s + 1
        OW = (W + 2 * p - d * (k - 1) - 1) // s + 1
        flop = 2.0 * B * cout * cin * k * k * OH * OW
        assert lib.skd_leaky_relu(256 * (idx + 1), marker.data_ptr(), 1.0, st)

        def timed(fn):
            # round 6: >= 30 ms of warm-up and the median of five groups (tools/_timing.py): with three warm-up launches the FIRST
            # problem of the table read 15 % slow (the 'same problem, two speeds' of profiles/r03z: student 1333 us vs teacher 1153 us
            # for the identical 512 -> 512 d4 call -- the student row is simply measured first)
            return warm_timed(fn, reps)

        with torch.no_grad():
            fwd_ms = timed(lambda: F.conv2d(xin, w, bvec, s, p, d))
        row = {"idx": idx, "net": net, "cin": cin, "cout": cout, "k": k, "stride": s, "pad": p, "dil": d, "in_hw": [H, W],
               "out_hw": [OH, OW], "bias": bias, "count": count, "gflop": round(flop / 1e9, 2), "fwd_us": round(fwd_ms * 1e3, 1),
               "fwd_tflops": round(flop / (fwd_ms * 1e-3) / 1e12, 1), "fwd_frac": round(flop / (fwd_ms * 1e-3) / 1e12 / PEAK, 3)}
        tot[net + "_fwd_ms"] += count * fwd_ms
        tot[net + "_fwd_tflop"] += count * flop / 1e12
        if net == "student":
            xg = xin.clone().requires_grad_(cin != 3)
            wg = w.clone().requires_grad_(True)
            y = F.conv2d(xg, wg, bvec, s, p, d)
            gy = torch.randn_like(y)
            ins = [t for t in (xg, wg) if t.requires_grad]
            bwd_ms = timed(lambda: torch.autograd.grad(y, ins, gy, retain_graph=True))
            nb = len(ins)
            row.update({"bwd_us(dgrad+wgrad)": round(bwd_ms * 1e3, 1), "bwd_tflops": round(nb * flop / (bwd_ms * 1e-3) / 1e12, 1),
                        "bwd_frac": round(nb * flop / (bwd_ms * 1e-3) / 1e12 / PEAK, 3)})
            tot["student_bwd_ms"] += count * bwd_ms
        print(json.dumps(row), flush=True)
    tot = {k: round(v, 3) for k, v in tot.items()}
    tot["teacher_fwd_tflops"] = round(tot["teacher_fwd_tflop"] / (tot["teacher_fwd_ms"] * 1e-3), 1)
    tot["student_fwd_tflops"] = round(tot["student_fwd_tflop"] / (tot["student_fwd_ms"] * 1e-3), 1)
    tot["student_bwd_tflops"] = round(2 * tot["student_fwd_tflop"] / (tot["student_bwd_ms"] * 1e-3), 1)
    print(json.dumps({"totals": tot}), flush=True)


if __name__ == "__main__":
    main()
