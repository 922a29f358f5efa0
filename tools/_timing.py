"""Warm timing for the measurement tools (round 6).  A case that follows host work (allocation, a read-back, random-number
generation on the host) starts on a GPU that has dropped its clocks; two or three warm-up launches do not bring them back, and the
case reads 5-20 % slow (profiles/r06_gemm_lab_notes.md: that artefact produced a phantom +21 % in the GEMM lab and the
'same MIOpen problem at two speeds' of profiles/r03z_conv_shapes.md).  ``warm_timed`` launches ``fn`` until >= ``warm_ms`` of GPU
time have passed, then returns the MEDIAN of ``groups`` groups of ``reps`` back-to-back launches (milliseconds per launch)."""
import torch


def warm_timed(fn, reps=10, warm_ms=30.0, groups=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    spent = 0.0
    for _ in range(200):
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        e1.synchronize()
        spent += e0.elapsed_time(e1)
        if spent >= warm_ms:
            break
    t = []
    for _ in range(groups):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        t.append(e0.elapsed_time(e1) / reps)
    t.sort()
    return t[len(t) // 2]
