"""tools/timeline.py on a synthetic rocprofv3 kernel trace: the step cutting (end of the D queue's last kernel), the per-stream busy /
idle accounting, main-stream gaps and the exposed D tail -- the arithmetic DESIGN.md section 9.5's critical-path statement rests on."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, steps=6):
    """Per step (1 ms apart, times in ns): main queue 1: teacher 0-300 us, pixel-wise loss 300-310, backward 310-700 with a 40 us hole
    at 500; D queue 4: sn_bwd + conv 350-650 (overlapping the backward) and a tail 700-760 after the main stream's last kernel."""
    rows = []
    for k in range(steps):
        t = k * 1_000_000
        rows += [(1, "igemm_fwd_teacher", t + 0, t + 300_000), (1, "void skd::(anonymous namespace)::pixelwise_kernel(float const*)", t + 300_000, t + 310_000),
                 (1, "igemm_bwd_a", t + 310_000, t + 500_000), (1, "igemm_wrw_b", t + 540_000, t + 700_000),
                 (4, "skd::(anonymous namespace)::sn_bwd_dot_multi_kernel(skd::SnBatch)", t + 350_000, t + 400_000), (4, "igemm_fwd_d", t + 400_000, t + 650_000),
                 (4, "void at::native::(anonymous namespace)::multi_tensor_apply_kernel<x>", t + 700_000, t + 760_000)]
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kind", "Queue_Id", "Stream_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for q, n, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", q, 0, n, s, e])


def test_timeline_cuts_steps_and_accounts_streams(tmp_path):
    src, dst = tmp_path / "bench_kernel_trace.csv", tmp_path / "timeline.md"
    _trace(str(src))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), str(src), str(dst), "1", "4"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    text = dst.read_text()
    assert "`q1` (**main**)" in text and "`q4` (**D step**)" in text
    mean = [l for l in text.splitlines() if l.startswith("| **mean**")][0].split("|")
    wall, main_busy, main_idle, d_busy, both, idle, tail = [float(x) for x in mean[2:9]]
    assert abs(wall - 1.0) < 1e-6                                     # a step = the interval between two ends of the D queue
    assert abs(main_busy - 0.66) < 1e-6 and abs(main_idle - 0.34) < 1e-6
    assert abs(d_busy - 0.36) < 1e-6 and abs(both - 0.26) < 1e-6      # 350-500 and 540-650 overlap the backward
    assert abs(tail - 0.06) < 1e-6                                    # the D stream's last 60 us are exposed
    assert abs(idle - 0.24) < 1e-6                                    # the 240 us between the steps (the 40 us hole of the main stream is covered by the D stream)
    gaps = [l for l in text.splitlines() if l.startswith("| 40 |") or l.startswith("| 240 |")]
    assert any("igemm_bwd_a" in g and "igemm_wrw_b" in g for g in gaps), text      # the 40 us hole, with its neighbours named
