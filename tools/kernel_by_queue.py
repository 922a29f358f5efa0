#!/usr/bin/env python3
"""Per-queue launch statistics of one kernel from a rocprofv3 kernel trace: where did it run, and how long did a launch take there?

    python tools/kernel_by_queue.py <..._kernel_trace.csv> <name pattern> <out.md>

Written for `python bench.py` in its default configuration, where the frozen teacher runs on its OWN HIP stream beside the student's
forward in the timed steps (SKD_TEACHER_STREAM) and on the MAIN stream in the extra steps bench.py times the roofline kernel in
(bench.py "roofline.measured_in").  A hardware queue per stream: the kernel's launches on the teacher's queue shared the chip with the
student's forward (their durations are stretched by that sharing, while the step as a whole got shorter); its launches on the main queue
ran alone -- those are the ones bench.py brackets with HIP events, and the averages must agree with the line's `roofline.detail`.
The table lists, per queue and grid size (= problem shape), calls / average / min / max, and what else ran on that queue."""
import collections
import csv
import sys


def main():
    trace, pattern, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = list(csv.DictReader(open(trace, newline="")))
    qkey = "Queue_Id"
    per_queue = collections.defaultdict(lambda: [0, 0])
    hits = collections.defaultdict(list)
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        q = r[qkey]
        per_queue[q][0] += 1
        per_queue[q][1] += d
        if pattern in r["Kernel_Name"]:
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
            hits[(q, grid)].append(d)
    main_q = max(per_queue, key=lambda q: per_queue[q][1])
    lines = ["# `%s` by hardware queue (one queue per HIP stream)\n" % pattern,
             "Queues of the trace: " + ", ".join("`q%s`%s: %d kernels, %.1f ms" % (q, " (**main**)" if q == main_q else "", n, t / 1e6)
                                                  for q, (n, t) in sorted(per_queue.items(), key=lambda kv: -kv[1][1])) + "\n",
             "| queue | grid (threads) | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|---|"]
    tot = collections.defaultdict(lambda: [0, 0])
    for (q, grid), ds in sorted(hits.items(), key=lambda kv: (kv[0][0] != main_q, kv[0][0], -sum(kv[1]))):
        lines.append("| q%s%s | %s | %d | %.1f | %.1f | %.1f | %.2f |" % (q, " (main)" if q == main_q else "", grid, len(ds), sum(ds) / len(ds) / 1e3,
                                                                        min(ds) / 1e3, max(ds) / 1e3, sum(ds) / 1e6))
        tot[q][0] += len(ds)
        tot[q][1] += sum(ds)
    lines.append("")
    for q, (n, t) in tot.items():
        lines.append("* q%s%s: %d launches, average %.1f us" % (q, " (main: the teacher issued serially -- what bench.py's HIP events bracket)" if q == main_q
                                                             else " (beside the student's forward)", n, t / n / 1e3))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-(len(tot) + 1):]))


if __name__ == "__main__":
    main()
