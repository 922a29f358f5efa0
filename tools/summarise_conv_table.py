"""conv_table.jsonl (+ the rocprofv3 kernel trace of the same run) -> profiles/<tag>_conv_shapes.md

    python tools/summarise_conv_table.py <conv_table.jsonl> <trace_dir> <out.md>

The kernel trace is cut at the marker launches of tools/conv_table.py (act_kernel<0>, grid = 256 * (index + 1)), so
every row gets the names of the kernels MIOpen actually dispatched for it (solver + layout transposes + zero fills).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    if n.startswith("_ZN2ck"):
        m = re.search(r"kernel_(grouped_conv_[a-z_]+)", n)
        t = re.findall(r"Li(\d+)E", n)[:6]
        return "ck::" + (m.group(1) if m else "kernel") + "<" + ",".join(t) + ">"
    return n.split("(")[0][:100]


def main():
    table, trace_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rows, totals = [], None
    for line in open(table):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        if "totals" in d:
            totals = d["totals"]
        elif "idx" in d:
            rows.append(d)
    kernels = collections.defaultdict(lambda: collections.OrderedDict())
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        disp = sorted(csv.DictReader(open(f, newline="")), key=lambda r: int(r["Dispatch_Id"]))
        cur = None
        for r in disp:
            n = r["Kernel_Name"]
            if "act_kernel<0>" in n:
                cur = int(r["Grid_Size_X"]) // 256 - 1
                continue
            if cur is None:
                continue
            k = short(n)
            e = kernels[cur].setdefault(k, [0, 0.0])
            e[0] += 1
            e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(out, "w") as fh:
        fh.write("# Convolution problems of the batch-8 512x512 step on this MIOpen build (fp32, channels-last, committed find-db)\n\n"
                 "`rocprofv3 --kernel-trace -- python tools/conv_table.py 10`; each problem timed in isolation with HIP events "
                 "(10 repetitions after 3 warm-ups); `frac` = TFLOP/s / 157.3 (fp32 MFMA peak); `x` = calls per step. Kernel column: what MIOpen dispatched "
                 "(share of the traced time for that problem).\n\n")
        if totals:
            fh.write("Totals (sum of count x isolated time): teacher forward %.2f ms (%.1f TFLOP/s), student forward %.2f ms (%.1f TFLOP/s), "
                     "student backward %.2f ms (%.1f TFLOP/s).\n\n" % (totals["teacher_fwd_ms"], totals["teacher_fwd_tflops"], totals["student_fwd_ms"],
                                                                      totals["student_fwd_tflops"], totals["student_bwd_ms"], totals["student_bwd_tflops"]))
        fh.write("| net | problem | x | GFLOP | fwd us | fwd TF/s | frac | bwd us | bwd TF/s | frac | ms/step | kernels |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for d in sorted(rows, key=lambda d: -(d["count"] * (d["fwd_us"] + d.get("bwd_us(dgrad+wgrad)", 0.0)))):
            ks = kernels.get(d["idx"], {})
            tt = sum(v[1] for v in ks.values()) or 1.0
            kdesc = "; ".join("%s (%.0f%%)" % (k, 100.0 * v[1] / tt) for k, v in sorted(ks.items(), key=lambda kv: -kv[1][1])[:4])
            prob = "%d->%d k%d s%d d%d %dx%d%s" % (d["cin"], d["cout"], d["k"], d["stride"], d["dil"], d["in_hw"][0], d["in_hw"][1], " +b" if d["bias"] else "")
            ms = d["count"] * (d["fwd_us"] + d.get("bwd_us(dgrad+wgrad)", 0.0)) / 1e3
            fh.write("| %s | %s | %d | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %.2f | %s |\n" % (
                d["net"], prob, d["count"], d["gflop"], d["fwd_us"], d["fwd_tflops"], d["fwd_frac"],
                d.get("bwd_us(dgrad+wgrad)", ""), d.get("bwd_tflops", ""), d.get("bwd_frac", ""), ms, kdesc))
    print("wrote", out)


if __name__ == "__main__":
    main()
