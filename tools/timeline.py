"""Per-stream timeline of the distillation step from ONE rocprofv3 kernel trace ("measure the step's critical
path": per-stream busy / idle per step, main-stream gaps > 5 us, which D-stream kernels extend past the student's backward).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o bench -- python bench.py --steps 5 --warmup 3 \
        --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing
    python tools/timeline.py <dir>/.../bench_kernel_trace.csv profiles/r05_timeline.md [first_step last_step]

How a step is cut out of the trace without host markers: the D step runs on its own HIP stream (its own hardware queue; it is the only
place ``sn_bwd_*`` kernels run) and the main stream waits for it at the end of every step (kd_model.py: main.wait_stream(side)), so
the END of the D queue's last kernel of a step is an instant at which nothing of that step is left and nothing of the next one has
started: steps are the intervals between those instants (one ``pixelwise_kernel`` launch each).  Queues are identified by what runs
on them, not by their ids: main = the queue with the most kernel time, D = the queue that runs ``sn_bwd``.
"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarise_profiles import category  # noqa: E402

GAP_US = 5.0


def load(path):
    rows = []
    with open(path, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        pick = lambda *names: next((n for n in names if n in cols), None)
        c_name, c_s, c_e = pick("Kernel_Name", "Name"), pick("Start_Timestamp", "Start"), pick("End_Timestamp", "End")
        c_q, c_st = pick("Queue_Id", "Queue"), pick("Stream_Id", "Stream")
        for r in rd:
            q = r.get(c_q, "0") if c_q else "0"
            st = r.get(c_st, "") if c_st else ""
            rows.append((int(r[c_s]), int(r[c_e]), "q%s" % q + ("/s%s" % st if st not in ("", "0") else ""), r[c_name]))
    rows.sort()
    return rows


def union(iv):
    """total length and merged list of [(s, e)] (sorted by s)"""
    out, tot = [], 0
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            if e > out[-1][1]:
                tot += e - out[-1][1]
                out[-1][1] = e
        else:
            out.append([s, e])
            tot += e - s
    return tot, out


def overlap(a, b):
    """total overlap of two merged interval lists"""
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def short(n):
    n = n.replace("void ", "").replace("skd::(anonymous namespace)::", "skd::").replace("at::native::", "")
    return n[:86]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = load(src)
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    busy = {q: sum(e - s for s, e, _, _ in v) for q, v in byq.items()}
    main_q = max(busy, key=busy.get)
    d_q = next((q for q, v in byq.items() if any("sn_bwd" in n for _, _, _, n in v)), None)
    anchors = [s for s, e, q, n in rows if "pixelwise_kernel" in n]
    out = ["# Step timeline from one rocprofv3 kernel trace (`%s`)\n" % os.path.basename(src)]
    out.append("Queues (HIP stream -> hardware queue; identified by what runs on them): " +
               ", ".join("`%s`%s: %d kernels, %.1f ms" % (q, " (**main**)" if q == main_q else (" (**D step**)" if q == d_q else ""), len(v), busy[q] / 1e6)
                         for q, v in sorted(byq.items(), key=lambda kv: -busy[kv[0]])) + "\n")
    if d_q is None or len(anchors) < 3:
        out.append("no D queue / fewer than 3 steps found: nothing to cut\n")
        open(dst, "w").write("\n".join(out))
        return
    d_ends = [e for _, e, _, _ in byq[d_q]]
    bounds = []
    for a in anchors:          # the last D-queue kernel that ended before this step's pixel-wise loss = the end of the previous step
        prev = [e for e in d_ends if e < a]
        bounds.append(max(prev) if prev else rows[0][0])
    steps = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else min(len(steps), lo + 5)
    sel = steps[lo:hi]
    out.append("%d steps in the trace; steps %d..%d (the timed region of `bench.py --warmup 3 --steps 5`: teacher = one hipGraph replay) are summarised below.\n"
               % (len(steps), lo, hi - 1))
    out.append("| step | wall ms | main busy | main idle | D busy | both busy | device idle (no queue busy) | D tail after main's last kernel | launches |")
    out.append("|---|---|---|---|---|---|---|---|---|")
    agg = collections.Counter()
    gaps_all, cat_main, cat_d, tails = [], collections.Counter(), collections.Counter(), collections.Counter()
    stock, stock_n, stock_ctx = collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)
    for k, (b0, b1) in enumerate(sel):
        inside = [r for r in rows if r[0] >= b0 and r[1] <= b1 + 1]
        mi = [(s, e) for s, e, q, _ in inside if q == main_q]
        di = [(s, e) for s, e, q, _ in inside if q == d_q]
        oi = [(s, e) for s, e, q, _ in inside if q not in (main_q, d_q)]
        tm, um = union(mi)
        td, ud = union(di)
        ta, ua = union(mi + di + oi)
        both = overlap(um, ud)
        wall = b1 - b0
        main_last = max(e for s, e in mi)
        d_last = max(e for s, e in di) if di else main_last
        tail = max(0, d_last - main_last)
        out.append("| %d | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %d |" % (lo + k, wall / 1e6, tm / 1e6, (wall - tm) / 1e6, td / 1e6, both / 1e6,
                                                                            (wall - ta) / 1e6, tail / 1e6, len(inside)))
        for key, v in (("wall", wall), ("main", tm), ("d", td), ("both", both), ("idle", wall - ta), ("tail", tail), ("n", len(inside))):
            agg[key] += v
        # main-stream gaps
        ms = sorted((s, e, n) for s, e, q, n in inside if q == main_q)
        cur_end, cur_name = b0, "(step start: end of the previous step's D stream)"
        for s, e, n in ms:
            if s - cur_end > GAP_US * 1e3:
                d_busy = overlap([[cur_end, s]], ud)
                gaps_all.append((s - cur_end, d_busy, cur_name, n, lo + k, (cur_end - b0) / 1e6))
            if e > cur_end:
                cur_end, cur_name = e, n
        if b1 - cur_end > GAP_US * 1e3:
            gaps_all.append((b1 - cur_end, overlap([[cur_end, b1]], ud), cur_name, "(step end: D stream's last kernel)", lo + k, (cur_end - b0) / 1e6))
        # round 6: the STOCK kernels on the main stream by name, each with the library / hand-written kernel that
        # follows it most often (the consumer names the producer: a zero-fill in front of a split-K wrw solver, a gradient add in
        # front of an ABN backward ...)
        for i, (s, e, n) in enumerate(ms):
            c = category(n)
            if c.startswith(("torch", "MIOpen layout")):
                key = short(n)[:70]
                stock[key] += e - s
                stock_n[key] += 1
                nxt = next((short(n2)[:60] for _, _, n2 in ms[i + 1:i + 6] if not category(n2).startswith(("torch", "MIOpen layout"))), "(none within 5)")
                stock_ctx[key][nxt] += 1
        for s, e, q, n in inside:
            (cat_main if q == main_q else cat_d)[category(n)] += e - s
            if q == d_q and e > main_last:
                tails[short(n)] += min(e, d_last) - max(s, main_last)
    n = len(sel)
    out.append("| **mean** | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %d |\n" % (agg["wall"] / n / 1e6, agg["main"] / n / 1e6, (agg["wall"] - agg["main"]) / n / 1e6,
                                                                                 agg["d"] / n / 1e6, agg["both"] / n / 1e6, agg["idle"] / n / 1e6, agg["tail"] / n / 1e6, agg["n"] // n))
    out.append("Reading: the step's critical path is the main stream's busy time + its idle time; `D tail` is the part of the D step that is NOT hidden behind the "
               "student's backward (the main stream has nothing left and waits for the D stream before the next step may start).\n")
    out.append("## Main stream, kernel time per step by category\n\n| category | ms / step |\n|---|---|")
    for c, v in cat_main.most_common():
        out.append("| %s | %.2f |" % (c, v / n / 1e6))
    out.append("\n## Main stream: stock (torch / MIOpen tensor-op) kernels by name\n\n%.2f ms per step in %d launches.\n\n| kernel | launches / step | ms / step | "
               "next non-stock kernel (most frequent) |\n|---|---|---|---|" % (sum(stock.values()) / n / 1e6, sum(stock_n.values()) // n))
    for key, v in stock.most_common(30):
        ctx = ", ".join("`%s` x%d" % (a, b // n) for a, b in stock_ctx[key].most_common(3))
        out.append("| `%s` | %.1f | %.3f | %s |" % (key, stock_n[key] / n, v / n / 1e6, ctx))
    out.append("\n## D stream, kernel time per step by category\n\n| category | ms / step |\n|---|---|")
    for c, v in cat_d.most_common():
        out.append("| %s | %.2f |" % (c, v / n / 1e6))
    tot_gap = sum(g[0] for g in gaps_all)
    host_gap = sum(g[0] - g[1] for g in gaps_all)
    out.append("\n## Main-stream gaps > %.0f us\n\n%d gaps per step, %.2f ms per step in total, of which %.2f ms with the D stream idle as well (nothing running: the host "
               "was behind, or a cross-stream wait).  Largest (one line per gap, all selected steps):\n" % (GAP_US, len(gaps_all) // n, tot_gap / n / 1e6, host_gap / n / 1e6))
    out.append("| gap us | D busy us | step | at ms | after | before |\n|---|---|---|---|---|---|")
    for g, db, a, b, st, at in sorted(gaps_all, reverse=True)[:40]:
        out.append("| %.0f | %.0f | %d | %.2f | `%s` | `%s` |" % (g / 1e3, db / 1e3, st, at, short(a), short(b)))
    out.append("\n## D-stream kernels running after the main stream's last kernel (the exposed tail), ms per step\n\n| kernel | ms / step |\n|---|---|")
    for kname, v in tails.most_common(20):
        out.append("| `%s` | %.3f |" % (kname, v / n / 1e6))
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:14]))


if __name__ == "__main__":
    main()
