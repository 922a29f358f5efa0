"""Turn the rocprofv3 outputs merged into gpurun_out/ into the small summaries committed under profiles/.

    python tools/summarise_profiles.py <tag> [steps] [session]     e.g. r02b 8 s4  (session = sub-directory of gpurun_out/
                                                                   written by tools/gpu_session.sh; default: gpurun_out/ itself)

Writes profiles/<tag>_bench_kernel_stats.csv (verbatim rocprofv3 --stats table), profiles/<tag>_step_breakdown.md
(per-category ms/step) and profiles/<tag>_abn_pmc.json (FETCH_SIZE / WRITE_SIZE per ABN kernel launch; FETCH_SIZE is
doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, units are KB)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def category(n):
    if "skd::" in n:
        for key, name in (("conv1x1", "skd fused bottleneck-tail GEMM (conv1x1 + ABN)"), ("sync_", "skd SyncABN mailbox exchange"), ("abn", "skd InPlace-ABN"), ("maxpool3x3s2", "skd stem max-pool"), ("ppm_fold", "skd PSP bottleneck fold"), ("gram", "skd pair-wise"), ("pairwise", "skd pair-wise"), ("maxpool", "skd pair-wise"),
                          ("maxunpool", "skd pair-wise"), ("l2_norm", "skd pair-wise"), ("ce_", "skd CE+upsample (DSN)"),
                          ("ppm", "skd pyramid pooling"), ("pixelwise", "skd pixel-wise"), ("sn_", "skd spectral norm"), ("head_", "skd classifier heads")):
            if key in n:
                return name
        return "skd other"
    if n.startswith(("igemm_", "miopenSp3AsmConv", "Cijk_", "naive_conv", "gcnAsmConv", "MIOpenConv")) or "Conv" in n or "grouped_conv" in n or "_ZN2ck" in n:
        return "convolutions / GEMMs (MIOpen, rocBLAS)"
    if "batched_transpose" in n or "SubTensorOp" in n:
        return "MIOpen layout transposes / tensor ops"
    if "upsample_bilinear2d_backward" in n:
        return "torch upsample backward"
    if "upsample_bilinear2d" in n:
        return "torch upsample forward"
    if "adaptive_average" in n:
        return "torch adaptive avg pool"
    if "nll_loss" in n or "SoftMax" in n:
        return "torch CE pieces"
    if "max_pool" in n:
        return "torch max-pool"
    return "torch element-wise / other"


def main():
    global G
    tag = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    if len(sys.argv) > 3:
        G = os.path.join(G, sys.argv[3])
    os.makedirs(P, exist_ok=True)
    src = os.path.join(G, "prof_bench", "bench_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, tag + "_bench_kernel_stats.csv"))
        rows = list(csv.DictReader(open(src)))
        tot = sum(int(r["TotalDurationNs"]) for r in rows)
        agg, calls = collections.Counter(), collections.Counter()
        for r in rows:
            agg[category(r["Name"])] += int(r["TotalDurationNs"])
            calls[category(r["Name"])] += int(r["Calls"])
        with open(os.path.join(P, tag + "_step_breakdown.md"), "w") as fh:
            fh.write("# %s: kernel time per step by category\n\n`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline "
                     "--no-pairwise-sweep` (%d steps in the trace: 3 warm-up + 5 timed + -- with the teacher hipGraph on -- 1 + 5 eager-teacher steps for the roofline kernel + 3 kernel-timing steps; "
                     "kernels replayed from the teacher's hipGraph appear as ordinary dispatches)\n\n| category | ms / step | %% | launches / step |\n|---|---|---|---|\n" % (tag, steps))
            for k, v in agg.most_common():
                fh.write("| %s | %.2f | %.1f | %d |\n" % (k, v / 1e6 / steps, 100.0 * v / tot, calls[k] // steps))
            fh.write("| **total** | %.2f | 100 | %d |\n\nTop kernels:\n\n| kernel | calls | avg us | %% |\n|---|---|---|---|\n" % (tot / 1e6 / steps, sum(calls.values()) // steps))
            for r in rows[:25]:
                fh.write("| `%s` | %s | %.1f | %s |\n" % (r["Name"][:110].replace("|", "/"), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    src = os.path.join(G, "prof_abn", "abn_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, tag + "_abn_microbench_kernel_stats.csv"))
    pmc = {}
    for name, sub in (("FETCH_SIZE_KB", "pmc_fetch"), ("WRITE_SIZE_KB", "pmc_write")):
        path = os.path.join(G, sub, "abn_counter_collection.csv")
        if not os.path.exists(path):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            if "skd::" in n:
                key = n.split("skd::(anonymous namespace)::")[1].split("(")[0]
                acc[(key, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        for (k, grid), v in acc.items():
            pmc.setdefault("%s grid=%d" % (k, grid), {})[name] = round(sum(v) / len(v), 1)
    # per kernel name over ALL its launches of the microbench (every shape contributes the same number of
    # launches, so the averages pair with the mean element count of tools/abn_microbench.SHAPES)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from abn_microbench import SHAPES
        mean_elems = sum(n * c * s_ for n, c, s_ in SHAPES) / float(len(SHAPES))
    except Exception:
        mean_elems = None
    byname = {}
    for name, sub in (("FETCH_SIZE_KB", "pmc_fetch"), ("WRITE_SIZE_KB", "pmc_write")):
        path = os.path.join(G, sub, "abn_counter_collection.csv")
        if not os.path.exists(path):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            if "skd::" in n:
                acc[n.split("skd::(anonymous namespace)::")[1].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            byname.setdefault("ALL LAUNCHES " + k, {})[name] = round(sum(v) / len(v), 1)
    for k, d in byname.items():
        d["launch_mean_elems"] = mean_elems
        if mean_elems and "nhwc" not in k and "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
            bpe = 12 if "residual" in k else (8 if "apply_kernel" in k else None)
            if bpe:
                d["hbm_over_algorithmic"] = round((2 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1e3 / (bpe * mean_elems), 4)
    # the channels-last apply kernel: 32 elements per launched thread, so every launch's algorithmic bytes follow
    # from its Grid_Size and the ratio can be summed launch by launch (template arg 2 = has residual -> 12 B/elem)
    sums = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for idx, sub in ((0, "pmc_fetch"), (1, "pmc_write")):
        path = os.path.join(G, sub, "abn_counter_collection.csv")
        if not os.path.exists(path):
            continue
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            if "abn_apply_nhwc_kernel<" in n:
                key = n.split("skd::(anonymous namespace)::")[1].split("(")[0]
                sums[key][idx] += float(r["Counter_Value"])
                if idx == 0:
                    sums[key][2] += float(r["Grid_Size"])
    for key, (f, w, threads) in sums.items():
        bpe = 12 if ", true," in key else 8
        d = byname.setdefault("ALL LAUNCHES " + key, {})
        d.pop("launch_mean_elems", None)
        d["algorithmic_MB_sum"] = round(bpe * 32 * threads / 1e6, 1)
        d["hbm_MB_sum (2*FETCH + WRITE)"] = round((2 * f + w) / 1e3, 1)
        d["hbm_over_algorithmic"] = round((2 * f + w) * 1e3 / (bpe * 32 * threads), 4) if threads else None
    pmc.update(byname)
    if pmc:
        for k, d in pmc.items():
            if "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
                d["hbm_MB (2*FETCH + WRITE)"] = round((2 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) / 1e3, 1)
        json.dump(pmc, open(os.path.join(P, tag + "_abn_pmc.json"), "w"), indent=1, sort_keys=True)
    for f in ("bench.json", "bench_quick.json", "abn_microbench.log", "micro.jsonl", "conv_shapes.md"):
        if os.path.exists(os.path.join(G, f)):
            shutil.copy(os.path.join(G, f), os.path.join(P, tag + "_" + f))


if __name__ == "__main__":
    main()
