"""Fold rocprofv3 --pmc passes over `tools/kernel_microbench.py pmc` into one small JSON for profiles/.

    python tools/summarise_pmc.py <out.json> <microbench_stdout.log> <pass_dir> [<pass_dir> ...]

Every pass directory holds a *_counter_collection.csv (one row per dispatch per counter).  Dispatches are mapped to
microbench cases through the marker launches (act_kernel<0>, grid = 256 * (case id + 1) threads) and averaged per
(case, kernel).  HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE (KB) is DOUBLED for wide coalesced reads on
gfx950 (it tallies 128-byte requests at 64 bytes), WRITE_SIZE (KB) is taken as is; when the request-size pass
(TCC_EA0_RDREQ_{32B,64B,128B}_sum) is present the exact read bytes are reported next to the doubled figure.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"skd::\(anonymous namespace\)::([^(]+)\(", name)
    if m:
        return m.group(1).strip()
    return name[:60]


def read_pass(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            rows.extend(csv.DictReader(fh))
    by_dispatch = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        d_ = by_dispatch.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                                                              "wg": int(r["Workgroup_Size"]), "counters": {},
                                                              "vgpr": int(r.get("VGPR_Count", 0) or 0),
                                                              "lds": int(r.get("LDS_Block_Size", 0) or 0)})
        d_["counters"][r["Counter_Name"]] = d_["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        try:
            d_["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        except Exception:
            pass
    return by_dispatch


def main():
    out_path, log_path, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    manifest = None
    for line in open(log_path):
        line = line.strip()
        if line.startswith("{") and '"manifest"' in line:
            manifest = json.loads(line)["manifest"]
    assert manifest, "no manifest line in %s" % log_path
    cases = {c["id"]: dict(c, kernels={}) for c in manifest}
    for d in dirs:
        cur = None
        for disp in read_pass(d).values():
            n = disp["name"]
            if "act_kernel<0>" in n:
                cur = disp["grid"] // 256 - 1
                continue
            if cur is None or cur not in cases or "skd::" not in n:
                continue
            k = cases[cur]["kernels"].setdefault(short(n), {"grid_threads": disp["grid"], "workgroup": disp["wg"],
                                                             "vgpr": disp["vgpr"], "lds_bytes": disp["lds"], "_acc": collections.defaultdict(list)})
            for cn, cv in disp["counters"].items():
                k["_acc"][cn].append(cv)
            if "dur_us" in disp:
                k["_acc"]["dur_us_under_pmc"].append(disp["dur_us"])
    for c in cases.values():
        tot_fetch = tot_write = tot_exact = 0.0
        have_f = have_w = have_x = False
        for k in c["kernels"].values():
            acc = k.pop("_acc")
            for cn, vals in acc.items():
                k[cn] = round(sum(vals) / len(vals), 3)
            k["launches_averaged"] = max(len(v) for v in acc.values())
            if "FETCH_SIZE" in k:
                have_f = True
                tot_fetch += 2.0 * k["FETCH_SIZE"] * 1024.0
            if "WRITE_SIZE" in k:
                have_w = True
                tot_write += k["WRITE_SIZE"] * 1024.0
            if "TCC_EA0_RDREQ_128B_sum" in k:
                have_x = True
                r128, r64, r32 = k.get("TCC_EA0_RDREQ_128B_sum", 0.0), k.get("TCC_EA0_RDREQ_64B_sum", 0.0), k.get("TCC_EA0_RDREQ_32B_sum", 0.0)
                other = max(0.0, k.get("TCC_EA0_RDREQ_sum", 0.0) - r128 - r64 - r32)
                k["read_bytes_by_request_size"] = r128 * 128 + r64 * 64 + r32 * 32 + other * 64
                tot_exact += k["read_bytes_by_request_size"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k and k["GRBM_GUI_ACTIVE"]:
                # Fraction of the chip's SIMD-cycles with the matrix pipe busy.  SQ_VALU_MFMA_BUSY_CYCLES is summed over
                # all SIMDs (= 64 cycles x SQ_INSTS_MFMA for v_mfma_f32_32x32x2_f32, checked against the instruction
                # count); GRBM_GUI_ACTIVE is reported summed over the 8 XCCs (8 x clock x duration), so one XCC's active
                # cycles = GRBM / 8 and the chip offers 256 CUs x 4 SIMDs x GRBM / 8 SIMD-cycles.
                k["mfma_pipe_busy_frac"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / (k["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4), 4)
                k["clock_GHz_under_pmc"] = round(k["GRBM_GUI_ACTIVE"] / 8.0 / (k.get("dur_us_under_pmc", 0) * 1e3), 3) if k.get("dur_us_under_pmc") else None
        if c.get("algo_bytes") and have_f and have_w:
            c["hbm_bytes (2*FETCH_SIZE + WRITE_SIZE, all kernels of one call)"] = round(tot_fetch + tot_write)
            c["hbm_over_algorithmic"] = round((tot_fetch + tot_write) / c["algo_bytes"], 4)
            c["hbm_read_MB"], c["hbm_write_MB"] = round(tot_fetch / 1e6, 2), round(tot_write / 1e6, 2)
            if have_x:
                c["hbm_over_algorithmic (reads by request size)"] = round((tot_exact + tot_write) / c["algo_bytes"], 4)
    json.dump({"passes": [os.path.basename(os.path.normpath(d)) for d in dirs], "cases": list(cases.values())},
              open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
