"""Per-kernel averages of the rocprofv3 --pmc passes run over tools/gemm_lab (tools/gpu_session.sh lab_pmc).
    python tools/summarise_lab_pmc.py gpurun_out/<tag> [out.json]
MFMA-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES-normalised GRBM_GUI_ACTIVE x 4 SIMDs x 256 CUs / ...):
reported simply as SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256) -- busy cycles summed over the CUs' matrix pipes
per active GPU cycle per CU; multiply by 2 for the convention of tools/summarise_pmc.py (GRBM_GUI_ACTIVE is summed over the 8
XCCs, the chip has 4 SIMDs per CU); HBM bytes = 2 x FETCH_SIZE KB (reads) and WRITE_SIZE KB (writes)."""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
agg = collections.OrderedDict()
for d in sorted(glob.glob(root + "/labpmc*")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"(?:skd::\(anonymous namespace\)::)?([A-Za-z0-9_]+(?:<[^>]*>)?)\(", name)
            short = m.group(1) if m and "skd::" in name else name[:48]
            key = (short, int(r["Grid_Size"]))
            agg.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (name, grid), d in agg.items():
    if name.startswith("__amd") or "sum_final" in name or "l2_norm" in name:
        continue
    m = {k: sum(v) / len(v) for k, v in d.items()}
    row = {"kernel": name, "grid": grid}
    gui = m.get("GRBM_GUI_ACTIVE")
    if gui:
        row["gpu_cycles"] = gui
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            row["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256.0) , 4)
            row["mfma_pipe_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 256 * 4), 4)   # tools/summarise_pmc.py convention
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
            row["mfma_mops_f32_per_cycle_per_cu"] = round(m["SQ_INSTS_VALU_MFMA_MOPS_F32"] / gui / 256.0, 3)
    wc = m.get("SQ_WAVE_CYCLES")
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
        if k in m and wc:
            row[k.lower() + "_per_wave_cycle"] = round(m[k] / wc, 4)
    for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
        if k in m:
            row[k.lower()] = m[k]
    if "FETCH_SIZE" in m:
        row["hbm_read_MB"] = round(m["FETCH_SIZE"] * 1024 * 2 / 1e6, 2)   # FETCH_SIZE is in KB; x2: the gfx950 correction of MI355X_MICROARCH.md (as tools/summarise_pmc.py)
    if "WRITE_SIZE" in m:
        row["hbm_write_MB"] = round(m["WRITE_SIZE"] * 1024 / 1e6, 2)
    rows.append(row)
    print(json.dumps(row))
if len(sys.argv) > 2:
    json.dump(rows, open(sys.argv[2], "w"), indent=1)
