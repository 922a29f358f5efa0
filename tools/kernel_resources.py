"""Static resource table of every gfx950 kernel of libskd_hip.so: VGPRs / AGPRs / SGPRs, LDS bytes per workgroup, scratch bytes per
lane and the occupancy the compiler reports (waves per SIMD), from hipcc's `-Rpass-analysis=kernel-resource-usage` remarks -- no GPU
needed (hipcc cross-compiles).  Used for DESIGN.md's statements about which launches fill a compute unit (the grid-barrier
InPlace-ABN passes: 16 waves x 128 VGPRs = the whole register file) and which kernels spill.

    python tools/kernel_resources.py [out.md]          (default: prints markdown to stdout)
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from structure_knowledge_distillation_amd import build as B  # noqa: E402

FIELDS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def demangle(names):
    import shutil
    filt = next((f for f in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", shutil.which("llvm-cxxfilt"), shutil.which("c++filt")) if f and os.path.exists(f)), None)
    if filt is None:
        return names
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return out if len(out) == len(names) else names


def one(src, tmp):
    flags = [f for f in B.FLAGS if f not in ("-shared",)]
    cmd = [B.hipcc_path()] + flags + ["-c", "-I", B.INCLUDE, "-I", B.CSRC, src, "-Rpass-analysis=kernel-resource-usage", "-o",
                                      os.path.join(tmp, os.path.basename(src) + ".o")]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "file": os.path.basename(src)}
            rows.append(cur)
            continue
        for f in FIELDS:
            m = re.search(r"remark:\s+" + re.escape(f) + r": (\d+)", line)
            if m and cur is not None:
                cur[f] = int(m.group(1))
    return rows


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return name.replace("skd::", "")


def main(out):
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(8) as ex:
        rows = [r for rs in ex.map(lambda s: one(s, tmp), B.sources()) for r in rs]
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["pretty"] = short(d)
    lines = ["# Static resources of the gfx950 kernels (hipcc -Rpass-analysis=kernel-resource-usage; `python tools/kernel_resources.py`)", "",
             "Occupancy = waves per SIMD the compiler reports from the register / LDS budget (8 = the hardware maximum); a 1024-thread",
             "workgroup is 4 waves per SIMD, a 256-thread workgroup 1.  LDS: static allocations only (the MFMA GEMM kernels take their",
             "64 KiB panels as dynamic shared memory).  %d kernels (template instantiations counted)." % len(rows), "",
             "| file | kernel | VGPRs | AGPRs | SGPRs | LDS B / workgroup | scratch B / lane | occupancy |", "|---|---|---|---|---|---|---|---|"]
    for r in sorted(rows, key=lambda r: (r["file"], r["pretty"])):
        lines.append("| %s | `%s` | %s | %s | %s | %s | %s | %s |" % (r["file"], r["pretty"][:110], r.get("VGPRs", ""), r.get("AGPRs", ""), r.get("TotalSGPRs", ""),
                                                                 r.get("LDS Size [bytes/block]", ""), r.get("ScratchSize [bytes/lane]", ""),
                                                                 r.get("Occupancy [waves/SIMD]", "")))
    spill = [r for r in rows if r.get("ScratchSize [bytes/lane]", 0) > 0]
    lines += ["", "Kernels with scratch (register spills or dynamically indexed arrays): %d" % len(spill)]
    for r in sorted(spill, key=lambda r: -r["ScratchSize [bytes/lane]"]):
        lines.append("* `%s` (%s): %d B / lane, %s VGPRs" % (r["pretty"][:140], r["file"], r["ScratchSize [bytes/lane]"], r.get("VGPRs")))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
