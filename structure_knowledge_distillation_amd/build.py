"""Build libskd_hip.so (all gfx950 kernels + the C ABI of include/skd.h) in-tree with hipcc.

    python -m structure_knowledge_distillation_amd.build [--force] [--verbose]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels with the
working tree (it must sit next to the sources so the loader in _lib.py finds it).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))
LIB_PATH = os.path.join(HERE, "libskd_hip.so")
STAMP = LIB_PATH + ".stamp"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH, "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    files.append(os.path.join(INCLUDE, "skd.h"))
    for f in files:
        h.update(f.encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def build(force=False, verbose=False):
    digest = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == digest:
                return LIB_PATH
    cmd = [hipcc_path()] + FLAGS + ["-I", INCLUDE, "-I", CSRC] + sources() + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed building libskd_hip.so (exit %d)" % res.returncode)
    if verbose and res.stdout:
        print(res.stdout)
    with open(STAMP, "w") as fh:
        fh.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
