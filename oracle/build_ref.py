"""oracle/_ref/libbn_ref.so -- the reference's OWN native kernels (libs/src/bn.cu + common.h + bn.h) built for gfx950
as a TEST-ONLY checker.  TEST INFRASTRUCTURE: nothing in the product imports or links it.

    python -m oracle.build_ref            (needs /root/reference and hipcc; a no-op elsewhere)

The reference's build (libs/build.sh: nvcc + torch.utils.ffi + THC) cannot run here, but its three kernel source files
are self-contained CUDA runtime code, and hipcc compiles CUDA-dialect device code once the RUNTIME API NAMES are
spelled the HIP way.  The recipe therefore reads the sources where they lie under /root/reference, applies the
token substitutions listed in SUBSTITUTIONS to an in-memory copy (nothing of the reference is written into this
repository; the translated text lives in a temporary directory that is deleted again), and compiles that with
hipcc into oracle/_ref/libbn_ref.so (git-ignored, travels to the GPU box like the product's own .so).

What the substitutions do, and nothing else:
  * cuda* runtime names -> hip* ; <cuda_runtime_api.h> -> <hip/hip_runtime.h> ; thrust::cuda::par -> thrust::hip::par
    (rocThrust ships with ROCm);
  * `#if __CUDA_ARCH__ >= 300` -> `#if 1`: hipcc does not define __CUDA_ARCH__, and without this the file would
    silently take its pre-Kepler shared-memory warpSum instead of the shuffle reduction every CUDA build of the
    reference runs;  `#if CUDART_VERSION >= 9000` stays false, i.e. the `__shfl_xor(value, laneMask, width)` branch.
The algorithm is untouched: WARP_SIZE stays 32 (common.h:8) -- on a 64-wide wavefront the width-32 shuffles reduce
the two half-waves independently and `shared[32]` collects one partial per 32 threads, exactly as on NVIDIA hardware
(threadIdx-based indexing only) -- two-pass variance, one workgroup per channel, thrust transform_if activations.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/libs/src"
OUT_DIR = os.path.join(HERE, "_ref")
LIB_PATH = os.path.join(OUT_DIR, "libbn_ref.so")
FILES = ("bn.cu", "common.h", "bn.h")
SUBSTITUTIONS = (
    ("#include <cuda_runtime_api.h>", "#include <hip/hip_runtime.h>"),
    ("cudaStream_t", "hipStream_t"),
    ("cudaError_t", "hipError_t"),
    ("cudaGetLastError", "hipGetLastError"),
    ("cudaSuccess", "hipSuccess"),
    ("thrust::cuda::par", "thrust::hip::par"),
    ("#if __CUDA_ARCH__ >= 300", "#if 1"),
)


def available():
    return all(os.path.isfile(os.path.join(REF_SRC, f)) for f in FILES)


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return shutil.which("hipcc")


def build(force=False, verbose=False):
    """Returns the path of the built library, or None when the reference tree / hipcc is absent."""
    if not available() or hipcc() is None:
        return LIB_PATH if os.path.exists(LIB_PATH) else None
    srcs = [os.path.join(REF_SRC, f) for f in FILES] + [os.path.abspath(__file__)]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="skd_bn_ref_")
    try:
        for f in FILES:
            with open(os.path.join(REF_SRC, f)) as fh:
                text = fh.read()
            for a, b in SUBSTITUTIONS:
                text = text.replace(a, b)
            with open(os.path.join(tmp, f if not f.endswith(".cu") else f[:-3] + ".hip"), "w") as fh:
                fh.write(text)
        cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-gpu-rdc", "-I", tmp,
               "-Wno-unused-result", "-Wno-deprecated-declarations", os.path.join(tmp, "bn.hip"), "-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout)
            raise RuntimeError("hipcc failed building oracle/_ref/libbn_ref.so (exit %d)" % res.returncode)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB_PATH


def load():
    """ctypes handle with the reference's export names (libs/src/bn.h:7-19), or None when the library is not built."""
    import ctypes
    if not os.path.exists(LIB_PATH):
        return None
    lib = ctypes.CDLL(LIB_PATH)
    I, F, P = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
    sig = {
        "_bn_mean_var_cuda": [I, I, I, P, P, P, P],
        "_bn_forward_cuda": [I, I, I, P, P, P, P, P, P, P, F, P],
        "_bn_edz_eydz_cuda": [I, I, I, P, P, P, P, P, P, F, P],
        "_bn_backward_cuda": [I, I, I, P, P, P, P, P, P, P, P, P, P, F, P],
        "_leaky_relu_cuda": [I, P, F, P],
        "_leaky_relu_backward_cuda": [I, P, P, F, P],
        "_elu_cuda": [I, P, P],
        "_elu_backward_cuda": [I, P, P, P],
        "_elu_inv_cuda": [I, P, P],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.restype = I
        fn.argtypes = args
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
