"""ctypes loader of oracle/libskd_ref.so -- the plain-C restatement behind the SAME C ABI as
include/skd.h, on HOST pointers.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Used two ways by tests/: called directly on numpy / CPU-tensor pointers as the scalar checker of
the HIP kernels, and installed through ``_lib.install_test_backend`` as a C-ABI double so the
host-side autograd / distributed logic runs on a box without a GPU."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libskd_ref.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("abn_ref.c", "losses_ref.c", "input_ref.c", "sync_ref.c")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.run(["make", "-C", HERE, "-s", "-B"], check=True)
    return LIB_PATH


def load(signatures):
    """``signatures``: the {name: (restype, argtypes)} table of the product loader, so both
    libraries are typed from one place."""
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in signatures.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
