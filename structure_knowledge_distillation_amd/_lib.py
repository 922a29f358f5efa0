"""ctypes binding of libskd_hip.so (the C ABI declared in include/skd.h).

Replaces the reference's cffi shim (libs/_ext/__init__.py + libs/src/lib_cffi.cpp, which unwrap
THCudaTensor*): here the Python side passes raw device pointers (``tensor.data_ptr()``) and the
current HIP stream.  There is NO fallback: if the shared library is missing or a symbol cannot
be resolved, every op raises -- a GPU box must never silently run an eager substitute.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libskd_hip.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "include", "skd.h"))

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float
_D = _c.c_double

# name -> (restype, argtypes); mirrors include/skd.h one to one (tests/test_abi.py checks that
# every prototype in the header is listed here and exported by the .so).
SIGNATURES = {
    "skd_abi_version": (_I, []),
    "skd_target_arch": (_I, []),
    "skd_bn_mean_var": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "skd_bn_forward": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P]),
    "skd_bn_edz_eydz": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P]),
    "skd_bn_backward": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P]),
    "skd_leaky_relu": (_I, [_L, _P, _F, _P]),
    "skd_leaky_relu_backward": (_I, [_L, _P, _P, _F, _P]),
    "skd_elu": (_I, [_L, _P, _P]),
    "skd_elu_backward": (_I, [_L, _P, _P, _P]),
    "skd_elu_inv": (_I, [_L, _P, _P]),
    "skd_abn_workspace_floats": (_L, [_I, _I, _I]),
    "skd_abn_forward_train": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _F, _P, _P]),
    "skd_abn_stats": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "skd_abn_apply": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _F, _I, _F, _P]),
    "skd_abn_apply_residual": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P]),
    "skd_abn_apply_to": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P]),
    "skd_abn_forward_train_to": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _F, _P, _P]),
    "skd_abn_relu_backward_reduce": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "skd_abn_relu_backward_dx": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P]),
    "skd_abn_apply_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P]),
    "skd_abn_nhwc_workspace_floats": (_L, [_L, _I]),
    "skd_abn_stats_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P]),
    "skd_abn_apply_nhwc_to": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P]),
    "skd_abn_forward_train_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _F, _P, _P]),
    "skd_abn_backward_reduce_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P, _P]),
    "skd_abn_backward_dx_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _I, _P]),
    "skd_abn_relu_backward_reduce_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "skd_abn_relu_backward_reduce_nhwc_x": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "skd_abn_relu_backward_dx_nhwc_x": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "skd_abn_backward_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _I, _P, _P]),
    "skd_abn_relu_backward_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P, _P]),
    "skd_abn_relu_backward_dx_nhwc": (_I, [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "skd_abn_combine_stats": (_I, [_I, _I, _P, _P, _I, _P, _P, _P, _P, _F, _D, _P]),
    "skd_abn_update_running": (_I, [_I, _P, _P, _P, _P, _F, _D, _P]),
    "skd_abn_backward_reduce": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P, _P]),
    "skd_abn_backward_dx": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P]),
    "skd_abn_backward": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _I, _P, _P]),
    "skd_pixelwise_workspace_floats": (_L, [_I, _I]),
    "skd_pixelwise_loss": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "skd_maxpool_argmax": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "skd_pairwise_ldm": (_I, [_I]),
    "skd_channel_l2_normalise": (_I, [_I, _I, _I, _P, _P, _I, _P, _I, _P, _P]),
    "skd_pairwise_workspace_floats": (_L, [_I, _I]),
    "skd_pairwise_gram_loss": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "skd_pairwise_backward_workspace_floats": (_L, [_I, _I, _I]),
    "skd_pairwise_backward": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "skd_maxunpool_scatter": (_I, [_I, _I, _I, _I, _I, _P, _L, _P, _P, _P]),
    "skd_maxpool_argmax_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "skd_maxunpool_scatter_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _L, _P, _P, _P]),
    "skd_spectral_workspace_floats": (_L, [_I, _I]),
    "skd_spectral_norm_forward": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "skd_spectral_norm_backward": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "skd_spectral_norm_forward_multi": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "skd_spectral_norm_backward_multi": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "skd_ce_dsn_workspace_floats": (_L, [_I, _I, _I, _I, _I, _I]),
    "skd_ce_dsn_forward": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P]),
    "skd_ppm_pooled_floats": (_L, [_I, _I, _P]),
    "skd_ppm_pool": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "skd_ppm_pool_backward": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "skd_ppm_concat": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "skd_ppm_concat_backward": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "skd_conv1x1_abn_supported": (_I, [_L, _I, _I]),
    "skd_conv1x1_abn_nhwc": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P]),
    "skd_abn_pack_eval_params": (_I, [_I, _P, _P, _P, _P, _F, _P, _P]),
    "skd_conv1x1_abn_pro_nhwc": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _F, _P]),
    "skd_pairwise_small": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "skd_cs_transform": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _I, _P, _P]),
    "skd_ppm_nhwc_workspace_floats": (_L, [_I, _I, _I, _I, _I, _I, _P]),
    "skd_ppm_pool_nhwc": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "skd_ppm_pool_backward_nhwc": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "skd_ppm_concat_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "skd_ppm_concat_backward_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "skd_ppm_fold_nhwc_workspace_floats": (_L, [_I, _I, _I, _I, _I, _P]),
    "skd_ppm_fold_nhwc": (_I, [_I, _I, _I, _I, _I, _P, _P, _L, _P, _P]),
    "skd_ppm_fold_backward_nhwc": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P]),
    "skd_maxpool3x3s2_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "skd_head1x1_supported": (_I, [_I, _I, _I]),
    "skd_head1x1_forward_nhwc": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "skd_head1x1_backward_workspace_floats": (_L, [_I, _I, _I, _I]),
    "skd_head1x1_backward_nhwc": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "skd_maxpool3x3s2_backward_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "skd_abn_relu_maxpool3x3s2_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P]),
    "skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "skd_abn_relu_maxpool3x3s2_backward_dx_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F,
                                                         _I, _P]),
    "skd_seg_confusion": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "skd_sum_f32": (_I, [_L, _P, _P, _F, _P, _P]),
    "skd_sync_handle_bytes": (_I, []),
    "skd_sync_max_floats": (_I, []),
    "skd_sync_create": (_P, [_I, _I, _P]),
    "skd_sync_connect": (_I, [_P, _P]),
    "skd_sync_destroy": (_I, [_P]),
    "skd_sync_all_gather": (_I, [_P, _I, _P, _P, _P]),
    "skd_abn_sync_stats": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _F, _D, _P]),
    "skd_abn_sync_grad_stats": (_I, [_P, _I, _P, _P, _P]),
    "skd_sync_set_timeout": (_I, [_P, _D]),
    "skd_abn_forward_train_nhwc_sync": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _F, _D, _P, _P]),
    "skd_abn_backward_nhwc_sync": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _I, _P, _P]),
    "skd_abn_relu_backward_nhwc_sync": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P, _P]),
    "skd_status_words": (_I, []),
    "skd_status_read": (_I, [_P]),
    "skd_status_clear": (_I, []),
    "skd_abn_set_fused_max_workgroups": (_I, [_I]),
    "skd_abn_set_fused": (_I, [_I]),
    "skd_abn_get_fused": (_I, []),
    "skd_abn_set_sync_fused": (_I, [_I]),
    "skd_abn_get_sync_fused": (_I, []),
    "skd_abn_sync_form_counts": (_I, [_P]),
}

_lib = None
_test_backend = None  # see install_test_backend()


class SkdLibraryError(RuntimeError):
    pass


def header_prototypes(path=HEADER_PATH):
    """Names of all functions declared in include/skd.h (used by tests/test_abi.py)."""
    with open(path) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(skd_[a-z0-9_]+)\s*\(", text)))


def load(path=None):
    """dlopen libskd_hip.so and type every entry point.  Raises SkdLibraryError when absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise SkdLibraryError(
            "libskd_hip.so not found at %s -- build it with "
            "`python -m structure_knowledge_distillation_amd.build` (there is no CPU/eager fallback)" % path)
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise SkdLibraryError("cannot load %s: %s" % (path, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise SkdLibraryError("libskd_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    if path == LIB_PATH:
        _lib = lib
    return lib


def get():
    """The library object the ops call into (the HIP library, or a test double)."""
    if _test_backend is not None:
        return _test_backend
    if _timing is not None:
        return _TimedLib(load())
    return load()


# ---- optional per-entry HIP-event timing (bench.py's roofline leg) ---------------------------------
_timing = None


class _TimedLib:
    """Brackets the selected C-ABI calls with HIP events recorded on the stream the kernels are
    launched on (torch's current stream, the `stream` argument every op passes)."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        rec = _timing.get(name) if _timing is not None else None
        if rec is None:
            return fn
        import torch

        def timed(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*args)
            e1.record()
            rec.append((e0, e1, args[:4]))
            return r
        return timed


def enable_kernel_timing(names):
    """Start collecting (start event, end event, first four args) for every call of ``names``."""
    global _timing
    _timing = {n: [] for n in names}


def disable_kernel_timing():
    """Stop collecting; returns {name: [(ms, first four args), ...]} (synchronises the device)."""
    global _timing
    import torch
    out = {}
    if _timing is not None:
        torch.cuda.synchronize()
        for n, recs in _timing.items():
            out[n] = [(e0.elapsed_time(e1), dims) for e0, e1, dims in recs]
    _timing = None
    return out


def install_test_backend(backend):
    """TEST HOOK ONLY: route the C-ABI calls to ``backend`` (an object exposing the same
    functions on HOST pointers).  tests/ uses it to exercise the host-side logic (autograd
    wiring, in-place contract, cross-rank statistics exchange over gloo) on a box without a GPU.
    Product code never calls this; with no backend installed every op requires CUDA tensors."""
    global _test_backend
    _test_backend = backend


def test_backend_active():
    return _test_backend is not None


class SkdDeviceError(RuntimeError):
    """A kernel gave up waiting inside a launch (include/skd.h section 13): its outputs are NaN from there on."""


_STATUS_MEANING = {
    0: "a cross-replica InPlace-ABN exchange timed out waiting for a peer rank (SKD_SYNC_TIMEOUT_S): a rank stalled, died or "
       "called the synchronised layers in a different order; statistics are NaN from that exchange on",
    1: "the grid barrier of a one-launch InPlace-ABN pass timed out: its workgroups were not co-resident (device shared with "
       "another grid-barrier launch, partitioned or CU-masked device) -- set SKD_ABN_FUSED=0 or lower "
       "skd_abn_set_fused_max_workgroups",
}


def device_status():
    """The device-raised error words (no device synchronisation), or None when the buffer does not exist."""
    lib = get()
    n = lib.skd_status_words()
    buf = (ctypes.c_uint * n)()
    if not lib.skd_status_read(ctypes.cast(buf, ctypes.c_void_p)):
        return None
    return list(buf)


def sync_form_counts():
    """(one-launch, three-launch) counts of the synchronised InPlace-ABN calls so far (include/skd.h section 13)."""
    buf = (ctypes.c_int64 * 2)()
    get().skd_abn_sync_form_counts(ctypes.cast(buf, ctypes.c_void_p))
    return int(buf[0]), int(buf[1])


def raise_on_device_errors():
    """Called once per step (NetModel.optimize_parameters) and after a logged scalar has been read back: turns a timed-out
    in-kernel wait into an exception instead of NaN that silently spreads through the gradient all-reduce."""
    words = device_status()
    if words and any(words):
        get().skd_status_clear()
        what = "; ".join("%s (code 0x%08x)" % (_STATUS_MEANING.get(i, "status word %d" % i), w) for i, w in enumerate(words) if w)
        raise SkdDeviceError("structure_knowledge_distillation_amd: " + what)


def check(ok, what):
    """1 = success / 0 = failure, like libs/functions.py:13-16 (_check)."""
    if not ok:
        raise RuntimeError("HIP error encountered in %s" % what)


def ptr(t):
    """Device pointer of a tensor, or NULL for None (lib_cffi.cpp:62-63 convention)."""
    return None if t is None else t.data_ptr()


def int_array(values):
    """Host int[] argument (e.g. the pyramid sizes of skd_ppm_*)."""
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def ptr_array(tensors):
    """Host array of device pointers (``const float *const *`` arguments)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def stream_of(t):
    """hipStream_t of torch's current stream on t's device (replaces THCState_getCurrentStream)."""
    if _test_backend is not None or not t.is_cuda:
        return None
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def require_device(*tensors):
    """Fail loudly instead of falling back when handed CPU tensors (unless a test double is in)."""
    if _test_backend is not None:
        return
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SkdLibraryError(
                "structure_knowledge_distillation_amd ops run on MI355X only: got a %s tensor "
                "(no CPU fallback exists)" % t.device)
