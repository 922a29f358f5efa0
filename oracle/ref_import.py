"""Import the *real* reference Python (read-only tree at /root/reference) with
the three import stubs SURVEY.md appendix C describes.  Only usable in the build
container; the GPU box has no /root/reference, so nothing that runs there may
call this.  Used by tests/golden/make_golden.py to generate fixtures and by the
``-m "not gpu"`` tests that cross-check the restatements when the tree exists.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("SKD_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "utils", "criterion.py"))


_loaded = {}


def load_reference(abn_module_factory):
    """Returns a namespace with the reference's modules.

    abn_module_factory: module object exposing ``InPlaceABN`` and
    ``InPlaceABNSync`` (the CPU restatement in oracle/abn_torch.py) -- the
    reference's own ``libs`` needs nvcc/THC/torch.utils.ffi and cannot load.
    """
    if "ns" in _loaded:
        return _loaded["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True  # never drop __pycache__ into the reference tree
    import torch

    # utils/utils.py:10  -> from tensorboardX import SummaryWriter
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    # networks/sagan_models.py:6 -> from torchvision import transforms
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    saved = {k: sys.modules.get(k) for k in
             ("tensorboardX", "torchvision", "torchvision.transforms", "libs", "utils", "networks",
              "utils.utils", "utils.criterion", "networks.pspnet_combine",
              "networks.sagan_models", "networks.spectral")}
    sys.modules["tensorboardX"] = tbx
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms
    # networks/pspnet_combine.py:11 -> from libs import InPlaceABN, InPlaceABNSync
    libs = types.ModuleType("libs")
    libs.InPlaceABN = abn_module_factory.InPlaceABN
    libs.InPlaceABNSync = abn_module_factory.InPlaceABNSync
    sys.modules["libs"] = libs
    for k in ("utils", "networks", "utils.utils", "utils.criterion", "networks.pspnet_combine",
              "networks.sagan_models", "networks.spectral"):
        sys.modules.pop(k, None)
    sys.path.insert(0, REF_ROOT)
    try:
        import importlib
        ns = types.SimpleNamespace()
        ns.utils_utils = importlib.import_module("utils.utils")
        ns.criterion = importlib.import_module("utils.criterion")
        ns.pspnet = importlib.import_module("networks.pspnet_combine")
        ns.sagan = importlib.import_module("networks.sagan_models")
        ns.spectral = importlib.import_module("networks.spectral")
    finally:
        sys.path.remove(REF_ROOT)
        # give the names back so the product package / tests can own them
        for k in ("utils", "networks", "utils.utils", "utils.criterion", "networks.pspnet_combine",
                  "networks.sagan_models", "networks.spectral", "libs",
                  "tensorboardX", "torchvision", "torchvision.transforms"):
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    # criterion.py:104,109 hard-code .cuda(); neutralise on a CPU-only box
    if not torch.cuda.is_available():
        ns.cuda_is_identity = True
    _loaded["ns"] = ns
    return ns


class cpu_cuda_identity:
    """Context manager: make Tensor.cuda() the identity (CriterionAdditionalGP, criterion.py:104,109)."""

    def __enter__(self):
        import torch
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        return self

    def __exit__(self, *exc):
        import torch
        torch.Tensor.cuda = self._orig
        return False


def load_reference_evaluate(abn_module_factory):
    """The reference's networks/evaluate.py (confusion matrix, whole-image prediction, mIoU), imported from where it lies
    with stubs for what this image lacks: ``cv2`` (only the out-of-scope dataset readers call it), ``torchvision`` /
    ``torchvision.models`` (imported, unused), ``libs`` (as in load_reference).  scipy.ndimage and PIL are installed.
    Returns the module; the caller wraps calls in ``evaluate_shims()`` (numpy 2 dropped ``np.int``, evaluate.py:194;
    ``.cuda()`` on a CPU-only box, evaluate.py:108,166)."""
    if "evaluate" in _loaded:
        return _loaded["evaluate"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    names = ("cv2", "torchvision", "torchvision.models", "torchvision.transforms", "tensorboardX", "libs", "networks", "dataset",
             "utils", "networks.evaluate", "networks.pspnet_combine", "dataset.datasets")
    saved = {k: sys.modules.get(k) for k in names}
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.transforms = types.ModuleType("torchvision.transforms")
    libs = types.ModuleType("libs")
    libs.InPlaceABN = abn_module_factory.InPlaceABN
    libs.InPlaceABNSync = abn_module_factory.InPlaceABNSync
    for k in names:
        sys.modules.pop(k, None)
    sys.modules.update({"cv2": types.ModuleType("cv2"), "torchvision": tv, "torchvision.models": tv.models,
                        "torchvision.transforms": tv.transforms, "libs": libs})
    sys.path.insert(0, REF_ROOT)
    try:
        import importlib
        mod = importlib.import_module("networks.evaluate")
    finally:
        sys.path.remove(REF_ROOT)
        for k in names:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    _loaded["evaluate"] = mod
    return mod


class evaluate_shims(cpu_cuda_identity):
    """cpu_cuda_identity + nn.Module.cuda identity + ``np.int`` for the duration of a call into the reference's evaluate.py."""

    def __enter__(self):
        import numpy as np
        import torch
        super().__enter__()
        self._mod_cuda = torch.nn.Module.cuda
        torch.nn.Module.cuda = lambda self, *a, **k: self
        self._had_int = hasattr(np, "int")
        if not self._had_int:
            np.int = int
        return self

    def __exit__(self, *exc):
        import numpy as np
        import torch
        torch.nn.Module.cuda = self._mod_cuda
        if not self._had_int:
            del np.int
        return super().__exit__(*exc)
