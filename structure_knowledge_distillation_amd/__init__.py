"""MI355X-native distillation hot path of irfanICMLL/structure_knowledge_distillation.

Sub-packages mirror the reference's import surface for this path so that its
``train_and_eval.py`` loop runs unchanged after the aliasing shown in INTEGRATION.md:

    libs      -> InPlaceABN / InPlaceABNSync            (reference: libs/)
    networks  -> kd_model.NetModel, pspnet_combine, sagan_models, spectral
    utils     -> criterion (the six loss classes), utils (sim_dis_compute ...), parallel shims

All device work goes through ``libskd_hip.so`` (hand-written gfx950 kernels behind the C ABI of
include/skd.h) or MIOpen/rocBLAS via PyTorch-ROCm for the convolutions.  There is no CPU path.
"""
__version__ = "0.1.0"

import os as _os

# MIOpen user find-db + kernel cache tuned for this step's convolution shapes on gfx950 (made by
# tools/miopen_tune.py with MIOpen's own tuner; the ROCm image ships no gfx950 database).  Must be in
# the environment before the first convolution initialises MIOpen.
MIOPEN_DB_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")
if _os.path.isdir(MIOPEN_DB_DIR):
    _os.environ.setdefault("MIOPEN_USER_DB_PATH", MIOPEN_DB_DIR)
    _os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", _os.path.join(MIOPEN_DB_DIR, "cache"))
