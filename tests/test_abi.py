"""The C-ABI library loads without a GPU and exports every symbol include/skd.h declares; the
ctypes table in _lib.py mirrors the header one to one; the plain-C oracle exports the same ABI."""
import ctypes
import os
import re

import pytest

from structure_knowledge_distillation_amd import _lib, build


@pytest.fixture(scope="module")
def so_path():
    return build.build()


def test_header_and_ctypes_table_agree():
    protos = _lib.header_prototypes()
    assert protos, "no prototypes parsed from include/skd.h"
    assert sorted(protos) == sorted(_lib.SIGNATURES.keys())


def test_header_argument_counts_match_ctypes_table():
    with open(_lib.HEADER_PATH) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    for m in re.finditer(r"\b(skd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else args.count(",") + 1
        assert n == len(_lib.SIGNATURES[name][1]), name


def test_library_exports_every_symbol(so_path):
    lib = ctypes.CDLL(so_path)
    for name in _lib.header_prototypes():
        assert hasattr(lib, name), "libskd_hip.so does not export %s" % name
    typed = _lib.load()
    assert typed.skd_abi_version() == 1 and typed.skd_target_arch() == 950
    # pure host-side size queries work without a device
    assert typed.skd_pairwise_ldm(9) == 128 and typed.skd_pairwise_ldm(4225) == 4352
    assert typed.skd_abn_workspace_floats(8, 64, 65536) >= 2 * 64
    assert typed.skd_spectral_workspace_floats(512, 4096) > 0


def test_code_object_is_gfx950(so_path):
    with open(so_path, "rb") as fh:
        blob = fh.read()
    assert b"gfx950" in blob and b"sm_" not in blob[:0]  # offload bundle target


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.SkdLibraryError):
        _lib.load(str(tmp_path / "nope.so"))


def test_c_oracle_exports_same_abi():
    from oracle import cref
    ref = cref.load(_lib.SIGNATURES)
    assert ref.skd_target_arch() == 0 and ref.skd_pairwise_ldm(129) == 256


def test_no_cpu_fallback_for_ops():
    import torch
    from structure_knowledge_distillation_amd import functional as SF, libs
    if _lib.test_backend_active():
        pytest.skip("C double installed")
    with pytest.raises(_lib.SkdLibraryError):
        SF.pixel_wise_loss(torch.randn(1, 3, 4, 4), torch.randn(1, 3, 4, 4))
    with pytest.raises(_lib.SkdLibraryError):
        libs.InPlaceABN(3)(torch.randn(2, 3, 4, 4))
