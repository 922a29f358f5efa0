"""The cffi-replacement stub of INTEGRATION.md section 2 (tests/integration/libs_ext): the document shows the file
verbatim, its nine functions carry the names and parameter lists of the reference's libs/src/lib_cffi.h, and on the GPU
they reproduce the reference's own call sequence of libs/functions.py:75-162 (forward and backward of InPlaceABN)."""
import importlib.util
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "integration", "libs_ext", "__init__.py")


def _load():
    spec = importlib.util.spec_from_file_location("skd_libs_ext_stub", STUB)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_integration_md_shows_the_stub_verbatim():
    body = open(STUB).read().split('"""', 2)[2].strip()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert body in doc, "INTEGRATION.md section 2 must contain tests/integration/libs_ext/__init__.py verbatim (no elisions)"
    assert "…" not in doc.split("## 2.")[1].split("## 3.")[0]


def test_stub_exports_match_lib_cffi_h():
    ext = _load()
    want = {"bn_mean_var_cuda": 3, "bn_forward_cuda": 8, "bn_edz_eydz_cuda": 7, "bn_backard_cuda": 11, "leaky_relu_cuda": 2,
            "leaky_relu_backward_cuda": 3, "elu_cuda": 1, "elu_backward_cuda": 2, "elu_inv_cuda": 1}
    for name, nargs in want.items():
        assert len(inspect.signature(getattr(ext, name)).parameters) == nargs, name
    hdr = "/root/reference/libs/src/lib_cffi.h"
    if os.path.exists(hdr):          # build container: the parameter counts come from the reference's own header
        text = open(hdr).read()
        for m in re.finditer(r"int (\w+)\(([^;]*)\);", text):
            assert want[m.group(1)] == len([a for a in m.group(2).split(",") if a.strip()]), m.group(1)
        assert len(re.findall(r"int \w+\(", text)) == 9


@pytest.mark.gpu
def test_stub_runs_the_reference_call_sequence():
    """libs/functions.py:75-108 (forward) and :112-162 (backward) of InPlaceABN, line for line, through the stub."""
    from oracle import abn_torch
    ext = _load()
    dev = "cuda"
    torch.manual_seed(0)
    x = (torch.randn(4, 6, 9, 7) * 2 + 1)
    w, b = torch.randn(6), torch.randn(6)
    gz = torch.randn(4, 6, 9, 7)
    xo = x.double().requires_grad_(True)
    wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
    zo = abn_torch.abn_autograd(xo, wo, bo, torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64), True, 0.1,
                                1e-5, "leaky_relu", 0.01)
    zo.backward(gz.double())
    # forward: mean_var -> forward (in place: y = z = x) -> leaky_relu            functions.py:84-98
    xg, wg, bg = x.to(dev), w.to(dev), b.to(dev)
    mean, var = xg.new_empty(6), xg.new_empty(6)
    assert ext.bn_mean_var_cuda(xg, mean, var) == 1
    assert ext.bn_forward_cuda(xg, mean, var, wg, bg, xg, xg, 1e-5) == 1
    assert ext.leaky_relu_cuda(xg, 0.01) == 1
    assert float((xg.cpu().double() - zo.detach()).abs().max()) < 1e-5
    # backward: undo the activation on (z, dz) -> edz/eydz -> backward            functions.py:118-152, 54-62
    z, dz = xg, gz.to(dev).clone()
    assert ext.leaky_relu_backward_cuda(z, dz, 0.01) == 1
    assert ext.leaky_relu_cuda(z, 1.0 / 0.01) == 1
    edz, eydz = z.new_empty(6), z.new_empty(6)
    assert ext.bn_edz_eydz_cuda(z, dz, wg, bg, edz, eydz, 1e-5) == 1
    dx, dw, db = torch.zeros_like(z), torch.zeros(6, device=dev), torch.zeros(6, device=dev)
    assert ext.bn_backard_cuda(dz, z, var, wg, bg, edz, eydz, dx, dw, db, 1e-5) == 1
    rel = lambda a, c: float((a.cpu().double() - c).norm() / c.norm())
    assert rel(dx, xo.grad) < 1e-4 and rel(dw, wo.grad) < 1e-4 and rel(db, bo.grad) < 1e-4
    # non-affine call: 0-dim weight / bias mean "absent" (lib_cffi.cpp:62-63)
    y2 = x.to(dev)
    assert ext.bn_forward_cuda(y2, mean, var, torch.tensor(0.0, device=dev), torch.tensor(0.0, device=dev), y2, y2, 1e-5) == 1
    assert abs(float(y2.mean())) < 1e-5
    e = torch.tensor([-1.0, 2.0], device=dev)
    assert ext.elu_cuda(e) == 1 and ext.elu_inv_cuda(e) == 1 and float((e - torch.tensor([-1.0, 2.0], device=dev)).abs().max()) < 1e-5
    d = torch.ones(2, device=dev)
    assert ext.elu_backward_cuda(torch.tensor([-0.5, 2.0], device=dev), d) == 1 and abs(float(d[0]) - 0.5) < 1e-6
