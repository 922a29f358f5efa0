"""-m gpu: every entry point of libskd_hip.so, called through the C ABI (ctypes, raw device
pointers), against the plain-C oracle (oracle/libskd_ref.so, same ABI on host pointers) on the
same seeded inputs, plus size-independent properties at BASELINE.json's full sizes.

Tolerances (fp32 kernels; the oracle accumulates in double):
  element-wise outputs            <= 2e-5 relative to the tensor's max magnitude
  per-channel reductions          <= 2e-5 relative (mean/var/edz/eydz), running stats <= 1e-6
  scalar losses                   <= 1e-5 relative   (north_star asks 1e-4)
  max-pool argmax / pooled value  bit-exact
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import structure_knowledge_distillation_amd.networks.pspnet_combine as PC_MOD  # noqa: E402  (its fused-form constants are patched by tests)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import cref
from structure_knowledge_distillation_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    return _lib.load()


@pytest.fixture(scope="module")
def ref():
    return cref.load(_lib.SIGNATURES)


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def gpu(t):
    return None if t is None else t.to(DEV)


def close(got, want, tol=2e-5, what="", floor=0.0):
    """max |got - want| <= tol * max(max|want|, floor); ``floor`` = magnitude of the terms an output is a
    difference of (a cancellation residue cannot be more accurate than its operands' rounding)."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(float(want.abs().max()), floor, 1e-30)
    err = float((got - want).abs().max()) / scale
    assert err <= tol, "%s: max err %.3e (rel to %.3e) > %.1e" % (what, err, scale, tol)


ABN_SHAPES = [(2, 3, 1), (1, 5, 7), (3, 4, 36), (2, 19, 4225), (4, 64, 4225), (2, 7, 8193), (2, 16, 16641),
              (1, 3, 65536), (5, 130, 9), (2, 128, 4)]


def _abn_inputs(N, C, S, seed, affine=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, S, generator=g) * 3.0 + torch.randn(1, C, 1, generator=g) * 5.0
    w = torch.randn(C, generator=g) if affine else None
    b = torch.randn(C, generator=g) if affine else None
    if affine and C >= 3:
        w[0] = 0.0          # dweight sign trick: zero weight gets zero grad (bn.cu:217-223)
        w[1] = -abs(w[1])   # gamma = |w| + eps
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    return x, w, b, rm, rv


@pytest.mark.parametrize("shape", ABN_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_abn_train_forward_backward(hip, ref, shape, act):
    N, C, S = shape
    if N * S == 1:
        pytest.skip("n == 1: running_var divides by zero in the reference too (functions.py:91)")
    x, w, b, rm, rv = _abn_inputs(N, C, S, seed=N * 1000 + C + S)
    if act == 2:
        # ELU is inverted from the OUTPUT in backward (log1p, bn.cu:364-377): keep the pre-activation above
        # about -4 so that exp(z)-1 stays invertible in fp32 (the reference has the same limit; ELU is never
        # instantiated on this path)
        w = w.clamp(-1.2, 1.2)
        w = torch.where(w.abs() < 0.3, torch.full_like(w, 0.7), w)   # 1/gamma amplifies log1p's rounding
        b = b.clamp(-0.5, 0.5)
    slope, eps, mom = 0.01, 1e-5, 0.1
    # ---- forward
    xr, rmr, rvr = x.clone(), rm.clone(), rv.clone()
    mr, vr = torch.empty(C), torch.empty(C)
    wsr = torch.empty(max(1, ref.skd_abn_workspace_floats(N, C, S)))
    assert ref.skd_abn_forward_train(N, C, S, P(xr), P(w), P(b), P(rmr), P(rvr), P(mr), P(vr), mom, eps, act, slope, P(wsr), None)
    xg, wg, bg, rmg, rvg = gpu(x), gpu(w), gpu(b), gpu(rm), gpu(rv)
    mg, vg = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    wsg = torch.empty(max(1, hip.skd_abn_workspace_floats(N, C, S)), device=DEV)
    assert hip.skd_abn_forward_train(N, C, S, P(xg), P(wg), P(bg), P(rmg), P(rvg), P(mg), P(vg), mom, eps, act, slope, P(wsg), None)
    torch.cuda.synchronize()
    close(mg, mr, 2e-5, "mean")
    close(vg, vr, 5e-5, "var")
    close(rmg, rmr, 2e-6, "running_mean")
    close(rvg, rvr, 1e-5, "running_var")
    close(xg, xr, 1e-4 if act == 2 else 3e-5, "z")
    # ---- backward (z from the ORACLE forward on both sides, so only the backward is compared)
    g = torch.Generator().manual_seed(99)
    dz = torch.randn(N, C, S, generator=g)
    dxr, dwr, dbr = torch.empty_like(x), torch.zeros(C), torch.zeros(C)
    er, eyr = torch.empty(C), torch.empty(C)
    assert ref.skd_abn_backward(N, C, S, P(xr), P(dz), P(vr), P(w), P(b), P(er), P(eyr), P(dxr), P(dwr), P(dbr), eps, act, slope, 1, P(wsr), None)
    zg, dzg, vgg = gpu(xr), gpu(dz), gpu(vr)
    dxg, dwg, dbg = torch.empty_like(zg), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    eg, eyg = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    assert hip.skd_abn_backward(N, C, S, P(zg), P(dzg), P(vgg), P(wg), P(bg), P(eg), P(eyg), P(dxg), P(dwg), P(dbg), eps, act, slope, 1, P(wsg), None)
    torch.cuda.synchronize()
    close(eg, er, 5e-5, "edz")
    close(eyg, eyr, 5e-5, "eydz")
    mul = float(((w.abs() + eps) / torch.sqrt(vr + eps)).max())       # dx = (dz - edz - y*eydz) * gamma * invstd
    close(dxg, dxr, 1e-4, "dx", floor=float(dz.abs().max()) * mul)
    close(dwg, dwr, 5e-5, "dweight")
    close(dbg, dbr, 5e-5, "dbias")
    assert float(dwg[0]) == 0.0 if (C >= 3 and act != 2) else True
    assert torch.equal(zg.cpu(), xr), "backward must not rewrite the saved output"


@pytest.mark.parametrize("shape", [(2, 5, 49), (1, 2048, 4225), (3, 64, 16641)])
def test_abn_eval_apply_and_nonaffine(hip, ref, shape):
    N, C, S = shape
    x, w, b, rm, rv = _abn_inputs(N, C, S, seed=5)
    for (ww, bb, act) in ((w, b, 0), (w, b, 1), (None, None, 0)):
        xr = x.clone()
        assert ref.skd_abn_apply(N, C, S, P(xr), P(rm), P(rv), P(ww), P(bb), 1e-5, act, 0.01, None)
        xg = gpu(x)
        assert hip.skd_abn_apply(N, C, S, P(xg), P(gpu(rm)), P(gpu(rv)), P(gpu(ww)), P(gpu(bb)), 1e-5, act, 0.01, None)
        close(xg, xr, 2e-5, "eval z")
    # eval-mode backward: edz = eydz = 0 (functions.py:146-147)
    z, dz = torch.randn(N, C, S), torch.randn(N, C, S)
    dxr, e, ey = torch.empty_like(z), torch.empty(C), torch.empty(C)
    assert ref.skd_abn_backward(N, C, S, P(z), P(dz), P(rv), P(w), P(b), P(e), P(ey), P(dxr), None, None, 1e-5, 0, 0.01, 0, P(torch.empty(8)), None)
    dxg, eg, eyg = torch.empty(N, C, S, device=DEV), torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    wsg = torch.empty(max(1, hip.skd_abn_workspace_floats(N, C, S)), device=DEV)
    assert hip.skd_abn_backward(N, C, S, P(gpu(z)), P(gpu(dz)), P(gpu(rv)), P(gpu(w)), P(gpu(b)), P(eg), P(eyg), P(dxg), None, None, 1e-5, 0, 0.01, 0, P(wsg), None)
    close(dxg, dxr, 2e-5, "eval dx")
    assert float(eg.abs().max()) == 0.0 and float(eyg.abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(2, 5, 49), (8, 256, 4225), (2, 64, 16641), (1, 7, 3)])
@pytest.mark.parametrize("act", [0, 1, 3])
def test_abn_fused_relu_and_residual(hip, ref, shape, act):
    """Inference fusion: x <- act(bn(x)) with act = ReLU, and x <- act(bn(x) + residual) in one pass."""
    N, C, S = shape
    x, w, b, rm, rv = _abn_inputs(N, C, S, seed=S)
    r = torch.randn(N, C, S, generator=torch.Generator().manual_seed(3))
    xr = x.clone()
    assert ref.skd_abn_apply(N, C, S, P(xr), P(rm), P(rv), P(w), P(b), 1e-5, act, 0.01, None)
    xg = gpu(x)
    assert hip.skd_abn_apply(N, C, S, P(xg), P(gpu(rm)), P(gpu(rv)), P(gpu(w)), P(gpu(b)), 1e-5, act, 0.01, None)
    close(xg, xr, 2e-5, "act(bn(x))")
    if act == 3:
        assert float(xg.min()) == 0.0
    xr, xg, rg = x.clone(), gpu(x), gpu(r)
    assert ref.skd_abn_apply_residual(N, C, S, P(xr), P(r), P(rm), P(rv), P(w), P(b), 1e-5, act, 0.01, None)
    assert hip.skd_abn_apply_residual(N, C, S, P(xg), P(rg), P(gpu(rm)), P(gpu(rv)), P(gpu(w)), P(gpu(b)), 1e-5, act, 0.01, None)
    close(xg, xr, 2e-5, "act(bn(x)+r)")
    assert torch.equal(rg.cpu(), r)
    # ReLU cannot be inverted from the output: the backward entries refuse it (return 0 -> RuntimeError in Python)
    if act == 3:
        e = torch.empty(C, device=DEV)
        ws = torch.empty(max(1, hip.skd_abn_workspace_floats(N, C, S)), device=DEV)
        assert hip.skd_abn_backward_reduce(N, C, S, P(xg), P(rg), P(gpu(w)), P(gpu(b)), P(e), P(e), 1e-5, 3, 0.01, P(ws), None) == 0


@pytest.mark.parametrize("shape", [(2, 5, 49), (8, 64, 16641), (4, 128, 4225), (1, 3, 65536), (3, 7, 1)])
@pytest.mark.parametrize("with_res", [False, True])
def test_abn_relu_training_fusion(hip, ref, shape, with_res):
    """out = relu(bn_batch(x) [+ r]) out of place; backward from (x, out, dout)."""
    N, C, S = shape
    x, w, b, rm, rv = _abn_inputs(N, C, S, seed=S + 7)
    g = torch.Generator().manual_seed(5)
    r = torch.randn(N, C, S, generator=g) if with_res else None
    outr, mr, vr, rmr, rvr = torch.empty_like(x), torch.empty(C), torch.empty(C), rm.clone(), rv.clone()
    assert ref.skd_abn_forward_train_to(N, C, S, P(x), P(r), P(outr), P(w), P(b), P(rmr), P(rvr), P(mr), P(vr), 0.1, 1e-5, 3, 0.0, P(torch.empty(2 * C)), None)
    xg = gpu(x)
    outg, mg, vg, rmg, rvg = torch.empty_like(xg), torch.empty(C, device=DEV), torch.empty(C, device=DEV), gpu(rm), gpu(rv)
    ws = torch.empty(max(1, hip.skd_abn_workspace_floats(N, C, S)), device=DEV)
    assert hip.skd_abn_forward_train_to(N, C, S, P(xg), P(gpu(r)), P(outg), P(gpu(w)), P(gpu(b)), P(rmg), P(rvg), P(mg), P(vg), 0.1, 1e-5, 3, 0.0, P(ws), None)
    assert torch.equal(xg.cpu(), x), "the convolution output must stay untouched"
    close(mg, mr, 2e-5, "mean"); close(vg, vr, 5e-5, "var"); close(rmg, rmr, 2e-6, "running_mean"); close(rvg, rvr, 1e-5, "running_var")
    close(outg, outr, 3e-5, "out")
    assert float(outg.min()) >= 0.0
    # apply_to with given statistics (the synchronised path)
    out2 = torch.empty_like(xg)
    assert hip.skd_abn_apply_to(N, C, S, P(xg), P(gpu(r)), P(out2), P(gpu(mr)), P(gpu(vr)), P(gpu(w)), P(gpu(b)), 1e-5, 3, 0.0, None)
    close(out2, outr, 3e-5, "apply_to")
    # backward on identical (x, out, dout, mean, var)
    dout = torch.randn(N, C, S, generator=g)
    er, eyr = torch.empty(C), torch.empty(C)
    assert ref.skd_abn_relu_backward_reduce(N, C, S, P(x), P(outr), P(dout), P(mr), P(vr), P(er), P(eyr), 1e-5, P(torch.empty(2 * C)), None)
    eg, eyg = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    og, dg, mgg, vgg = gpu(outr), gpu(dout), gpu(mr), gpu(vr)
    assert hip.skd_abn_relu_backward_reduce(N, C, S, P(xg), P(og), P(dg), P(mgg), P(vgg), P(eg), P(eyg), 1e-5, P(ws), None)
    close(eg, er, 5e-5, "edz"); close(eyg, eyr, 5e-5, "eydz")
    dxr, drr, dwr, dbr = torch.empty_like(x), (torch.empty_like(x) if with_res else None), torch.zeros(C), torch.zeros(C)
    assert ref.skd_abn_relu_backward_dx(N, C, S, P(x), P(outr), P(dout), P(mr), P(vr), P(w), P(er), P(eyr), P(dxr), P(drr), P(dwr), P(dbr), 1e-5, 1, None)
    dxg, drg, dwg, dbg = torch.empty_like(xg), (torch.empty_like(xg) if with_res else None), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    assert hip.skd_abn_relu_backward_dx(N, C, S, P(xg), P(og), P(dg), P(mgg), P(vgg), P(gpu(w)), P(gpu(er)), P(gpu(eyr)), P(dxg), P(drg), P(dwg), P(dbg), 1e-5, 1, None)
    mul = float(((w.abs() + 1e-5) / torch.sqrt(vr + 1e-5)).max())
    close(dxg, dxr, 1e-4, "dx", floor=float(dout.abs().max()) * mul)
    close(dwg, dwr, 5e-5, "dweight", floor=1e-6); close(dbg, dbr, 5e-5, "dbias", floor=1e-6)
    if with_res:
        assert torch.equal(drg.cpu(), drr), "dres = dout * (out > 0), exact"


@pytest.mark.parametrize("rows,C", [(35, 8), (8 * 65 * 65, 256), (2 * 129 * 129, 64), (1000, 2048), (7, 20), (4225, 48)])
@pytest.mark.parametrize("act", [0, 1, 3])
def test_abn_apply_nhwc(hip, ref, rows, C, act):
    """Channels-last inference BN -> (+residual) -> activation (the frozen teacher's layout)."""
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 3 + torch.randn(1, C, generator=g) * 5
    r = torch.randn(rows, C, generator=g)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    for res in (None, r):
        xr, xg = x.clone(), gpu(x)
        assert ref.skd_abn_apply_nhwc(rows, C, P(xr), P(res), P(rm), P(rv), P(w), P(b), 1e-5, act, 0.01, None)
        assert hip.skd_abn_apply_nhwc(rows, C, P(xg), P(gpu(res)), P(gpu(rm)), P(gpu(rv)), P(gpu(w)), P(gpu(b)), 1e-5, act, 0.01, None)
        close(xg, xr, 2e-5, "nhwc apply")
    assert hip.skd_abn_apply_nhwc(rows, 6, P(gpu(x)), None, P(gpu(rm)), P(gpu(rv)), None, None, 1e-5, 0, 0.01, None) == 0   # C % 4


@pytest.mark.parametrize("rows,C", [(8 * 65 * 65, 128), (8 * 129 * 129, 64), (2 * 33 * 33, 512), (8 * 36, 128), (50, 4), (1000, 1024),
                                    (8 * 65 * 65, 512), (8 * 65 * 65, 256), (4 * 256 * 256, 64), (2, 128), (8, 128), (300000, 8)])
@pytest.mark.parametrize("act", [0, 1])
def test_abn_nhwc_training(hip, ref, rows, C, act):
    """Channels-last training entries vs the C oracle: in-place ABN (act none / leaky) and the fused BN+ReLU form."""
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 3 + torch.randn(1, C, generator=g) * 5
    r = torch.randn(rows, C, generator=g)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    w[0], w[1] = 0.0, -abs(float(w[1]))
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    ws_r = torch.empty(max(1, ref.skd_abn_nhwc_workspace_floats(rows, C)))
    ws_g = torch.empty(max(1, hip.skd_abn_nhwc_workspace_floats(rows, C)), device=DEV)
    # in place, activation `act`
    zr, mr, vr, rmr, rvr = x.clone(), torch.empty(C), torch.empty(C), rm.clone(), rv.clone()
    assert ref.skd_abn_forward_train_nhwc(rows, C, P(zr), None, P(zr), P(w), P(b), P(rmr), P(rvr), P(mr), P(vr), 0.1, 1e-5, act, 0.01, P(ws_r), None)
    zg, mg, vg, rmg, rvg = gpu(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV), gpu(rm), gpu(rv)
    assert hip.skd_abn_forward_train_nhwc(rows, C, P(zg), None, P(zg), P(gpu(w)), P(gpu(b)), P(rmg), P(rvg), P(mg), P(vg), 0.1, 1e-5, act, 0.01, P(ws_g), None)
    close(mg, mr, 2e-5, "mean"); close(vg, vr, 5e-5, "var"); close(rmg, rmr, 2e-6, "running_mean"); close(rvg, rvr, 1e-5, "running_var")
    close(zg, zr, 3e-5, "z")
    m2, v2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    assert hip.skd_abn_stats_nhwc(rows, C, P(gpu(x)), P(m2), P(v2), P(ws_g), None)
    close(m2, mr, 2e-5, "stats mean"); close(v2, vr, 5e-5, "stats var")
    dz = torch.randn(rows, C, generator=g)
    er, eyr, eg, eyg = torch.empty(C), torch.empty(C), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    assert ref.skd_abn_backward_reduce_nhwc(rows, C, P(zr), P(dz), P(w), P(b), P(er), P(eyr), 1e-5, act, 0.01, P(ws_r), None)
    assert hip.skd_abn_backward_reduce_nhwc(rows, C, P(gpu(zr)), P(gpu(dz)), P(gpu(w)), P(gpu(b)), P(eg), P(eyg), 1e-5, act, 0.01, P(ws_g), None)
    close(eg, er, 5e-5, "edz"); close(eyg, eyr, 5e-5, "eydz", floor=float(eyr.abs().max()) * 1e-1)
    dxr, dwr, dbr = torch.empty_like(x), torch.zeros(C), torch.zeros(C)
    assert ref.skd_abn_backward_dx_nhwc(rows, C, P(zr), P(dz), P(vr), P(w), P(b), P(er), P(eyr), P(dxr), P(dwr), P(dbr), 1e-5, act, 0.01, 1, None)
    dxg, dwg, dbg = torch.empty(rows, C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    assert hip.skd_abn_backward_dx_nhwc(rows, C, P(gpu(zr)), P(gpu(dz)), P(gpu(vr)), P(gpu(w)), P(gpu(b)), P(gpu(er)), P(gpu(eyr)), P(dxg), P(dwg), P(dbg), 1e-5, act, 0.01, 1, None)
    mul = float(((w.abs() + 1e-5) / torch.sqrt(vr + 1e-5)).max())
    close(dxg, dxr, 1e-4, "dx", floor=float(dz.abs().max()) * mul)
    close(dwg, dwr, 5e-5, "dweight", floor=1e-6); close(dbg, dbr, 5e-5, "dbias", floor=1e-6)
    # fused BN -> (+ residual) -> ReLU, out of place
    for res in (None, r):
        outr, outg = torch.empty_like(x), torch.empty(rows, C, device=DEV)
        assert ref.skd_abn_forward_train_nhwc(rows, C, P(x), P(res), P(outr), P(w), P(b), None, None, P(mr), P(vr), 0.1, 1e-5, 3, 0.0, P(ws_r), None)
        xg = gpu(x)
        assert hip.skd_abn_forward_train_nhwc(rows, C, P(xg), P(gpu(res)), P(outg), P(gpu(w)), P(gpu(b)), None, None, P(mg), P(vg), 0.1, 1e-5, 3, 0.0, P(ws_g), None)
        close(outg, outr, 3e-5, "relu out")
        assert torch.equal(xg.cpu(), x)
        out2 = torch.empty(rows, C, device=DEV)
        assert hip.skd_abn_apply_nhwc_to(rows, C, P(xg), P(gpu(res)), P(out2), P(gpu(mr)), P(gpu(vr)), P(gpu(w)), P(gpu(b)), 1e-5, 3, 0.0, None)
        close(out2, outr, 3e-5, "apply_to")
        assert ref.skd_abn_relu_backward_reduce_nhwc(rows, C, P(x), P(outr), P(dz), P(mr), P(vr), P(er), P(eyr), 1e-5, P(ws_r), None)
        assert hip.skd_abn_relu_backward_reduce_nhwc(rows, C, P(xg), P(gpu(outr)), P(gpu(dz)), P(gpu(mr)), P(gpu(vr)), P(eg), P(eyg), 1e-5, P(ws_g), None)
        close(eg, er, 5e-5, "relu edz"); close(eyg, eyr, 5e-5, "relu eydz")
        drr = torch.empty_like(x) if res is not None else None
        drg = torch.empty(rows, C, device=DEV) if res is not None else None
        dwr.zero_(); dbr.zero_(); dwg.zero_(); dbg.zero_()
        assert ref.skd_abn_relu_backward_dx_nhwc(rows, C, P(x), P(outr), P(dz), P(mr), P(vr), P(w), P(er), P(eyr), P(dxr), P(drr), P(dwr), P(dbr), 1e-5, 1, None)
        assert hip.skd_abn_relu_backward_dx_nhwc(rows, C, P(xg), P(gpu(outr)), P(gpu(dz)), P(gpu(mr)), P(gpu(vr)), P(gpu(w)), P(gpu(er)), P(gpu(eyr)), P(dxg), P(drg), P(dwg), P(dbg), 1e-5, 1, None)
        close(dxg, dxr, 1e-4, "relu dx", floor=float(dz.abs().max()) * mul)
        close(dwg, dwr, 5e-5, "relu dweight", floor=1e-6); close(dbg, dbr, 5e-5, "relu dbias", floor=1e-6)
        if res is not None:
            assert torch.equal(drg.cpu(), drr)
        else:
            # forward without residual: the `_x` entries recompute the ReLU mask from x with the forward pass's own
            # expression -- exactly the mask of the forward's output (out2), so dx is bit-identical to the entry that reads it
            dzg, mrg, vrg, wg_, bg_ = gpu(dz), gpu(mr), gpu(vr), gpu(w), gpu(b)
            ea, eya, eb, eyb = (torch.empty(C, device=DEV) for _ in range(4))
            assert hip.skd_abn_relu_backward_reduce_nhwc(rows, C, P(xg), P(out2), P(dzg), P(mrg), P(vrg), P(ea), P(eya), 1e-5, P(ws_g), None)
            assert hip.skd_abn_relu_backward_reduce_nhwc_x(rows, C, P(xg), P(dzg), P(mrg), P(vrg), P(wg_), P(bg_), P(eb), P(eyb), 1e-5, P(ws_g), None)
            close(eb, ea, 2e-6, "edz, mask from x", floor=1e-3); close(eyb, eya, 2e-6, "eydz, mask from x", floor=1e-3)
            dxa, dxb = torch.empty(rows, C, device=DEV), torch.full((rows, C), float("nan"), device=DEV)
            dwa, dba, dwb, dbb = (torch.full((C,), float("nan"), device=DEV) for _ in range(4))
            assert hip.skd_abn_relu_backward_dx_nhwc(rows, C, P(xg), P(out2), P(dzg), P(mrg), P(vrg), P(wg_), P(ea), P(eya), P(dxa), None, P(dwa), P(dba), 1e-5, 0, None)
            assert hip.skd_abn_relu_backward_dx_nhwc_x(rows, C, P(xg), P(dzg), P(mrg), P(vrg), P(wg_), P(bg_), P(ea), P(eya), P(dxb), P(dwb), P(dbb), 1e-5, 0, None)
            assert torch.equal(dxa, dxb) and torch.equal(dwa, dwb) and torch.equal(dba, dbb)
            assert ref.skd_abn_relu_backward_reduce_nhwc_x(rows, C, P(x), P(dz), P(mr), P(vr), P(w), P(b), P(er), P(eyr), 1e-5, P(ws_r), None)
            close(eb, er, 5e-5, "edz, mask from x vs oracle"); close(eyb, eyr, 5e-5, "eydz, mask from x vs oracle")
    assert hip.skd_abn_nhwc_workspace_floats(100, 48) == 0     # not a power of two: caller must use NCHW


@pytest.mark.parametrize("rows,C", [(8 * 65 * 65, 128), (8 * 129 * 129, 64), (8 * 65 * 65, 256), (8 * 65 * 65, 512), (2 * 33 * 33, 512),
                                    (8 * 36, 128), (50, 4), (2, 128), (300000, 8), (4 * 256 * 256, 64)])
def test_abn_nhwc_one_call_backward(hip, ref, rows, C):
    """skd_abn_backward_nhwc / skd_abn_relu_backward_nhwc (reduce + dx in one call; ONE register-resident launch for the
    sizes that fit, the last two shapes and C = 512 fall back to two launches) vs the C oracle and vs the two separate
    entries: every output incl. edz / eydz, dres, written (not accumulated) dweight / dbias."""
    g = torch.Generator().manual_seed(rows * 3 + C)
    x = torch.randn(rows, C, generator=g) * 3 + torch.randn(1, C, generator=g) * 5
    r, dz = torch.randn(rows, C, generator=g), torch.randn(rows, C, generator=g)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    w[0], w[1] = 0.0, -abs(float(w[1]))
    ws_r = torch.empty(max(1, ref.skd_abn_nhwc_workspace_floats(rows, C)))
    ws_g = torch.empty(max(1, hip.skd_abn_nhwc_workspace_floats(rows, C)), device=DEV)
    nan = lambda *shape: torch.full(shape, float("nan"), device=DEV)
    # ---- in-place ABN (leaky): z from the oracle's forward ----
    zr, mr, vr = x.clone(), torch.empty(C), torch.empty(C)
    assert ref.skd_abn_forward_train_nhwc(rows, C, P(zr), None, P(zr), P(w), P(b), None, None, P(mr), P(vr), 0.1, 1e-5, 1, 0.01, P(ws_r), None)
    st_r, dxr, dwr, dbr = torch.empty(2, C), torch.empty_like(x), torch.empty(C), torch.empty(C)
    assert ref.skd_abn_backward_nhwc(rows, C, P(zr), P(dz), P(vr), P(w), P(b), P(st_r[0]), P(st_r[1]), P(dxr), P(dwr), P(dbr), 1e-5, 1, 0.01, 0, P(ws_r), None)
    st_g, dxg, dwg, dbg = nan(2, C), nan(rows, C), nan(C), nan(C)
    zg, dzg, vg, wg_, bg_ = gpu(zr), gpu(dz), gpu(vr), gpu(w), gpu(b)
    assert hip.skd_abn_backward_nhwc(rows, C, P(zg), P(dzg), P(vg), P(wg_), P(bg_), P(st_g[0]), P(st_g[1]), P(dxg), P(dwg), P(dbg), 1e-5, 1, 0.01, 0, P(ws_g), None)
    mul = float(((w.abs() + 1e-5) / torch.sqrt(vr + 1e-5)).max())
    close(st_g[0], st_r[0], 5e-5, "edz"); close(st_g[1], st_r[1], 5e-5, "eydz", floor=float(st_r[1].abs().max()) * 1e-1)
    close(dxg, dxr, 1e-4, "dx", floor=float(dz.abs().max()) * mul)
    close(dwg, dwr, 5e-5, "dweight", floor=float(dwr.abs().max()) * 1e-2 + 1e-6); close(dbg, dbr, 5e-5, "dbias", floor=float(dbr.abs().max()) * 1e-2 + 1e-6)
    assert torch.equal(zg.cpu(), zr) and torch.equal(dzg.cpu(), dz)               # inputs untouched
    # the two separate entries on the same inputs: same maths, another summation order
    e2, ey2, dx2, dw2, db2 = nan(C), nan(C), nan(rows, C), nan(C), nan(C)
    assert hip.skd_abn_backward_reduce_nhwc(rows, C, P(zg), P(dzg), P(wg_), P(bg_), P(e2), P(ey2), 1e-5, 1, 0.01, P(ws_g), None)
    assert hip.skd_abn_backward_dx_nhwc(rows, C, P(zg), P(dzg), P(vg), P(wg_), P(bg_), P(e2), P(ey2), P(dx2), P(dw2), P(db2), 1e-5, 1, 0.01, 0, None)
    close(st_g[0], e2, 2e-5, "edz vs two launches", floor=1e-3); close(dxg, dx2, 2e-5, "dx vs two launches", floor=float(dz.abs().max()) * mul)
    # accumulate = 1 adds to dweight / dbias
    dwa, dba = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    assert hip.skd_abn_backward_nhwc(rows, C, P(zg), P(dzg), P(vg), P(wg_), P(bg_), P(st_g[0]), P(st_g[1]), P(dxg), P(dwa), P(dba), 1e-5, 1, 0.01, 1, P(ws_g), None)
    close(dwa, dwg + 1.0, 1e-6, "accumulated dweight"); close(dba, dbg + 1.0, 1e-6, "accumulated dbias")     # (e * n + old may be one fma)
    # ---- fused BN -> (+ residual) -> ReLU ----
    for res in (None, r):
        outr = torch.empty_like(x)
        assert ref.skd_abn_forward_train_nhwc(rows, C, P(x), P(res), P(outr), P(w), P(b), None, None, P(mr), P(vr), 0.1, 1e-5, 3, 0.0, P(ws_r), None)
        xg, outg, mg, vg = gpu(x), gpu(outr), gpu(mr), gpu(vr)
        for out_r, out_g in ((outr, outg),) + (((None, None),) if res is None else ()):
            drr = torch.empty_like(x) if res is not None else None
            drg = nan(rows, C) if res is not None else None
            assert ref.skd_abn_relu_backward_nhwc(rows, C, P(x), P(out_r), P(dz), P(mr), P(vr), P(w), P(b), P(st_r[0]), P(st_r[1]), P(dxr), P(drr), P(dwr), P(dbr), 1e-5, 0, P(ws_r), None)
            st_g, dxg, dwg, dbg = nan(2, C), nan(rows, C), nan(C), nan(C)
            assert hip.skd_abn_relu_backward_nhwc(rows, C, P(xg), P(out_g), P(dzg), P(mg), P(vg), P(wg_), P(bg_), P(st_g[0]), P(st_g[1]), P(dxg), P(drg), P(dwg), P(dbg), 1e-5, 0, P(ws_g), None)
            close(st_g[0], st_r[0], 5e-5, "relu edz"); close(st_g[1], st_r[1], 5e-5, "relu eydz")
            close(dxg, dxr, 1e-4, "relu dx", floor=float(dz.abs().max()) * mul)
            close(dwg, dwr, 5e-5, "relu dweight", floor=float(dwr.abs().max()) * 1e-2 + 1e-6); close(dbg, dbr, 5e-5, "relu dbias", floor=float(dbr.abs().max()) * 1e-2 + 1e-6)
            if res is not None:
                assert torch.equal(drg.cpu(), drr)
    assert hip.skd_abn_relu_backward_nhwc(rows, C, P(xg), None, P(dzg), P(mg), P(vg), P(wg_), P(bg_), P(st_g[0]), P(st_g[1]), P(dxg), P(dxg), P(dwg), P(dbg), 1e-5, 0, P(ws_g), None) == 0


@pytest.mark.parametrize("rows,C", [(8 * 65 * 65, 256), (8 * 65 * 65, 128), (8 * 129 * 129, 64), (2 * 33 * 33, 512)])
def test_abn_one_launch_passes_are_repeatable_under_a_busy_second_stream(hip, rows, C):
    """The register-resident one-launch passes hand statistics from a channel block's last arriver to its other workgroups
    through memory (write-through stores, generation word, L1-bypassing loads).  A stale read would be rare and silent, so:
    200 forward and 200 backward launches on the same inputs while a second stream keeps the CUs busy with GEMMs of varying
    size (changing which workgroups are resident when) -- every launch must reproduce the first one's bits, and no NaN
    (the time-out poison) may appear."""
    g = torch.Generator().manual_seed(rows + 7 * C)
    x = gpu(torch.randn(rows, C, generator=g) * 2 + torch.randn(1, C, generator=g) * 3)
    r, dz = gpu(torch.randn(rows, C, generator=g)), gpu(torch.randn(rows, C, generator=g))
    w, b = gpu(torch.randn(C, generator=g)), gpu(torch.randn(C, generator=g))
    ws = torch.empty(max(1, hip.skd_abn_nhwc_workspace_floats(rows, C)), device=DEV)
    side = torch.cuda.Stream(device=DEV)
    mats = [torch.randn(n, n, device=DEV) for n in (512, 1024, 2048, 3072)]
    first = None
    for it in range(200):
        with torch.cuda.stream(side):
            m = mats[it % 4]
            for _ in range(1 + it % 3):
                m = (m @ mats[it % 4]) * 1e-3
        out, st = torch.full((rows, C), float("nan"), device=DEV), torch.full((2, C), float("nan"), device=DEV)
        assert hip.skd_abn_forward_train_nhwc(rows, C, P(x), P(r), P(out), P(w), P(b), None, None, P(st[0]), P(st[1]), 0.1, 1e-5, 3, 0.0, P(ws), None)
        gst, dx, dres = torch.full((2, C), float("nan"), device=DEV), torch.full((rows, C), float("nan"), device=DEV), torch.full((rows, C), float("nan"), device=DEV)
        dw, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        assert hip.skd_abn_relu_backward_nhwc(rows, C, P(x), P(out), P(dz), P(st[0]), P(st[1]), P(w), P(b), P(gst[0]), P(gst[1]), P(dx), P(dres),
                                              P(dw), P(db), 1e-5, 0, P(ws), None)
        cur = (out, st, gst, dx, dres, dw, db)
        if first is None:
            torch.cuda.synchronize()
            first = cur
            assert all(bool(torch.isfinite(t).all()) for t in cur)
        else:
            for name, a, c in zip(("out", "mean/var", "edz/eydz", "dx", "dres", "dweight", "dbias"), first, cur):
                assert torch.equal(a, c), "launch %d: %s differs from the first launch" % (it, name)
    torch.cuda.synchronize()


def test_fused_abn_grid_cap_and_device_status_words(hip):
    """include/skd.h section 13: the one-launch passes size their grid barrier by what the DEVICE can hold (its
    compute-unit count, queried -- not the constant 256), a caller can lower the cap (ranks sharing a device), a tensor that
    no longer fits takes the two-launch path with the same numbers; the device-raised error words exist and are clear."""
    import ctypes
    n = hip.skd_status_words()
    words = (ctypes.c_uint * n)()
    assert n >= 2 and hip.skd_status_read(ctypes.cast(words, ctypes.c_void_p)) and not any(words)
    assert _lib.device_status() == [0] * n
    _lib.raise_on_device_errors()                        # nothing raised
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert hip.skd_abn_set_fused_max_workgroups(0) == min(256, cus)
    rows, C = 8 * 65 * 65, 256
    g = torch.Generator().manual_seed(5)
    x = gpu(torch.randn(rows, C, generator=g) * 3 + 1)
    dz = gpu(torch.randn(rows, C, generator=g))
    w, b = gpu(torch.randn(C, generator=g)), gpu(torch.randn(C, generator=g))
    ws = torch.empty(hip.skd_abn_nhwc_workspace_floats(rows, C), device=DEV)

    def run():
        z, st = x.clone(), torch.empty(2, C, device=DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        assert hip.skd_abn_forward_train_nhwc(rows, C, P(z), None, P(z), P(w), P(b), P(rm), P(rv), P(st[0]), P(st[1]), 0.1, 1e-5, 1, 0.01, P(ws), None)
        e, dx, dw, db = torch.empty(2, C, device=DEV), torch.empty(rows, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        assert hip.skd_abn_backward_nhwc(rows, C, P(z), P(dz), P(st[1]), P(w), P(b), P(e[0]), P(e[1]), P(dx), P(dw), P(db), 1e-5, 1, 0.01, 0, P(ws), None)
        torch.cuda.synchronize()
        return z, st, rm, rv, e, dx, dw, db

    whole = run()
    try:
        assert hip.skd_abn_set_fused_max_workgroups(16) == 16      # 16 workgroups x 1024 threads x 17 rows cannot hold 33800 x 256
        capped = run()
    finally:
        assert hip.skd_abn_set_fused_max_workgroups(0) == min(256, cus)
    for a, c, name in zip(whole, capped, ("z", "stat", "rm", "rv", "e", "dx", "dw", "db")):
        close(c, a, 2e-5, name + " (grid cap 16 -> two-launch path)", floor=float(a.abs().max()) * 1e-2)
    assert _lib.device_status() == [0] * n


def test_operational_switches_of_the_one_launch_abn_passes(hip, monkeypatch):
    """SKD_ABN_FUSED=0 (a device shared with another grid-barrier launch: INTEGRATION.md) sends the same calls down the two-launch
    passes -- same numbers; SKD_ABN_FUSED_MAXWG caps the one-launch grids of a process from its environment (read once per device,
    hence a fresh process)."""
    import subprocess
    import sys
    rows, C = 8 * 65 * 65, 128
    g = torch.Generator().manual_seed(6)
    x = gpu(torch.randn(rows, C, generator=g) * 3 + 1)
    dz = gpu(torch.randn(rows, C, generator=g))
    w, b = gpu(torch.randn(C, generator=g)), gpu(torch.randn(C, generator=g))
    ws = torch.empty(hip.skd_abn_nhwc_workspace_floats(rows, C), device=DEV)

    def run():
        z, st = x.clone(), torch.empty(2, C, device=DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        assert hip.skd_abn_forward_train_nhwc(rows, C, P(z), None, P(z), P(w), P(b), P(rm), P(rv), P(st[0]), P(st[1]), 0.1, 1e-5, 1, 0.01, P(ws), None)
        e, dx, dw, db = torch.empty(2, C, device=DEV), torch.empty(rows, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        assert hip.skd_abn_backward_nhwc(rows, C, P(z), P(dz), P(st[1]), P(w), P(b), P(e[0]), P(e[1]), P(dx), P(dw), P(db), 1e-5, 1, 0.01, 0, P(ws), None)
        torch.cuda.synchronize()
        return z, st, rm, rv, e, dx, dw, db

    # (library state since round 6: the environment variable is only the DEFAULT, read once; include/skd.h section 13)
    assert hip.skd_abn_set_fused(1) and hip.skd_abn_get_fused() == 1
    one = run()
    assert hip.skd_abn_set_fused(0) and hip.skd_abn_get_fused() == 0
    try:
        two = run()
    finally:
        hip.skd_abn_set_fused(-1)
    for a, c, name in zip(one, two, ("z", "stat", "rm", "rv", "e", "dx", "dw", "db")):
        close(c, a, 2e-5, name + " (SKD_ABN_FUSED=0 -> two-launch passes)", floor=float(a.abs().max()) * 1e-2)
    code = ("from structure_knowledge_distillation_amd import _lib; import torch; torch.zeros(1, device='cuda'); "
            "print('CAP', _lib.load().skd_abn_set_fused_max_workgroups(-1))")
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SKD_ABN_FUSED_MAXWG="24"), cwd=ROOT, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0 and "CAP 24" in res.stdout, (res.stdout[-300:], res.stderr[-600:])


def test_abn_single_sample_running_var_is_finite(hip):
    """One sample per channel (PSP 1x1 stage at batch 1, one replica): the reference's n / (n - 1) poisons
    running_var with NaN; here the biased variance (0) is kept -- DESIGN.md section 7."""
    C = 128
    for nhwc in (True, False):
        x = torch.randn(1, C, device=DEV)
        w, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        m, v = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        if nhwc:
            ws = torch.empty(hip.skd_abn_nhwc_workspace_floats(1, C), device=DEV)
            assert hip.skd_abn_forward_train_nhwc(1, C, P(x), None, P(x), P(w), P(b), P(rm), P(rv), P(m), P(v), 0.1, 1e-5, 1, 0.01, P(ws), None)
        else:
            ws = torch.empty(hip.skd_abn_workspace_floats(1, C, 1), device=DEV)
            assert hip.skd_abn_forward_train(1, C, 1, P(x), P(w), P(b), P(rm), P(rv), P(m), P(v), 0.1, 1e-5, 1, 0.01, P(ws), None)
        assert torch.isfinite(rv).all() and torch.allclose(rv.cpu(), torch.full((C,), 0.9)), nhwc
        assert torch.isfinite(x).all()


@pytest.mark.parametrize("rows,C", [(8 * 65 * 65, 512), (8 * 129 * 129, 64), (8 * 65 * 65, 128), (3000, 256)])
def test_abn_nhwc_reduction_handoff_stress(hip, rows, C):
    """The one-launch reductions hand their workgroup partials to the last-arriving workgroup through global memory
    (release -> ticket -> acquire).  The summation order is fixed, so results must be BIT-identical from launch to
    launch; a stale partial (a missed release / acquire, a counter that was not re-armed) would show up as a
    different value.  Alternate two inputs over ONE workspace, uneven load from a concurrent stream."""
    g = torch.Generator().manual_seed(C)
    xs = [gpu(torch.randn(rows, C, generator=g) * (2 + i) + i) for i in range(2)]
    dz = gpu(torch.randn(rows, C, generator=g))
    w, b = gpu(torch.randn(C, generator=g)), gpu(torch.randn(C, generator=g))
    ws = torch.empty(hip.skd_abn_nhwc_workspace_floats(rows, C), device=DEV)
    side = torch.cuda.Stream()
    noise = torch.randn(1 << 22, device=DEV)
    first = {}
    for it in range(60):
        i = it & 1
        if it % 7 == 0:
            with torch.cuda.stream(side):       # uneven load on the chip while the hand-off happens
                for _ in range(4):
                    noise.mul_(1.0001)
        m, v, e, ey = (torch.empty(C, device=DEV) for _ in range(4))
        assert hip.skd_abn_stats_nhwc(rows, C, P(xs[i]), P(m), P(v), P(ws), None)
        assert hip.skd_abn_backward_reduce_nhwc(rows, C, P(xs[i]), P(dz), P(w), P(b), P(e), P(ey), 1e-5, 1, 0.01, P(ws), None)
        e2, ey2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        assert hip.skd_abn_relu_backward_reduce_nhwc(rows, C, P(xs[i]), P(xs[1 - i]), P(dz), P(m), P(v), P(e2), P(ey2), 1e-5, P(ws), None)
        got = [t.clone() for t in (m, v, e, ey, e2, ey2)]
        if i not in first:
            first[i] = got
            xd = xs[i].double()
            close(m, xd.mean(0).float(), 2e-5, "mean"); close(v, xd.var(0, unbiased=False).float(), 5e-5, "var")
        else:
            for a, bb in zip(got, first[i]):
                assert torch.equal(a, bb), "reduction result changed between launches (iteration %d)" % it
    torch.cuda.synchronize()


@pytest.mark.parametrize("G,C", [(2, 6), (8, 512), (1, 64), (3, 1000)])
def test_abn_combine_stats(hip, ref, G, C):
    """Cross-replica statistics combine (functions.py:196-197) + running update with n = count * G."""
    g = torch.Generator().manual_seed(G * C)
    gathered = torch.cat([torch.randn(G, 1, C, generator=g), torch.rand(G, 1, C, generator=g) + 0.1], 1).contiguous()
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    mr, vr, rmr, rvr = torch.empty(C), torch.empty(C), rm.clone(), rv.clone()
    assert ref.skd_abn_combine_stats(G, C, P(gathered), None, 0, P(mr), P(vr), P(rmr), P(rvr), 0.1, float(4225 * 8 * G), None)
    mg, vg, rmg, rvg = torch.empty(C, device=DEV), torch.empty(C, device=DEV), gpu(rm), gpu(rv)
    assert hip.skd_abn_combine_stats(G, C, P(gpu(gathered)), None, 0, P(mg), P(vg), P(rmg), P(rvg), 0.1, float(4225 * 8 * G), None)
    close(mg, mr, 1e-6, "mean"); close(vg, vr, 1e-6, "var"); close(rmg, rmr, 1e-6, "running_mean"); close(rvg, rvr, 1e-6, "running_var")
    want_m = gathered[:, 0].double().mean(0)
    want_v = (gathered[:, 1].double() + (want_m - gathered[:, 0].double()) ** 2).mean(0)
    close(mg, want_m.float(), 1e-6, "mean vs formula"); close(vg, want_v.float(), 1e-6, "var vs formula")
    assert hip.skd_abn_combine_stats(G, C, P(gpu(gathered)), None, 0, P(mg), P(vg), None, None, 0.1, 100.0, None)   # no running buffers
    # unequal shards: weights w_g = n_g / sum(n) give the pooled statistics of the concatenated data exactly
    counts = torch.arange(1, G + 1, dtype=torch.float64) * 7
    xs = [torch.randn(int(n), C, generator=g, dtype=torch.float64) * (1 + i) + i for i, n in enumerate(counts)]
    gathered = torch.stack([torch.stack([x.mean(0), x.var(0, unbiased=False)]) for x in xs]).float().contiguous()
    w = (counts / counts.sum()).float()
    allx = torch.cat(xs)
    rank = G - 1
    rmg, rvg, rmr, rvr = gpu(rm), gpu(rv), rm.clone(), rv.clone()
    assert hip.skd_abn_combine_stats(G, C, P(gpu(gathered)), P(gpu(w)), rank, P(mg), P(vg), P(rmg), P(rvg), 0.1, float(counts[rank]), None)
    assert ref.skd_abn_combine_stats(G, C, P(gathered), P(w), rank, P(mr), P(vr), P(rmr), P(rvr), 0.1, float(counts[rank]), None)
    close(mg, allx.mean(0).float(), 2e-6, "pooled mean", floor=1.0); close(vg, allx.var(0, unbiased=False).float(), 5e-6, "pooled var")
    close(mg, mr, 1e-6, "weighted mean vs oracle"); close(vg, vr, 1e-6, "weighted var vs oracle"); close(rvg, rvr, 1e-6, "weighted running_var")
    n_tot = float(counts.sum())
    close(rvg, (rv * 0.9 + 0.1 * allx.var(0, unbiased=False).float() * n_tot / (n_tot - 1)), 1e-5, "running_var with the pooled count")
    assert hip.skd_abn_combine_stats(G, C, P(gpu(gathered)), P(gpu(w)), G, P(mg), P(vg), None, None, 0.1, 1.0, None) == 0       # rank out of range


def test_abn_reference_corner_semantics(hip, ref):
    """bn.cu:148-151: invStd = 0 when var == 0 and eps == 0;  bn.cu:153: gamma = 1 / beta = 0 without affine
    parameters (NULL pointers), forward and backward;  constant input (var == 0) with the default eps."""
    N, C, S = 2, 4, 33
    x = torch.randn(N, C, S)
    mean, var = torch.zeros(C), torch.tensor([0.0, 1.0, 0.0, 2.0])
    zr, zg = x.clone(), gpu(x)
    assert ref.skd_abn_apply(N, C, S, P(zr), P(mean), P(var), None, None, 0.0, 0, 0.01, None)
    assert hip.skd_abn_apply(N, C, S, P(zg), P(gpu(mean)), P(gpu(var)), None, None, 0.0, 0, 0.01, None)
    close(zg, zr, 1e-6, "eps = 0")
    assert float(zg[:, 0].abs().max()) == 0.0 and float(zg[:, 2].abs().max()) == 0.0     # invStd = 0 -> y = 0
    # non-affine training forward + backward through the fused entries
    xr, xg = x.clone(), gpu(x)
    mr, vr, mg, vg = torch.empty(C), torch.empty(C), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ws = torch.empty(max(1, hip.skd_abn_workspace_floats(N, C, S)), device=DEV)
    assert ref.skd_abn_forward_train(N, C, S, P(xr), None, None, None, None, P(mr), P(vr), 0.1, 1e-5, 1, 0.01, P(torch.empty(2 * C)), None)
    assert hip.skd_abn_forward_train(N, C, S, P(xg), None, None, None, None, P(mg), P(vg), 0.1, 1e-5, 1, 0.01, P(ws), None)
    close(xg, xr, 3e-5, "non-affine z")
    dz = torch.randn(N, C, S)
    dxr, er, eyr = torch.empty_like(x), torch.empty(C), torch.empty(C)
    assert ref.skd_abn_backward(N, C, S, P(xr), P(dz), P(vr), None, None, P(er), P(eyr), P(dxr), None, None, 1e-5, 1, 0.01, 1, P(torch.empty(2 * C)), None)
    dxg, eg, eyg = torch.empty(N, C, S, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    assert hip.skd_abn_backward(N, C, S, P(gpu(xr)), P(gpu(dz)), P(gpu(vr)), None, None, P(eg), P(eyg), P(dxg), None, None, 1e-5, 1, 0.01, 1, P(ws), None)
    close(dxg, dxr, 1e-4, "non-affine dx", floor=float(dz.abs().max()))
    # constant channel: var == 0 exactly (the shifted one-pass sums are exact zeros), z = beta
    xc = torch.full((N, C, S), 3.25)
    w, b = torch.ones(C), torch.tensor([0.5, -1.0, 2.0, 0.0])
    xcg = gpu(xc)
    assert hip.skd_abn_forward_train(N, C, S, P(xcg), P(gpu(w)), P(gpu(b)), None, None, P(mg), P(vg), 0.1, 1e-5, 0, 0.01, P(ws), None)
    assert float(vg.abs().max()) == 0.0 and torch.allclose(mg.cpu(), torch.full((C,), 3.25))
    assert torch.allclose(xcg.cpu(), b.view(1, C, 1).expand(N, C, S))


def test_abn_legacy_entries(hip, ref):
    """The nine reference exports (libs/src/bn.h:7-19) with their original argument lists."""
    N, C, S = 3, 6, 257
    x, w, b, _, _ = _abn_inputs(N, C, S, seed=11)
    m_r, v_r, m_g, v_g = torch.empty(C), torch.empty(C), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    assert ref.skd_bn_mean_var(N, C, S, P(x), P(m_r), P(v_r), None)
    xg = gpu(x)
    assert hip.skd_bn_mean_var(N, C, S, P(xg), P(m_g), P(v_g), None)
    close(m_g, m_r, 2e-5, "mean"); close(v_g, v_r, 5e-5, "var")
    y_r, z_r = torch.empty_like(x), torch.empty_like(x)
    assert ref.skd_bn_forward(N, C, S, P(x), P(m_r), P(v_r), P(w), P(b), P(y_r), P(z_r), 1e-5, None)
    y_g, z_g = torch.empty_like(xg), torch.empty_like(xg)
    assert hip.skd_bn_forward(N, C, S, P(xg), P(gpu(m_r)), P(gpu(v_r)), P(gpu(w)), P(gpu(b)), P(y_g), P(z_g), 1e-5, None)
    close(y_g, y_r, 2e-5, "y"); close(z_g, z_r, 2e-5, "z")
    dz = torch.randn(N, C, S)
    e_r, ey_r, e_g, ey_g = torch.empty(C), torch.empty(C), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    assert ref.skd_bn_edz_eydz(N, C, S, P(z_r), P(dz), P(w), P(b), P(e_r), P(ey_r), 1e-5, None)
    assert hip.skd_bn_edz_eydz(N, C, S, P(gpu(z_r)), P(gpu(dz)), P(gpu(w)), P(gpu(b)), P(e_g), P(ey_g), 1e-5, None)
    close(e_g, e_r, 5e-5, "edz"); close(ey_g, ey_r, 5e-5, "eydz")
    dx_r, dw_r, db_r = torch.empty_like(x), torch.ones(C), torch.ones(C)      # accumulate (+=) into ones
    assert ref.skd_bn_backward(N, C, S, P(dz), P(z_r), P(v_r), P(w), P(b), P(e_r), P(ey_r), P(dx_r), P(dw_r), P(db_r), 1e-5, None)
    dx_g, dw_g, db_g = torch.empty_like(xg), torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    assert hip.skd_bn_backward(N, C, S, P(gpu(dz)), P(gpu(z_r)), P(gpu(v_r)), P(gpu(w)), P(gpu(b)), P(gpu(e_r)), P(gpu(ey_r)), P(dx_g), P(dw_g), P(db_g), 1e-5, None)
    close(dx_g, dx_r, 5e-5, "dx"); close(dw_g, dw_r, 5e-5, "dweight"); close(db_g, db_r, 5e-5, "dbias")
    n = N * C * S
    for name, extra in (("skd_leaky_relu", (0.01,)), ("skd_elu", ()), ("skd_elu_inv", ())):
        a = (torch.rand(n) - 0.7) if name == "skd_elu_inv" else torch.randn(n)
        ar, ag = a.clone(), gpu(a)
        assert getattr(ref, name)(n, P(ar), *extra, None) and getattr(hip, name)(n, P(ag), *extra, None)
        close(ag, ar, 1e-5, name)
    xx, dd = torch.randn(n), torch.randn(n)
    for name, extra in (("skd_leaky_relu_backward", (0.01,)), ("skd_elu_backward", ())):
        dr, dg = dd.clone(), gpu(dd)
        assert getattr(ref, name)(n, P(xx), P(dr), *extra, None) and getattr(hip, name)(n, P(gpu(xx)), P(dg), *extra, None)
        close(dg, dr, 1e-5, name)
    assert hip.skd_bn_mean_var(0, C, S, None, None, None, None) == 0     # failure -> 0 (bn.cu:245-249)


@pytest.mark.parametrize("N,C,HW", [(1, 1, 1), (2, 19, 33 * 33), (8, 19, 65 * 65), (3, 11, 46 * 61), (2, 40, 100), (1, 2, 70000)])
def test_pixelwise(hip, ref, N, C, HW):
    g = torch.Generator().manual_seed(C)
    s, t = torch.randn(N, C, HW, generator=g) * 4, torch.randn(N, C, HW, generator=g) * 4
    lr, gr = torch.empty(1), torch.empty_like(s)
    assert ref.skd_pixelwise_loss(N, C, HW, P(s), P(t), P(lr), P(gr), P(torch.empty(1)), None)
    lg, gg = torch.empty(1, device=DEV), torch.empty(N, C, HW, device=DEV)
    ws = torch.empty(max(1, hip.skd_pixelwise_workspace_floats(N, HW)), device=DEV)
    assert hip.skd_pixelwise_loss(N, C, HW, P(gpu(s)), P(gpu(t)), P(lg), P(gg), P(ws), None)
    close(lg, lr, 1e-5, "loss"); close(gg, gr, 2e-5, "grad")
    lg2 = torch.empty(1, device=DEV)
    assert hip.skd_pixelwise_loss(N, C, HW, P(gpu(s)), P(gpu(t)), P(lg2), None, P(ws), None)
    assert float(lg2) == float(lg)                                         # deterministic, grad optional
    # independent closed form (criterion.py:223-225)
    want = torch.sum(-torch.softmax(t.double(), 1) * torch.log_softmax(s.double(), 1)) / HW
    close(lg, want.reshape(1).float(), 1e-5, "loss vs torch")


POOL_CASES = [(3, 65, 65, 32, 32), (4, 33, 33, 16, 16), (2, 65, 65, 8, 8), (2, 65, 65, 4, 4), (2, 65, 65, 1, 1),
              (5, 46, 61, 23, 30), (2, 7, 130, 3, 64), (1, 129, 129, 2, 2), (1, 3, 1500, 2, 7)]


@pytest.mark.parametrize("planes,H,W,kh,kw", POOL_CASES)
def test_maxpool_argmax_bit_exact(hip, ref, planes, H, W, kh, kw):
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(planes, H, W, generator=g)
    x[0] = torch.randint(0, 3, (H, W), generator=g).float()     # many ties: first maximum must win
    if planes > 1:
        x[1, H // 2, W // 3] = float("nan")                      # NaN propagates (PyTorch scan rule)
        x[1, 0, 0] = float("nan")
    OH, OW = -(-H // kh), -(-W // kw)
    pr, ir = torch.empty(planes, OH * OW), torch.empty(planes, OH * OW, dtype=torch.int32)
    assert ref.skd_maxpool_argmax(planes, H, W, kh, kw, P(x), P(pr), P(ir), None)
    pg, ig = torch.empty(planes, OH * OW, device=DEV), torch.empty(planes, OH * OW, dtype=torch.int32, device=DEV)
    assert hip.skd_maxpool_argmax(planes, H, W, kh, kw, P(gpu(x)), P(pg), P(ig), None)
    assert torch.equal(ig.cpu(), ir), "argmax indices must be bit-exact"
    assert np.array_equal(pg.cpu().numpy().view(np.uint32), pr.numpy().view(np.uint32)), "pooled values bit-exact"
    # torch's own CPU max-pool agrees with the oracle (ties / NaN included)
    tp, ti = torch.nn.functional.max_pool2d(x[None], (kh, kw), (kh, kw), 0, ceil_mode=True, return_indices=True)
    assert torch.equal(ti[0].reshape(planes, -1).int(), ir)
    # un-pool: dense scatter, every position written once
    dp = torch.randn(planes, OH * OW + 5, generator=g)
    dxr = torch.empty(planes, H, W)
    assert ref.skd_maxunpool_scatter(planes, H, W, kh, kw, P(dp), OH * OW + 5, P(ir), P(dxr), None)
    dxg = torch.full((planes, H, W), 7.0, device=DEV)
    assert hip.skd_maxunpool_scatter(planes, H, W, kh, kw, P(gpu(dp)), OH * OW + 5, P(ig), P(dxg), None)
    assert torch.equal(dxg.cpu(), dxr)


NHWC_POOL_CASES = [(2, 128, 65, 65, 32, 32), (1, 512, 65, 65, 32, 32), (2, 8, 33, 33, 16, 16), (2, 12, 65, 65, 8, 8), (1, 16, 65, 65, 4, 4),
                   (2, 4, 65, 65, 1, 1), (3, 20, 46, 61, 23, 30), (1, 8, 7, 130, 3, 64), (1, 136, 9, 70, 9, 70)]


@pytest.mark.parametrize("B,C,H,W,kh,kw", NHWC_POOL_CASES)
def test_maxpool_argmax_channels_last_bit_exact(hip, ref, B, C, H, W, kh, kw):
    """skd_maxpool_argmax_nhwc / skd_maxunpool_scatter_nhwc (round 5: the PSP features are pooled as they are, channels-last):
    pooled values and argmax indices bit-identical to the planar entries on the NCHW copy of the same data -- ties (first
    maximum wins), NaN (propagates, last one wins), partial border windows, both kernels (window per workgroup / lane per cell)."""
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, C, H, W, generator=g)
    x[0, 0] = torch.randint(0, 3, (H, W), generator=g).float()
    x[0, 1] = 1.0                                                   # a constant plane: the very first position of every window
    x[-1, 2, H // 2, W // 3] = float("nan")
    x[-1, 2, 0, 0] = float("nan")
    x[-1, 3] = float("-inf")
    OH, OW = -(-H // kh), -(-W // kw)
    M = OH * OW
    pr, ir = torch.empty(B * C, M), torch.empty(B * C, M, dtype=torch.int32)
    assert ref.skd_maxpool_argmax(B * C, H, W, kh, kw, P(x), P(pr), P(ir), None)
    xl = x.permute(0, 2, 3, 1).contiguous()                         # (B, H, W, C)
    pr2, ir2 = torch.empty(B * C, M), torch.empty(B * C, M, dtype=torch.int32)
    assert ref.skd_maxpool_argmax_nhwc(B, C, H, W, kh, kw, P(xl), P(pr2), P(ir2), None)
    assert torch.equal(ir2, ir) and np.array_equal(pr2.numpy().view(np.uint32), pr.numpy().view(np.uint32)), "oracle: the two layouts agree"
    pg, ig = torch.full((B * C, M), 7.0, device=DEV), torch.full((B * C, M), -1, dtype=torch.int32, device=DEV)
    xg = gpu(xl)
    assert hip.skd_maxpool_argmax_nhwc(B, C, H, W, kh, kw, P(xg), P(pg), P(ig), None)
    assert torch.equal(ig.cpu(), ir), "argmax indices must be bit-exact"
    assert np.array_equal(pg.cpu().numpy().view(np.uint32), pr.numpy().view(np.uint32)), "pooled values bit-exact"
    pg2 = torch.empty(B * C, M, device=DEV)
    assert hip.skd_maxpool_argmax_nhwc(B, C, H, W, kh, kw, P(xg), P(pg2), None, None)       # index optional
    assert np.array_equal(pg2.cpu().numpy().view(np.uint32), pr.numpy().view(np.uint32))
    dp = torch.randn(B * C, M + 5, generator=g)
    dxr = torch.empty(B, H, W, C)
    assert ref.skd_maxunpool_scatter_nhwc(B, C, H, W, kh, kw, P(dp), M + 5, P(ir), P(dxr), None)
    dxn = torch.empty(B * C, H, W)
    assert ref.skd_maxunpool_scatter(B * C, H, W, kh, kw, P(dp), M + 5, P(ir), P(dxn), None)
    assert torch.equal(dxr, dxn.reshape(B, C, H, W).permute(0, 2, 3, 1)), "oracle: the two layouts agree"
    dxg = torch.full((B, H, W, C), 7.0, device=DEV)
    assert hip.skd_maxunpool_scatter_nhwc(B, C, H, W, kh, kw, P(gpu(dp)), M + 5, P(ig), P(dxg), None)
    assert torch.equal(dxg.cpu(), dxr)
    assert not hip.skd_maxpool_argmax_nhwc(B, 6, H, W, kh, kw, P(xg), P(pg), P(ig), None)   # C must be whole quads


@pytest.mark.parametrize("B,Cs,Ct,M", [(2, 16, 40, 9), (8, 128, 512, 9), (2, 128, 512, 81), (1, 5, 3, 1), (2, 130, 70, 289),
                                        (1, 128, 512, 1089), (3, 8, 8, 128), (2, 12, 20, 129),
                                        (1, 128, 512, 4225)])      # M = 4225: the shape the MFMA roofline claim is quoted on (nt = 34)
def test_pairwise_stages(hip, ref, B, Cs, Ct, M):
    g = torch.Generator().manual_seed(M + Cs)
    ps, pt = torch.randn(B, Cs, M, generator=g), torch.randn(B, Ct, M, generator=g)
    ldm = hip.skd_pairwise_ldm(M)
    assert ldm == ref.skd_pairwise_ldm(M) and ldm % 128 == 0 and ldm >= M
    ldc = -(-Cs // 128) * 128

    def run(lib, to):
        fs, ft = to(torch.full((B, Cs, ldm), 9.0)), to(torch.full((B, Ct, ldm), 9.0))
        fst, nrm = to(torch.full((B, ldm, ldc), 9.0)), to(torch.empty(B, M))
        assert lib.skd_channel_l2_normalise(B, Cs, M, P(to(ps)), P(fs), ldm, P(fst), ldc, P(nrm), None)
        assert lib.skd_channel_l2_normalise(B, Ct, M, P(to(pt)), P(ft), ldm, None, 0, None, None)
        G, loss = to(torch.full((B, ldm, ldm), 9.0)), to(torch.empty(1))
        ws = to(torch.empty(max(1, lib.skd_pairwise_workspace_floats(B, M))))
        assert lib.skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, P(fs), P(ft), P(G), P(loss), P(ws), None)
        gl = to(torch.tensor([0.5]))
        dp = to(torch.full((B, Cs, ldm), 9.0))
        bws = to(torch.empty(max(1, lib.skd_pairwise_backward_workspace_floats(B, Cs, M))))
        assert lib.skd_pairwise_backward(B, Cs, M, ldm, P(fs), P(G), P(nrm), P(gl), P(dp), P(bws), None)
        return [t.cpu() for t in (fs, ft, fst, nrm, G, loss, dp)]

    r = run(ref, lambda t: t.clone())
    h = run(hip, lambda t: t.to(DEV))
    names = ("fhat_s", "fhat_t", "fhat_s_t", "norm", "G", "loss", "dpooled")
    tols = (1e-6, 1e-6, 1e-6, 1e-6, 2e-5, 1e-5, 5e-5)
    for a, b, n, tol in zip(h, r, names, tols):
        if n in ("G", "dpooled"):   # padding rows/cols of the HIP buffers are exact zeros too
            # G = A_T - A_S is a difference of Gram entries of magnitude <= 1
            # (at M = 1 both Grams are exactly 1: G, the loss and dpooled are pure rounding residue)
            close(a[..., :M], b[..., :M], tol, n, floor=1.0 if (n == "G" or M == 1) else 0.0)
            assert float(a[..., M:].abs().max()) == 0.0 if ldm > M else True
        else:
            close(a, b, tol, n, floor=1e-6 if n == "loss" else 0.0)   # M = 1: the loss is a pure cancellation residue
    # against autograd of the reference formula (utils.py:170-183) in fp64
    x = ps.double().requires_grad_(True)
    fh = x / ((x ** 2).sum(1, keepdim=True).sqrt() + 1e-8).detach()
    th = pt.double() / ((pt.double() ** 2).sum(1, keepdim=True).sqrt() + 1e-8)
    L = ((torch.einsum("icm,icn->imn", th, th) - torch.einsum("icm,icn->imn", fh, fh)) ** 2).sum() / M ** 2 / B
    L.backward()
    close(h[5], L.detach().reshape(1).float(), 1e-5, "loss vs autograd", floor=1e-6)
    close(h[6][..., :M], 0.5 * x.grad.float(), 5e-5, "dpooled vs autograd", floor=1.0 if M == 1 else 1e-6)


@pytest.mark.parametrize("B,Cs,Ct,M", [(8, 128, 512, 9), (2, 16, 40, 9), (3, 128, 512, 64), (2, 130, 70, 25), (1, 5, 3, 1), (2, 128, 512, 6)])
def test_pairwise_small_m_fused(hip, ref, B, Cs, Ct, M):
    """The one-launch small-graph entry (reference default: 3 x 3 = 9 nodes) against the C oracle's staged pipeline and
    against fp64 autograd of utils.py:170-183."""
    g = torch.Generator().manual_seed(M * 7 + Cs)
    ps, pt = torch.randn(B, Cs, M, generator=g), torch.randn(B, Ct, M, generator=g)
    lr, dr = torch.empty(1), torch.empty(B, Cs, M)
    assert ref.skd_pairwise_small(B, Cs, Ct, M, P(ps), P(pt), P(lr), P(dr), P(torch.empty(B)), None)
    lg, dg = torch.empty(1, device=DEV), torch.full((B, Cs, M), 9.0, device=DEV)
    ws = torch.empty(B, device=DEV)
    assert hip.skd_pairwise_small(B, Cs, Ct, M, P(gpu(ps)), P(gpu(pt)), P(lg), P(dg), P(ws), None)
    close(lg, lr, 1e-5, "loss", floor=1e-6)
    close(dg, dr, 5e-5, "dpooled", floor=1.0 if M == 1 else 1e-6)
    lg2 = torch.empty(1, device=DEV)
    assert hip.skd_pairwise_small(B, Cs, Ct, M, P(gpu(ps)), P(gpu(pt)), P(lg2), None, P(ws), None)      # loss only
    assert float(lg2) == float(lg)
    x = ps.double().requires_grad_(True)
    fh = x / ((x ** 2).sum(1, keepdim=True).sqrt() + 1e-8).detach()
    th = pt.double() / ((pt.double() ** 2).sum(1, keepdim=True).sqrt() + 1e-8)
    L = ((torch.einsum("icm,icn->imn", th, th) - torch.einsum("icm,icn->imn", fh, fh)) ** 2).sum() / M ** 2 / B
    L.backward()
    close(lg, L.detach().reshape(1).float(), 1e-5, "loss vs autograd", floor=1e-6)
    close(dg, x.grad.float(), 5e-5, "dpooled vs autograd", floor=1.0 if M == 1 else 1e-6)
    assert hip.skd_pairwise_small(B, Cs, Ct, 65, P(gpu(ps)), P(gpu(pt)), P(lg2), None, P(ws), None) == 0  # M > 64: the MFMA path


@pytest.mark.parametrize("h,w", [(64, 304), (128, 1024), (256, 2048), (512, 4096), (7, 5), (1, 1), (33, 1000)])
def test_spectral_norm(hip, ref, h, w):
    g = torch.Generator().manual_seed(h)
    W = torch.randn(h, w, generator=g) * 0.05
    u, v = torch.randn(h, generator=g), torch.randn(w, generator=g)
    u, v = u / u.norm(), v / v.norm()
    ur, vr, sr, wr = u.clone(), v.clone(), torch.empty(1), torch.empty_like(W)
    ug, vg, sg, wg = gpu(u), gpu(v), torch.empty(1, device=DEV), torch.empty(h, w, device=DEV)
    ws = torch.empty(max(1, hip.skd_spectral_workspace_floats(h, w)), device=DEV)
    for it in range(3):   # u, v persist across forwards
        assert ref.skd_spectral_norm_forward(h, w, P(W), P(ur), P(vr), P(sr), P(wr), P(torch.empty(1)), None)
        assert hip.skd_spectral_norm_forward(h, w, P(gpu(W)), P(ug), P(vg), P(sg), P(wg), P(ws), None)
        close(ug, ur, 2e-5, "u"); close(vg, vr, 2e-5, "v"); close(sg, sr, 2e-5, "sigma"); close(wg, wr, 2e-5, "w")
    gw = torch.randn(h, w, generator=g)
    gr, gg = torch.empty_like(W), torch.empty(h, w, device=DEV)
    assert ref.skd_spectral_norm_backward(h, w, P(W), P(ur), P(vr), P(sr), P(gw), P(gr), P(torch.empty(1)), None)
    assert hip.skd_spectral_norm_backward(h, w, P(gpu(W)), P(gpu(ur)), P(gpu(vr)), P(gpu(sr)), P(gpu(gw)), P(gg), P(ws), None)
    close(gg, gr, 5e-5, "grad_w_bar")


@pytest.mark.parametrize("shapes", [[(64, 304), (128, 1024), (256, 2048), (512, 4096)], [(7, 5), (1, 1)], [(33, 1000), (64, 304), (5, 3)]])
def test_spectral_norm_multi_is_bit_identical_to_the_single_layer_entries(hip, shapes):
    """All spectrally normalised layers of the discriminator in 3 + 2 launches (skd_spectral_norm_*_multi): per layer the
    same kernels' arithmetic, so u, v, sigma, w and the gradient have the SAME BITS as the single-layer entries."""
    L = len(shapes)
    g = torch.Generator().manual_seed(L * 100 + shapes[0][0])
    Ws = [gpu(torch.randn(h, w, generator=g) * 0.05) for h, w in shapes]
    u0 = [torch.randn(h, generator=g) for h, _ in shapes]
    v0 = [torch.randn(w, generator=g) for _, w in shapes]
    gws = [gpu(torch.randn(h, w, generator=g)) for h, w in shapes]
    hs, ws_ = (ctypes.c_int * L)(*[h for h, _ in shapes]), (ctypes.c_int * L)(*[w for _, w in shapes])
    arr = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])
    # single-layer entries, two consecutive forwards (u, v persist)
    us, vs = [gpu(u / u.norm()) for u in u0], [gpu(v / v.norm()) for v in v0]
    sig_s, out_s, gb_s = [torch.empty(1, device=DEV) for _ in shapes], [torch.empty_like(W) for W in Ws], [torch.empty_like(W) for W in Ws]
    for k, (h, w) in enumerate(shapes):
        ws = torch.empty(max(1, hip.skd_spectral_workspace_floats(h, w)), device=DEV)
        for _ in range(2):
            assert hip.skd_spectral_norm_forward(h, w, P(Ws[k]), P(us[k]), P(vs[k]), P(sig_s[k]), P(out_s[k]), P(ws), None)
        assert hip.skd_spectral_norm_backward(h, w, P(Ws[k]), P(us[k]), P(vs[k]), P(sig_s[k]), P(gws[k]), P(gb_s[k]), P(ws), None)
    # the multi entries
    um, vm = [gpu(u / u.norm()) for u in u0], [gpu(v / v.norm()) for v in v0]
    sig_m = torch.empty(L, device=DEV)
    sig_l = [sig_m[k:k + 1] for k in range(L)]
    out_m, gb_m = [torch.empty_like(W) for W in Ws], [torch.empty_like(W) for W in Ws]
    work = torch.empty(sum(max(1, hip.skd_spectral_workspace_floats(h, w)) for h, w in shapes), device=DEV)
    for _ in range(2):
        assert hip.skd_spectral_norm_forward_multi(L, hs, ws_, arr(Ws), arr(um), arr(vm), arr(sig_l), arr(out_m), P(work), None)
    assert hip.skd_spectral_norm_backward_multi(L, hs, ws_, arr(Ws), arr(um), arr(vm), arr(sig_l), arr(gws), arr(gb_m), P(work), None)
    torch.cuda.synchronize()
    for k in range(L):
        assert torch.equal(um[k], us[k]) and torch.equal(vm[k], vs[k]) and torch.equal(sig_l[k], sig_s[k]), k
        assert torch.equal(out_m[k], out_s[k]) and torch.equal(gb_m[k], gb_s[k]), k


@pytest.mark.parametrize("geom", [(2, 19, 9, 9, 65, 65), (8, 19, 65, 65, 512, 512), (2, 19, 33, 33, 256, 256), (3, 11, 46, 61, 360, 480),
                                  (1, 3, 1, 1, 4, 4), (1, 40, 5, 7, 5, 7), (2, 21, 17, 9, 100, 3)])
def test_ce_dsn(hip, ref, geom):
    B, C, h, w, H, W = geom
    g = torch.Generator().manual_seed(H + C)
    lm, ld = torch.randn(B, C, h, w, generator=g) * 3, torch.randn(B, C, h, w, generator=g) * 3
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[0, : max(1, H // 16)] = 255
    y[-1, -1, -1] = 255
    lr, gmr, gdr = torch.empty(1), torch.empty_like(lm), torch.empty_like(ld)
    assert ref.skd_ce_dsn_forward(B, C, h, w, H, W, P(lm), P(ld), P(y), 255, 0.4, P(lr), P(gmr), P(gdr), P(torch.empty(8)), None)
    lg, gmg, gdg = torch.empty(1, device=DEV), torch.empty(B, C, h, w, device=DEV), torch.empty(B, C, h, w, device=DEV)
    ws = torch.empty(hip.skd_ce_dsn_workspace_floats(B, C, h, w, H, W), device=DEV)
    yg = y.to(DEV)
    assert hip.skd_ce_dsn_forward(B, C, h, w, H, W, P(gpu(lm)), P(gpu(ld)), P(yg), 255, 0.4, P(lg), P(gmg), P(gdg), P(ws), None)
    close(lg, lr, 1e-5, "loss"); close(gmg, gmr, 5e-5, "grad main"); close(gdg, gdr, 5e-5, "grad dsn")
    # loss only / single head; bit-reproducible
    lg2 = torch.empty(1, device=DEV)
    assert hip.skd_ce_dsn_forward(B, C, h, w, H, W, P(gpu(lm)), P(gpu(ld)), P(yg), 255, 0.4, P(lg2), None, None, P(ws), None)
    assert float(lg2) == float(lg)
    assert hip.skd_ce_dsn_forward(B, C, h, w, H, W, P(gpu(lm)), None, P(yg), 255, 0.4, P(lg2), P(gmg), None, P(ws), None)
    up = torch.nn.functional.interpolate(lm.double(), size=(H, W), mode="bilinear", align_corners=True)
    want = torch.nn.functional.cross_entropy(up, y, ignore_index=255)
    close(lg2, want.reshape(1).float(), 1e-5, "single-head loss vs torch")
    close(gmg, gmr, 5e-5, "single-head grad")
    # every pixel ignored -> 0/0 = NaN, like CrossEntropyLoss(reduction='mean')
    y255 = torch.full((B, H, W), 255, dtype=torch.int64, device=DEV)
    assert hip.skd_ce_dsn_forward(B, C, h, w, H, W, P(gpu(lm)), P(gpu(ld)), P(y255), 255, 0.4, P(lg2), None, None, P(ws), None)
    assert float(lg2) != float(lg2)
    # a label outside [0, C) that is not ignore_index (F.cross_entropy asserts on it): never a silently smaller valid
    # set -- the loss and both gradients of the call come back NaN
    ybad = yg.clone()
    ybad[0, H - 1, W // 2] = C + 3
    assert hip.skd_ce_dsn_forward(B, C, h, w, H, W, P(gpu(lm)), P(gpu(ld)), P(ybad), 255, 0.4, P(lg2), P(gmg), P(gdg), P(ws), None)
    assert float(lg2) != float(lg2) and bool(torch.isnan(gmg).all()) and bool(torch.isnan(gdg).all())
    assert ref.skd_ce_dsn_forward(B, C, h, w, H, W, P(lm), P(ld), P(ybad.cpu()), 255, 0.4, P(lr), P(gmr), P(gdr), P(torch.empty(8)), None)
    assert float(lr) != float(lr)


@pytest.mark.parametrize("geom", [(8, 512, 65, 65, 128), (2, 2048, 65, 65, 512), (2, 7, 33, 33, 5), (3, 4, 46, 61, 3), (1, 2, 7, 9, 2), (1, 1, 129, 129, 1),
                                  (2, 8, 33, 47, 4), (1, 12, 129, 255, 8), (2, 48, 6, 6, 20)])
def test_ppm(hip, ref, geom):
    B, C, H, W, Cout = geom
    sizes = (1, 2, 3, 6)
    arr = (ctypes.c_int * 4)(*sizes)
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g)
    total = hip.skd_ppm_pooled_floats(B * C, 4, arr)
    assert total == ref.skd_ppm_pooled_floats(B * C, 4, arr) == B * C * 50
    pr, pg = torch.empty(total), torch.empty(total, device=DEV)
    assert ref.skd_ppm_pool(B * C, H, W, 4, arr, P(x), P(pr), None)
    assert hip.skd_ppm_pool(B * C, H, W, 4, arr, P(gpu(x)), P(pg), None)
    close(pg, pr, 1e-5, "pooled")
    gp = torch.randn(total, generator=g)
    dxr, dxg = torch.empty_like(x), torch.empty(B, C, H, W, device=DEV)
    assert ref.skd_ppm_pool_backward(B * C, H, W, 4, arr, P(gp), P(dxr), None)
    assert hip.skd_ppm_pool_backward(B * C, H, W, 4, arr, P(gpu(gp)), P(dxg), None)
    close(dxg, dxr, 1e-5, "pool backward")
    priors = [torch.randn(B, Cout, s, s, generator=g) for s in sizes]
    pgs = [gpu(t) for t in priors]
    cat_r, cat_g = torch.empty(B, 4 * Cout + C, H, W), torch.full((B, 4 * Cout + C, H, W), 7.0, device=DEV)
    assert ref.skd_ppm_concat(B, Cout, C, H, W, 4, arr, (ctypes.c_void_p * 4)(*[t.data_ptr() for t in priors]), P(x), P(cat_r), None)
    assert hip.skd_ppm_concat(B, Cout, C, H, W, 4, arr, (ctypes.c_void_p * 4)(*[t.data_ptr() for t in pgs]), P(gpu(x)), P(cat_g), None)
    close(cat_g, cat_r, 1e-5, "concat")
    assert torch.equal(cat_g[:, 4 * Cout:].cpu(), x), "feature slice is a bit-exact copy"
    gc = torch.randn(B, 4 * Cout + C, H, W, generator=g)
    gr, gg = [torch.empty_like(t) for t in priors], [torch.empty_like(t) for t in pgs]
    assert ref.skd_ppm_concat_backward(B, Cout, C, H, W, 4, arr, P(gc), (ctypes.c_void_p * 4)(*[t.data_ptr() for t in gr]), None)
    assert hip.skd_ppm_concat_backward(B, Cout, C, H, W, 4, arr, P(gpu(gc)), (ctypes.c_void_p * 4)(*[t.data_ptr() for t in gg]), None)
    for a, b, s_ in zip(gg, gr, sizes):
        close(a, b, 2e-5, "concat backward level %d" % s_)
    # against torch's own ops
    tp = torch.nn.functional.adaptive_avg_pool2d(x.double(), 3)
    off = B * C * 5
    close(pg[off:off + B * C * 9].view(B, C, 3, 3), tp.float(), 1e-5, "pool vs torch")
    # ---- channels-last entries: the same numbers as the NCHW oracle results, in (B, H, W, C) memory ----
    if C % 4:
        assert hip.skd_ppm_pool_nhwc(B, C, H, W, 4, arr, P(pg), P(pg), P(pg), None) == 0       # channel quads only
        return
    if Cout % 4:
        return
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()                       # (B, C, h, w) values -> (B, h, w, C) memory
    x_cl = gpu(nhwc(x))
    ws = torch.empty(max(1, hip.skd_ppm_nhwc_workspace_floats(B, C, Cout, H, W, 4, arr)), device=DEV)
    pcl = torch.full((total,), 7.0, device=DEV)
    assert hip.skd_ppm_pool_nhwc(B, C, H, W, 4, arr, P(x_cl), P(pcl), P(ws), None)
    off = 0
    for s_ in sizes:
        n = B * C * s_ * s_
        close(pcl[off:off + n].view(B, s_, s_, C).permute(0, 3, 1, 2), pr[off:off + n].view(B, C, s_, s_), 1e-5, "nhwc pooled level %d" % s_)
        off += n
    gp_cl, off = [], 0
    for s_ in sizes:
        n = B * C * s_ * s_
        gp_cl.append(nhwc(gp[off:off + n].view(B, C, s_, s_)).reshape(-1))
        off += n
    dx_cl = torch.full((B, H, W, C), 7.0, device=DEV)
    assert hip.skd_ppm_pool_backward_nhwc(B, C, H, W, 4, arr, P(gpu(torch.cat(gp_cl))), P(dx_cl), None)
    close(dx_cl.permute(0, 3, 1, 2), dxr, 1e-5, "nhwc pool backward")
    pr_cl = [gpu(nhwc(t)) for t in priors]
    cat_cl = torch.full((B, H, W, 4 * Cout + C), 7.0, device=DEV)
    assert hip.skd_ppm_concat_nhwc(B, Cout, C, H, W, 4, arr, (ctypes.c_void_p * 4)(*[t.data_ptr() for t in pr_cl]), P(x_cl), P(cat_cl), None)
    close(cat_cl.permute(0, 3, 1, 2), cat_r, 1e-5, "nhwc concat")
    assert torch.equal(cat_cl[..., 4 * Cout:].cpu(), nhwc(x)), "feature slice is a bit-exact copy"
    gg_cl = [torch.full((B, s_, s_, Cout), 7.0, device=DEV) for s_ in sizes]
    gf_cl = torch.full((B, H, W, C), 7.0, device=DEV)
    gc_cl = gpu(nhwc(gc))
    assert hip.skd_ppm_concat_backward_nhwc(B, Cout, C, H, W, 4, arr, P(gc_cl), (ctypes.c_void_p * 4)(*[t.data_ptr() for t in gg_cl]), P(gf_cl), P(ws), None)
    for a, b, s_ in zip(gg_cl, gr, sizes):
        close(a.permute(0, 3, 1, 2), b, 2e-5, "nhwc concat backward level %d" % s_)
    assert torch.equal(gf_cl.cpu(), nhwc(gc[:, 4 * Cout:])), "gradient of the feature map = its slice of gcat"
    # the C oracle's channels-last forms agree with its NCHW forms by construction; one spot check keeps them honest
    cat_o = torch.empty(B, H, W, 4 * Cout + C)
    keep = [nhwc(t) for t in priors] + [nhwc(x)]
    assert ref.skd_ppm_concat_nhwc(B, Cout, C, H, W, 4, arr, (ctypes.c_void_p * 4)(*[t.data_ptr() for t in keep[:4]]), P(keep[4]), P(cat_o), None)
    assert torch.equal(cat_o.permute(0, 3, 1, 2), cat_r)


@pytest.mark.parametrize("geom", [(1, 19, 129, 257, 1024, 2048), (2, 19, 65, 65, 512, 512), (2, 11, 46, 61, 360, 480), (1, 3, 1, 1, 4, 4),
                                  (3, 64, 5, 7, 33, 20), (1, 2, 9, 9, 9, 9)])
def test_seg_confusion_bit_exact(hip, ref, geom):
    """Evaluation tail: prediction (uint8 argmax) and int64 confusion counts are bit-exact with the C oracle."""
    B, C, h, w, H, W = geom
    g = torch.Generator().manual_seed(H + W + C)
    lg = torch.randn(B, C, h, w, generator=g) * 4
    lg[0, :, 0, 0] = 1.5                                   # an exact tie across all classes: first index must win
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[0, : max(1, H // 16)] = 255
    pr, cr = torch.empty(B, H, W, dtype=torch.uint8), torch.zeros(C, C, dtype=torch.int64)
    assert ref.skd_seg_confusion(B, C, h, w, H, W, P(lg), P(y), 255, P(pr), P(cr), None)
    pg, cg = torch.empty(B, H, W, dtype=torch.uint8, device=DEV), torch.zeros(C, C, dtype=torch.int64, device=DEV)
    assert hip.skd_seg_confusion(B, C, h, w, H, W, P(gpu(lg)), P(y.to(DEV)), 255, P(pg), P(cg), None)
    assert torch.equal(pg.cpu(), pr), "predictions must be bit-exact"
    assert torch.equal(cg.cpu(), cr), "confusion counts must be bit-exact"
    assert int(pg[0, 0, 0]) == 0 and int(cg.sum()) == int((y != 255).sum())
    # accumulation + prediction-only form
    assert hip.skd_seg_confusion(B, C, h, w, H, W, P(gpu(lg)), P(y.to(DEV)), 255, None, P(cg), None)
    assert torch.equal(cg.cpu(), 2 * cr)
    pg2 = torch.empty_like(pg)
    assert hip.skd_seg_confusion(B, C, h, w, H, W, P(gpu(lg)), None, 255, P(pg2), None, None)
    assert torch.equal(pg2, pg)
    # mIoU from the counts (evaluate.py:200-206)
    from structure_knowledge_distillation_amd.networks.evaluate import iou_from_confusion
    m1, _ = iou_from_confusion(cg.cpu().numpy())
    m2, _ = iou_from_confusion((2 * cr).numpy())
    assert m1 == m2


def test_sum_f32(hip):
    for n in (0, 1, 255, 4097, 1 << 20):
        x = torch.randn(max(n, 1), device=DEV)[:n]
        out = torch.empty(1, device=DEV)
        ws = torch.empty(2048, device=DEV)
        assert hip.skd_sum_f32(n, P(x) if n else None, P(out), 0.5, P(ws), None)
        want = 0.5 * float(x.double().sum()) if n else 0.0
        assert abs(float(out) - want) <= 1e-6 * max(1.0, float(x.double().abs().sum()) if n else 1.0)


# ---- size-independent properties at the full BASELINE sizes ------------------------------------------
def test_abn_full_size_properties(hip):
    """(8,64,256,256) stem tensor and (8,512,65,65) layer4 tensor: normalised output has per-channel
    mean beta and variance gamma^2 (to fp32 reduction accuracy); two runs are bit-identical."""
    for (N, C, S) in ((8, 64, 65536), (8, 512, 4225)):
        x = torch.randn(N, C, S, device=DEV) * 2 + 3
        w = torch.rand(C, device=DEV) + 0.5
        b = torch.randn(C, device=DEV)
        outs = []
        for _ in range(2):
            z = x.clone()
            m, v = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
            ws = torch.empty(hip.skd_abn_workspace_floats(N, C, S), device=DEV)
            assert hip.skd_abn_forward_train(N, C, S, P(z), P(w), P(b), None, None, P(m), P(v), 0.1, 1e-5, 0, 0.01, P(ws), None)
            outs.append((z, m, v))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
        z, m, v = outs[0]
        zd = z.double()
        close(zd.mean((0, 2)).float(), b, 1e-5, "mean(z) == beta")
        gamma2 = ((w + 1e-5) ** 2).double() * (v.double() / (v.double() + 1e-5))
        close(zd.var((0, 2), unbiased=False).float(), gamma2.float(), 1e-4, "var(z) == gamma^2")
        close(m, x.double().mean((0, 2)).float(), 1e-5, "mean")
        close(v, x.double().var((0, 2), unbiased=False).float(), 2e-5, "var")


def test_pairwise_full_size_properties(hip):
    """BASELINE config-2 features (8,128,65,65)/(8,512,65,65): Pa(F, F) == 0 exactly-ish, G symmetric,
    and the loss is invariant to a positive per-pixel rescale of the student features."""
    from structure_knowledge_distillation_amd import functional as SF
    B = 8
    fs = torch.randn(B, 128, 65, 65, device=DEV)
    ft = torch.randn(B, 512, 65, 65, device=DEV)
    for k in (32, 8, 2):
        l_same = SF.pair_wise_loss(ft, ft, k, k)
        assert abs(float(l_same)) < 1e-9
        l1 = SF.pair_wise_loss(fs, ft, k, k)
        l2 = SF.pair_wise_loss(fs * 3.0, ft * 0.25, k, k)
        assert abs(float(l1) - float(l2)) <= 1e-5 * abs(float(l1))
        want = _pa_torch(fs.double(), ft.double(), k)
        assert abs(float(l1) - float(want)) <= 1e-5 * abs(float(want))


def _pa_torch(fs, ft, k):
    F = torch.nn.functional
    ps, pt = F.max_pool2d(fs, (k, k), (k, k), 0, ceil_mode=True), F.max_pool2d(ft, (k, k), (k, k), 0, ceil_mode=True)

    def sim(f):
        f = f / ((f ** 2).sum(1, keepdim=True).sqrt() + 1e-8)
        f = f.reshape(f.shape[0], f.shape[1], -1)
        return torch.einsum("icm,icn->imn", f, f)
    return ((sim(pt) - sim(ps)) ** 2).sum() / (pt.shape[-1] * pt.shape[-2]) ** 2 / pt.shape[0]


# (33800 / 33791, 64, 512): more tiles than one round of the chip's slots -- the rows behind the last whole round run as HALF-HEIGHT
# tiles (round 6), with a ragged last half panel
@pytest.mark.parametrize("M,K,N", [(1000, 64, 128), (4225, 256, 1024), (777, 1024, 256), (129, 2048, 512), (128, 128, 128),
                                   (33800, 64, 512), (33791, 32, 384)])
@pytest.mark.parametrize("act,with_res", [(3, True), (3, False), (0, False), (1, True)])
def test_conv1x1_abn_gemm(hip, ref, M, K, N, act, with_res):
    """1x1 convolution + eval-mode ABN (+ residual) + activation as one fp32-MFMA GEMM vs the C oracle (double dot
    product, then the bn.cu forward formula)."""
    g = torch.Generator().manual_seed(M + K + N)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    r = torch.randn(M, N, generator=g) if with_res else None
    mean, var = torch.randn(N, generator=g) * 0.3, torch.rand(N, generator=g) + 0.5
    ga, be = torch.randn(N, generator=g), torch.randn(N, generator=g)
    o_r, o_g = torch.empty(M, N), torch.full((M, N), 7.0, device=DEV)
    assert ref.skd_conv1x1_abn_nhwc(M, K, N, P(x), P(w), P(r), P(o_r), P(mean), P(var), P(ga), P(be), 1e-5, act, 0.01, None)
    assert hip.skd_conv1x1_abn_nhwc(M, K, N, P(gpu(x)), P(gpu(w)), P(gpu(r)), P(o_g), P(gpu(mean)), P(gpu(var)), P(gpu(ga)), P(gpu(be)), 1e-5, act, 0.01, None)
    close(o_g, o_r, 2e-5, "conv1x1+abn")
    assert hip.skd_conv1x1_abn_supported(M, K, N) == 1 and hip.skd_conv1x1_abn_supported(M, K + 8, N) == 0 and hip.skd_conv1x1_abn_supported(M, K, N + 64) == 0
    assert hip.skd_conv1x1_abn_nhwc(M, K + 8, N, P(gpu(x)), P(gpu(w)), None, P(o_g), P(gpu(mean)), P(gpu(var)), None, None, 1e-5, act, 0.01, None) == 0


@pytest.mark.parametrize("M,K,N", [(1000, 64, 128), (4225, 256, 1024), (777, 96, 256), (129, 512, 2048), (33800, 48, 512)])
@pytest.mark.parametrize("with_res,affine", [(True, True), (False, True), (True, False)])
def test_conv1x1_abn_gemm_with_bn_relu_prologue(hip, ref, M, K, N, with_res, affine):
    """skd_conv1x1_abn_pro_nhwc: relu(bn_k(x)) applied to the GEMM's A operand on the way into LDS (the bottleneck tail
    conv2 -> bn2 -> relu -> conv3 -> bn3 -> + residual -> relu, pspnet_combine.py:71-82) vs the C oracle, which
    materialises the activated operand; x itself must stay untouched."""
    g = torch.Generator().manual_seed(M + K + N + 1)
    x, w = torch.randn(M, K, generator=g) * 2, torch.randn(N, K, generator=g) / K ** 0.5
    r = torch.randn(M, N, generator=g) if with_res else None
    mean, var = torch.randn(N, generator=g) * 0.3, torch.rand(N, generator=g) + 0.5
    ga, be = torch.randn(N, generator=g), torch.randn(N, generator=g)
    pm, pv = torch.randn(K, generator=g) * 0.5, torch.rand(K, generator=g) + 0.5
    pw, pb = (torch.randn(K, generator=g), torch.randn(K, generator=g) * 0.5) if affine else (None, None)
    o_r, o_g = torch.empty(M, N), torch.full((M, N), 7.0, device=DEV)
    pk_r, pk_g = torch.empty(4, K), torch.empty(4, K, device=DEV)
    assert ref.skd_abn_pack_eval_params(K, P(pm), P(pv), P(pw), P(pb), 1e-5, P(pk_r), None)
    assert hip.skd_abn_pack_eval_params(K, P(gpu(pm)), P(gpu(pv)), P(gpu(pw)), P(gpu(pb)), 1e-5, P(pk_g), None)
    assert torch.equal(pk_g.cpu()[0], pm) and torch.equal(pk_g.cpu(), pk_r)          # correctly rounded sqrt / divide on both sides
    assert ref.skd_conv1x1_abn_pro_nhwc(M, K, N, P(x), P(w), P(r), P(o_r), P(mean), P(var), P(ga), P(be), 1e-5, P(pk_r), 3, 0.01, None)
    xg = gpu(x)
    assert hip.skd_conv1x1_abn_pro_nhwc(M, K, N, P(xg), P(gpu(w)), P(gpu(r)), P(o_g), P(gpu(mean)), P(gpu(var)), P(gpu(ga)), P(gpu(be)), 1e-5,
                                        P(pk_g), 3, 0.01, None)
    close(o_g, o_r, 3e-5, "bn+relu -> conv1x1 -> abn")
    assert torch.equal(xg.cpu(), x)
    assert hip.skd_conv1x1_abn_pro_nhwc(M, K, N, P(xg), P(gpu(w)), None, P(o_g), P(gpu(mean)), P(gpu(var)), None, None, 1e-5, None, 3, 0.01, None) == 0


def test_teacher_bottleneck_fused_tail_equals_unfused(monkeypatch):
    """pspnet_combine.FUSED_TAIL (conv2 -> ONE GEMM with bn2 + ReLU in its prologue and bn3 + residual + ReLU in its epilogue) gives the
    frozen bottleneck's output of the default path (conv + in-place ABN passes) at the layer-3 shape."""
    from structure_knowledge_distillation_amd.networks import pspnet_combine as PC
    torch.manual_seed(3)
    blk = PC.Bottleneck(1024, 256, stride=1, dilation=2).to(DEV).eval().to(memory_format=torch.channels_last)
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2, blk.bn3):
            bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5); bn.weight.normal_(0, 1); bn.bias.normal_(0, 0.5)
        x = torch.randn(2, 1024, 65, 65, device=DEV).contiguous(memory_format=torch.channels_last)
        monkeypatch.setattr(PC_MOD, "FUSED_TAIL", False)
        want = blk(x.clone(memory_format=torch.channels_last))
        monkeypatch.setattr(PC_MOD, "FUSED_TAIL", True)
        got = blk(x.clone(memory_format=torch.channels_last))
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    close(got, want, 3e-5, "fused bottleneck tail")


def test_conv1x1_abn_gemm_full_size_vs_conv2d():
    """The teacher's layer3 block tail at batch 8 (M = 33800, K = 256, N = 1024) against conv2d + the fused ABN pass."""
    from structure_knowledge_distillation_amd import functional as SF
    import importlib
    IA = importlib.import_module("structure_knowledge_distillation_amd.libs.inplace_abn")
    torch.manual_seed(0)
    x = torch.randn(8, 256, 65, 65, device=DEV).contiguous(memory_format=torch.channels_last)
    res = torch.randn(8, 1024, 65, 65, device=DEV).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(256, 1024, 1, bias=False).to(DEV).to(memory_format=torch.channels_last)
    rm, rv = torch.randn(1024, device=DEV) * 0.1, torch.rand(1024, device=DEV) + 0.5
    w, b = torch.randn(1024, device=DEV), torch.randn(1024, device=DEV)
    with torch.no_grad():
        assert SF.conv1x1_abn_supported(x, conv)
        got = SF.conv1x1_abn_eval(x, conv.weight, rm, rv, w, b, 1e-5, "relu", 0.01, res)
        want = IA.abn_eval_fused(conv(x), w, b, rm, rv, 1e-5, "relu", 0.01, res)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    close(got, want, 2e-5, "fused GEMM vs conv2d + ABN pass")


@pytest.mark.parametrize("geom", [(2, 512, 65, 65, (1, 2, 3, 6)), (8, 128, 65, 65, (1, 2, 3, 6)), (2, 128, 33, 33, (1, 2, 3, 6)),
                                  (1, 8, 129, 257, (1, 2, 3, 6)), (3, 4, 7, 9, (1, 2, 3, 6)), (2, 12, 6, 6, (1, 2, 3, 6)),
                                  (1, 4, 1, 5, (1,)), (2, 20, 46, 61, (2, 6)), (1, 16, 3, 2, (1, 2))])
def test_ppm_fold(hip, ref, geom):
    """skd_ppm_fold_nhwc / skd_ppm_fold_backward_nhwc (the pyramid priors folded through the 3x3 bottleneck
    convolution, pspnet_combine.py:104-111) against the plain-C definition; the backward is the exact transpose."""
    B, Cout, H, W, sizes = geom
    L = len(sizes)
    arr = (ctypes.c_int * L)(*sizes)
    LD = 9 * Cout                                                     # dense rows; the strided form is checked below
    g = torch.Generator().manual_seed(Cout + H + W)
    zs = [torch.randn(B * s * s, 9 * Cout, generator=g) for s in sizes]
    base = torch.randn(B, H, W, Cout, generator=g)
    out_r = base.clone()
    assert ref.skd_ppm_fold_nhwc(B, Cout, H, W, L, arr, (ctypes.c_void_p * L)(*[z.data_ptr() for z in zs]), LD, P(out_r), None)
    zg = [gpu(z) for z in zs]
    out_g = gpu(base)
    assert hip.skd_ppm_fold_nhwc(B, Cout, H, W, L, arr, (ctypes.c_void_p * L)(*[z.data_ptr() for z in zg]), LD, P(out_g), None)
    close(out_g, out_r, 2e-5, "fold forward")
    # a second call accumulates on top (in-place += contract)
    assert hip.skd_ppm_fold_nhwc(B, Cout, H, W, L, arr, (ctypes.c_void_p * L)(*[z.data_ptr() for z in zg]), LD, P(out_g), None)
    close(out_g, 2 * out_r - base, 2e-5, "fold forward accumulates")
    gout = torch.randn(B, H, W, Cout, generator=g)
    gr = [torch.full_like(z, float("nan")) for z in zs]
    assert ref.skd_ppm_fold_backward_nhwc(B, Cout, H, W, L, arr, P(gout), (ctypes.c_void_p * L)(*[t.data_ptr() for t in gr]), LD, P(torch.empty(8)), None)
    nws = hip.skd_ppm_fold_nhwc_workspace_floats(B, Cout, H, W, L, arr)
    assert nws > 0
    ws = torch.full((nws,), float("nan"), device=DEV)
    gg = [torch.full_like(z, float("nan")) for z in zg]
    assert hip.skd_ppm_fold_backward_nhwc(B, Cout, H, W, L, arr, P(gpu(gout)), (ctypes.c_void_p * L)(*[t.data_ptr() for t in gg]), LD, P(ws), None)
    for k in range(L):
        close(gg[k], gr[k], 2e-5, "fold backward level %d" % sizes[k], floor=float(gout.abs().max()))
    # transpose identity at full precision: <fold(Z), G> == <Z, fold^T(G)>
    zero = torch.zeros(B, H, W, Cout, device=DEV)
    assert hip.skd_ppm_fold_nhwc(B, Cout, H, W, L, arr, (ctypes.c_void_p * L)(*[z.data_ptr() for z in zg]), LD, P(zero), None)
    lhs = float((zero.double() * gpu(gout).double()).sum())
    rhs = sum(float((zg[k].double() * gg[k].double()).sum()) for k in range(L))
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), float(zero.double().norm() * gout.double().norm()) * 1e-2), (lhs, rhs)
    # rejected: channel count not a multiple of 4, missing pointers
    assert hip.skd_ppm_fold_nhwc(B, Cout + 1, H, W, L, arr, (ctypes.c_void_p * L)(*[z.data_ptr() for z in zg]), LD, P(out_g), None) == 0
    assert hip.skd_ppm_fold_nhwc(B, Cout, H, W, L, arr, None, LD, P(out_g), None) == 0
    assert hip.skd_ppm_fold_nhwc(B, Cout, H, W, L, arr, (ctypes.c_void_p * L)(*[z.data_ptr() for z in zg]), LD - 4, P(out_g), None) == 0
    # strided rows: the levels as diagonal blocks of one (sum B s^2, L * 9 * Cout) matrix whose other blocks hold NaN
    rows = [B * s * s for s in sizes]
    z_all = torch.full((sum(rows), L * LD), float("nan"), device=DEV)
    g_all = torch.zeros_like(z_all)
    zp, gp, r0 = [], [], 0
    for k in range(L):
        z_all[r0:r0 + rows[k], k * LD:(k + 1) * LD] = zg[k]
        zp.append(z_all.data_ptr() + 4 * (r0 * L * LD + k * LD))
        gp.append(g_all.data_ptr() + 4 * (r0 * L * LD + k * LD))
        r0 += rows[k]
    out_s = gpu(base)
    assert hip.skd_ppm_fold_nhwc(B, Cout, H, W, L, arr, (ctypes.c_void_p * L)(*zp), L * LD, P(out_s), None)
    close(out_s, out_r, 2e-5, "fold forward, strided Z")
    assert hip.skd_ppm_fold_backward_nhwc(B, Cout, H, W, L, arr, P(gpu(gout)), (ctypes.c_void_p * L)(*gp), L * LD, P(ws), None)
    r0 = 0
    for k in range(L):
        assert torch.equal(g_all[r0:r0 + rows[k], k * LD:(k + 1) * LD], gg[k])
        g_all[r0:r0 + rows[k], k * LD:(k + 1) * LD] = 0
        r0 += rows[k]
    assert float(g_all.abs().max()) == 0.0                            # nothing written outside the diagonal blocks


@pytest.mark.parametrize("cfg", [(2, 512, 128, 65, 65, True), (2, 2048, 512, 33, 33, False), (1, 64, 16, 129, 257, False)])
def test_psp_module_fold_vs_concat(cfg, monkeypatch):
    """PSPModule on the GPU: folded bottleneck vs the concatenate-then-convolve sequence of pspnet_combine.py:104-111
    (same module, same weights) and vs plain torch ops in float64 on the CPU."""
    import torch.nn.functional as F
    from structure_knowledge_distillation_amd.networks.pspnet_combine import PSPModule
    B, Cf, Cm, H, W, train = cfg
    torch.manual_seed(3)
    m = PSPModule(Cf, Cm)
    for p in m.parameters():
        torch.nn.init.normal_(p, 0.0, 0.05)
    m = m.to(DEV).to(memory_format=torch.channels_last)
    x = (torch.randn(B, Cf, H, W) * 2).to(DEV).contiguous(memory_format=torch.channels_last)

    def run(fold):
        monkeypatch.setattr(PC_MOD, "PSP_FOLD", bool(fold))
        for p in m.parameters():
            p.grad = None
        if not train:
            with torch.no_grad():
                return m.eval()(x.clone()), None, None
        m.train()
        for mod in m.modules():
            if getattr(mod, "running_mean", None) is not None:
                mod.running_mean.zero_()
                mod.running_var.fill_(1.0)
        xx = x.clone().requires_grad_(True)
        torch.manual_seed(11)
        out = m(xx)
        torch.manual_seed(12)
        (out * torch.randn(out.shape, device=DEV)).sum().backward()
        return out.detach(), xx.grad, {k: v.grad.clone() for k, v in m.named_parameters()}

    o1, g1, p1 = run(True)
    o0, g0, p0 = run(False)
    close(o1, o0, 2e-5, "PSP output fold vs concat")
    if train:
        close(g1, g0, 5e-5, "PSP input gradient")
        for k in p0:
            close(p1[k], p0[k], 2e-4, "PSP grad " + k, floor=float(p0[k].abs().max()))
    # bottleneck convolution alone vs the reference graph in float64
    from structure_knowledge_distillation_amd import functional as SF
    with torch.no_grad():
        m.eval()
        sizes = (1, 2, 3, 6)
        priors = [st[2](st[1](p)) for st, p in zip(m.stages, SF.ppm_pool(x, sizes))]
        got = SF.ppm_fold_bottleneck(priors, x, m.bottleneck[0].weight, {})
        pri = [F.interpolate(p.cpu().double().contiguous(), size=(H, W), mode="bilinear", align_corners=True) for p in priors]
        want = F.conv2d(torch.cat(pri + [x.cpu().double().contiguous()], 1), m.bottleneck[0].weight.cpu().double(), None, 1, 1)
    close(got, want, 2e-5, "folded bottleneck vs float64 cat+conv")


@pytest.mark.parametrize("geom", [(8, 128, 256, 256), (2, 128, 128, 128), (1, 4, 9, 12), (2, 8, 65, 33), (1, 64, 512, 1024), (3, 12, 1, 2)])
def test_maxpool3x3s2_nhwc(hip, ref, geom):
    """skd_maxpool3x3s2_nhwc / _backward_nhwc (the stem's MaxPool2d(3, 2, 1, ceil_mode=True), pspnet_combine.py:135):
    values and argmax codes bit-exact vs the plain-C oracle and vs torch (ties, -inf, NaN); backward = torch's."""
    from structure_knowledge_distillation_amd.functional import _pool_out
    B, C, H, W = geom
    OH, OW = _pool_out(H, True), _pool_out(W, True)
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, H, W, C, generator=g)
    x[torch.rand(B, H, W, C, generator=g) < 0.3] = 1.0                # ties
    x.view(-1)[::997] = float("-inf")
    x.view(-1)[5::1999] = float("nan")
    small = B * H * W * C <= (1 << 24)
    yg = torch.empty(B, OH, OW, C, device=DEV)
    ag = torch.full((B, OH, OW, C), 255, dtype=torch.uint8, device=DEV)
    assert hip.skd_maxpool3x3s2_nhwc(B, C, H, W, OH, OW, P(gpu(x)), P(yg), P(ag), None)
    want, idx = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1, ceil_mode=True, return_indices=True)
    nn = lambda t: torch.nan_to_num(t, nan=7e7)                        # noqa: E731
    assert torch.equal(nn(yg.cpu()), nn(want.permute(0, 2, 3, 1)))
    # argmax code -> flat h * W + w index = torch's
    a = ag.cpu().long()
    oy = torch.arange(OH).view(1, OH, 1, 1)
    ox = torch.arange(OW).view(1, 1, OW, 1)
    flat = (2 * oy - 1 + a // 3) * W + (2 * ox - 1 + a % 3)
    assert torch.equal(flat, idx.permute(0, 2, 3, 1))
    if small:
        yr, ar = torch.empty(B, OH, OW, C), torch.empty(B, OH, OW, C, dtype=torch.uint8)
        assert ref.skd_maxpool3x3s2_nhwc(B, C, H, W, OH, OW, P(x), P(yr), P(ar), None)
        assert torch.equal(nn(yg.cpu()), nn(yr)) and torch.equal(ag.cpu(), ar)
    # inference form: no argmax written
    y2 = torch.empty_like(yg)
    assert hip.skd_maxpool3x3s2_nhwc(B, C, H, W, OH, OW, P(gpu(x)), P(y2), None, None)
    assert torch.equal(nn(y2), nn(yg))
    gy = torch.randn(B, OH, OW, C, generator=g)
    dxg = torch.full((B, H, W, C), float("nan"), device=DEV)
    assert hip.skd_maxpool3x3s2_backward_nhwc(B, C, H, W, OH, OW, P(gpu(gy)), P(ag), P(dxg), None)
    xt = torch.nan_to_num(x, nan=1e9).permute(0, 3, 1, 2).clone().requires_grad_(True)   # NaN wins in torch too; keep autograd finite
    torch.nn.functional.max_pool2d(xt, 3, 2, 1, ceil_mode=True).backward(gy.permute(0, 3, 1, 2))
    close(dxg, xt.grad.permute(0, 2, 3, 1), 1e-6, "max-pool backward vs torch", floor=float(gy.abs().max()))
    if small:
        dxr = torch.empty(B, H, W, C)
        assert ref.skd_maxpool3x3s2_backward_nhwc(B, C, H, W, OH, OW, P(gy), P(ar), P(dxr), None)
        close(dxg, dxr, 1e-6, "max-pool backward vs oracle", floor=float(gy.abs().max()))
    assert hip.skd_maxpool3x3s2_nhwc(B, C + 1, H, W, OH, OW, P(gpu(x)), P(yg), None, None) == 0
    assert hip.skd_maxpool3x3s2_nhwc(B, C, H, W, OH + 1, OW, P(gpu(x)), P(yg), None, None) == 0


@pytest.mark.parametrize("geom", [(8, 65 * 65, 128, 19), (2, 33 * 33, 128, 19), (1, 7, 128, 3), (3, 100, 128, 20), (8, 65 * 65, 512, 19), (2, 50, 1024, 11)])
def test_head1x1(hip, ref, geom):
    """Round 6: the 19-class 1x1 classifier heads on channels-last feature maps (skd_head1x1_forward_nhwc / _backward_nhwc; networks/
    pspnet_combine.py:138-154) against the plain-C oracle (double accumulation): NCHW logits out, channels-last feature gradient,
    weight / bias gradients WRITTEN in a fixed order (two runs give the same bits); ragged row counts, C not a multiple of 4."""
    B, HW, K, C = geom
    M = B * HW
    g = torch.Generator().manual_seed(M + K + C)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(C, K, generator=g) / K ** 0.5, torch.randn(C, generator=g)
    xg, wg, bg = gpu(x), gpu(w), gpu(b)
    o_r, o_g = torch.empty(B, C, HW), torch.full((B, C, HW), float("nan"), device=DEV)
    assert hip.skd_head1x1_supported(K, C, 0) == 1 and ref.skd_head1x1_supported(K, C, 0) == 1
    assert ref.skd_head1x1_forward_nhwc(B, HW, K, C, P(x), P(w), P(b), P(o_r), None)
    assert hip.skd_head1x1_forward_nhwc(B, HW, K, C, P(xg), P(wg), P(bg), P(o_g), None)
    close(o_g, o_r, 2e-5, "head forward")
    o_nb = torch.empty_like(o_g)
    assert hip.skd_head1x1_forward_nhwc(B, HW, K, C, P(xg), P(wg), None, P(o_nb), None)
    close(o_nb, o_r - b.view(1, C, 1), 2e-5, "head forward without bias", floor=float(o_r.abs().max()))
    assert hip.skd_head1x1_forward_nhwc(B, HW, K + 64, C, P(xg), P(wg), P(bg), P(o_g), None) == 0
    assert hip.skd_head1x1_forward_nhwc(B, HW, K, 21, P(xg), P(wg), P(bg), P(o_g), None) == 0
    if K != 128:
        assert hip.skd_head1x1_supported(K, C, 1) == 0
        return
    go = torch.randn(B, C, HW, generator=g)
    gog = gpu(go)
    ws = torch.empty(max(1, hip.skd_head1x1_backward_workspace_floats(B, HW, K, C)), device=DEV)
    gx_g, gw_g, gb_g = torch.full((M, K), float("nan"), device=DEV), torch.full((C, K), float("nan"), device=DEV), torch.full((C,), float("nan"), device=DEV)
    gx_r, gw_r, gb_r, ws_r = torch.empty(M, K), torch.empty(C, K), torch.empty(C), torch.empty(4)
    assert ref.skd_head1x1_backward_nhwc(B, HW, K, C, P(x), P(w), P(go), P(gx_r), P(gw_r), P(gb_r), P(ws_r), None)
    assert hip.skd_head1x1_backward_nhwc(B, HW, K, C, P(xg), P(wg), P(gog), P(gx_g), P(gw_g), P(gb_g), P(ws), None)
    close(gx_g, gx_r, 2e-5, "head dx")
    close(gw_g, gw_r, 3e-5, "head dW", floor=float(M ** 0.5))
    close(gb_g, gb_r, 3e-5, "head db", floor=float(M ** 0.5))
    gw2, gb2 = torch.empty_like(gw_g), torch.empty_like(gb_g)
    assert hip.skd_head1x1_backward_nhwc(B, HW, K, C, P(xg), P(wg), P(gog), None, P(gw2), P(gb2), P(ws), None)
    assert torch.equal(gw2, gw_g) and torch.equal(gb2, gb_g), "head dW / db are not bit-reproducible"
    gx2 = torch.empty_like(gx_g)
    assert hip.skd_head1x1_backward_nhwc(B, HW, K, C, None, P(wg), P(gog), P(gx2), None, None, P(ws), None)      # a frozen head: dx only
    assert torch.equal(gx2, gx_g)
    assert hip.skd_head1x1_backward_nhwc(B, HW, K, C, None, P(wg), P(gog), None, P(gw2), None, P(ws), None) == 0   # dW needs x


def test_classifier_head_kernel_equals_conv2d_full_size(monkeypatch):
    """pspnet_combine.ClassifierConv with HEAD_KERNEL (csrc/head.hip) against the same module through MIOpen, at the student's size:
    logits NCHW-contiguous, feature gradient channels-last, everything to rounding; the teacher's 512-channel head forward-only."""
    torch.manual_seed(2)
    head = PC_MOD.ClassifierConv(128, 19, 1, 1, 0, bias=True).to(DEV)
    x = torch.randn(8, 128, 65, 65, device=DEV).contiguous(memory_format=torch.channels_last)
    g = torch.randn(8, 19, 65, 65, device=DEV)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(PC_MOD, "HEAD_KERNEL", flag)
        head.zero_grad()
        xa = x.clone(memory_format=torch.channels_last).requires_grad_(True)
        y = head(xa)
        y.backward(g)
        res[flag] = (y.detach(), xa.grad, head.weight.grad.clone(), head.bias.grad.clone())
    assert res[True][0].is_contiguous() and res[True][1].is_contiguous(memory_format=torch.channels_last)
    for a, b, name in zip(res[True], res[False], ("logits", "dx", "dW", "db")):
        close(a, b, 3e-5, "classifier head " + name, floor=float(b.abs().max()))
    monkeypatch.setattr(PC_MOD, "HEAD_KERNEL", True)
    t = PC_MOD.ClassifierConv(512, 19, 1, 1, 0, bias=True).to(DEV)
    xt = torch.randn(8, 512, 65, 65, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        close(t(xt), torch.nn.functional.conv2d(xt, t.weight, t.bias), 3e-5, "teacher head")
        assert t(xt).is_contiguous()


@pytest.mark.parametrize("geom", [(8, 128, 256, 256), (2, 128, 128, 96), (1, 4, 9, 12), (2, 8, 65, 33), (3, 16, 1, 2), (2, 64, 17, 64)])
@pytest.mark.parametrize("affine", [True, False])
def test_abn_relu_maxpool_stem(hip, ref, geom, affine):
    """Round 6, the training stem fused (skd_abn_relu_maxpool3x3s2_nhwc / _backward_reduce_nhwc / _backward_dx_nhwc; networks/
    pspnet_combine.py:176-180: bn3 -> relu3 -> maxpool).  Forward: pooled values AND argmax bytes bit for bit what the library's own
    normalise + ReLU pass followed by its max-pool produce (the sequence the fusion replaces), and the C oracle's restatement of
    that sequence to the usual bound.  Backward: edz / eydz / dx / dweight / dbias against the un-fused GPU sequence (un-pool, then the
    BatchNorm + ReLU backward passes: same gathered terms in the same order, reductions in another order) and against the oracle."""
    from structure_knowledge_distillation_amd.functional import _pool_out
    B, C, H, W = geom
    OH, OW = _pool_out(H, True), _pool_out(W, True)
    rows = B * H * W
    g = torch.Generator().manual_seed(H * 3 + W + C)
    x = torch.randn(B, H, W, C, generator=g) * 2.0 + torch.randn(1, 1, 1, C, generator=g)
    x[torch.rand(B, H, W, C, generator=g) < 0.05] = 0.75                # exact ties between window positions
    w = torch.randn(C, generator=g) if affine else None
    b = torch.randn(C, generator=g) * 0.5 if affine else None
    if affine:
        w[0], w[1] = 0.0, -abs(w[1])                                     # gamma = |w| + eps; zero weight: zero dweight
    xg, wg, bg = gpu(x), gpu(w), gpu(b)
    ws = torch.empty(max(1, hip.skd_abn_nhwc_workspace_floats(rows, C)), device=DEV)
    stat = torch.empty(2, C, device=DEV)
    assert hip.skd_abn_stats_nhwc(rows, C, P(xg), P(stat[0]), P(stat[1]), P(ws), None)
    # ---- forward
    pooled = torch.full((B, OH, OW, C), float("nan"), device=DEV)
    arg = torch.full((B, OH, OW, C), 255, dtype=torch.uint8, device=DEV)
    assert hip.skd_abn_relu_maxpool3x3s2_nhwc(B, C, H, W, OH, OW, P(xg), P(stat[0]), P(stat[1]), P(wg), P(bg), 1e-5, P(pooled), P(arg), None)
    y = torch.empty(B, H, W, C, device=DEV)
    assert hip.skd_abn_apply_nhwc_to(rows, C, P(xg), None, P(y), P(stat[0]), P(stat[1]), P(wg), P(bg), 1e-5, 3, 0.0, None)
    pooled_u, arg_u = torch.empty_like(pooled), torch.empty_like(arg)
    assert hip.skd_maxpool3x3s2_nhwc(B, C, H, W, OH, OW, P(y), P(pooled_u), P(arg_u), None)
    assert torch.equal(pooled, pooled_u) and torch.equal(arg, arg_u), "fused stem forward differs from normalise-then-pool"
    want, idx = torch.nn.functional.max_pool2d(y.permute(0, 3, 1, 2), 3, 2, 1, ceil_mode=True, return_indices=True)
    a = arg.long()
    oy, ox = torch.arange(OH, device=DEV).view(1, OH, 1, 1), torch.arange(OW, device=DEV).view(1, 1, OW, 1)
    assert torch.equal((2 * oy - 1 + a // 3) * W + (2 * ox - 1 + a % 3), idx.permute(0, 2, 3, 1)), "argmax differs from torch's indices"
    small = rows * C <= (1 << 23)
    st_c = stat.cpu()
    if small:
        pr, ar = torch.empty(B, OH, OW, C), torch.empty(B, OH, OW, C, dtype=torch.uint8)
        assert ref.skd_abn_relu_maxpool3x3s2_nhwc(B, C, H, W, OH, OW, P(x), P(st_c[0]), P(st_c[1]), P(w), P(b), 1e-5, P(pr), P(ar), None)
        close(pooled, pr, 2e-5, "fused stem forward vs oracle")
        differ = arg.cpu() != ar                  # only where two window positions are within rounding of each other
        assert float(differ.float().mean()) <= 1e-3 and float((pooled.cpu() - pr)[differ].abs().max() if differ.any() else 0.0) <= 1e-4
    # ---- backward
    gp = torch.randn(B, OH, OW, C, generator=g)
    gpg = gpu(gp)
    e, e_u = torch.empty(2, C, device=DEV), torch.empty(2, C, device=DEV)
    assert hip.skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc(B, C, H, W, OH, OW, P(xg), P(gpg), P(arg), P(stat[0]), P(stat[1]), P(wg), P(bg),
                                                             P(e[0]), P(e[1]), 1e-5, P(ws), None)
    dy = torch.empty(B, H, W, C, device=DEV)
    assert hip.skd_maxpool3x3s2_backward_nhwc(B, C, H, W, OH, OW, P(gpg), P(arg), P(dy), None)
    assert hip.skd_abn_relu_backward_reduce_nhwc_x(rows, C, P(xg), P(dy), P(stat[0]), P(stat[1]), P(wg), P(bg), P(e_u[0]), P(e_u[1]), 1e-5, P(ws), None)
    scale = float(dy.abs().mean()) + 1e-12
    close(e, e_u, 1e-5, "fused stem edz / eydz vs the un-fused GPU sequence", floor=scale)
    dx = torch.full((B, H, W, C), float("nan"), device=DEV)
    dw, db = (torch.full((C,), 3.0, device=DEV), torch.full((C,), -2.0, device=DEV)) if affine else (None, None)
    dx_u = torch.empty_like(dx)
    dw_u, db_u = (torch.empty(C, device=DEV), torch.empty(C, device=DEV)) if affine else (None, None)
    assert hip.skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(B, C, H, W, OH, OW, P(xg), P(gpg), P(arg), P(stat[0]), P(stat[1]), P(wg), P(bg),
                                                         P(e_u[0]), P(e_u[1]), P(dx), P(dw), P(db), 1e-5, 1, None)
    assert hip.skd_abn_relu_backward_dx_nhwc_x(rows, C, P(xg), P(dy), P(stat[0]), P(stat[1]), P(wg), P(bg), P(e_u[0]), P(e_u[1]), P(dx_u), P(dw_u),
                                              P(db_u), 1e-5, 0, None)
    assert torch.equal(dx, dx_u), "fused stem dx differs from the un-fused sequence on the same edz / eydz"
    if affine:
        assert torch.equal(dw - 3.0, (dw_u + 3.0) - 3.0) or float((dw - 3.0 - dw_u).abs().max()) <= 1e-5 * float(dw_u.abs().max() + 1.0)
        close(db + 2.0, db_u, 1e-5, "dbias (accumulate = 1)", floor=float(db_u.abs().max()) + 2.0)
        assert float(dw[0]) == 3.0                                        # zero weight: nothing added (bn.cu:217-223)
    if small:
        er, dxr = torch.empty(2, C), torch.empty(B, H, W, C)
        wsr = torch.empty(max(1, ref.skd_abn_nhwc_workspace_floats(rows, C)))
        assert ref.skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc(B, C, H, W, OH, OW, P(x), P(gp), P(arg.cpu()), P(st_c[0]), P(st_c[1]), P(w), P(b),
                                                                 P(er[0]), P(er[1]), 1e-5, P(wsr), None)
        close(e, er, 2e-5, "fused stem edz / eydz vs oracle", floor=scale)
        dwr, dbr = (torch.zeros(C), torch.zeros(C)) if affine else (None, None)
        assert ref.skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(B, C, H, W, OH, OW, P(x), P(gp), P(arg.cpu()), P(st_c[0]), P(st_c[1]), P(w), P(b),
                                                             P(er[0]), P(er[1]), P(dxr), P(dwr), P(dbr), 1e-5, 0, None)
        dx2 = torch.empty_like(dx)
        assert hip.skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(B, C, H, W, OH, OW, P(xg), P(gpg), P(arg), P(stat[0]), P(stat[1]), P(wg), P(bg),
                                                             P(e[0]), P(e[1]), P(dx2), None, None, 1e-5, 0, None)
        close(dx2, dxr, 5e-5, "fused stem dx vs oracle", floor=float(gp.abs().max()) * float(((w.abs() if affine else torch.ones(C)) / (st_c[1] + 1e-5).sqrt()).max()))
    # ---- argument checks: odd channel count, impossible pooled size, missing pointers
    assert hip.skd_abn_relu_maxpool3x3s2_nhwc(B, C + 1, H, W, OH, OW, P(xg), P(stat[0]), P(stat[1]), None, None, 1e-5, P(pooled), P(arg), None) == 0
    assert hip.skd_abn_relu_maxpool3x3s2_nhwc(B, C, H, W, OH + 1, OW, P(xg), P(stat[0]), P(stat[1]), None, None, 1e-5, P(pooled), P(arg), None) == 0
    assert hip.skd_abn_relu_maxpool3x3s2_nhwc(B, C, H, W, OH, OW, P(xg), P(stat[0]), P(stat[1]), None, None, 1e-5, P(pooled), None, None) == 0
    assert hip.skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(B, C, H, W, OH, OW, P(xg), P(gpg), P(arg), P(stat[0]), P(stat[1]), None, None,
                                                         P(e[0]), P(e[1]), P(dx), P(e[0]), None, 1e-5, 0, None) == 0      # dweight without weight


def test_student_stem_fused_equals_unfused_modules(monkeypatch):
    """libs.modules.forward_relu_maxpool (what pspnet_combine.ResNet.forward calls for the training student when STEM_FUSED) against
    forward_relu followed by the stem pool, full size: identical output bits, identical running statistics, gradients to rounding."""
    from structure_knowledge_distillation_amd import libs
    from structure_knowledge_distillation_amd import functional as SF
    torch.manual_seed(11)
    pool = torch.nn.MaxPool2d(3, 2, 1, ceil_mode=True)
    bn_a = libs.InPlaceABNSync(128, activation="none").to(DEV).train()
    bn_b = libs.InPlaceABNSync(128, activation="none").to(DEV).train()
    with torch.no_grad():
        bn_a.weight.normal_(); bn_a.bias.normal_(0, 0.5)
        bn_b.load_state_dict(bn_a.state_dict())
    x = (torch.randn(8, 128, 256, 256, device=DEV) * 1.5 + 0.3).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone(memory_format=torch.channels_last).requires_grad_(True), x.clone(memory_format=torch.channels_last).requires_grad_(True)
    ya = bn_a.forward_relu_maxpool(xa * 1.0, pool)
    yb = SF.max_pool_stem(bn_b.forward_relu(xb * 1.0), pool)
    assert ya.shape == (8, 128, 129, 129) and ya.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(ya, yb) and torch.equal(bn_a.running_mean, bn_b.running_mean) and torch.equal(bn_a.running_var, bn_b.running_var)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    close(xa.grad, xb.grad, 2e-5, "stem dx, fused vs un-fused")
    close(bn_a.weight.grad, bn_b.weight.grad, 2e-5, "stem dweight", floor=float(bn_b.weight.grad.abs().max()))
    close(bn_a.bias.grad, bn_b.bias.grad, 2e-5, "stem dbias", floor=float(bn_b.bias.grad.abs().max()))
    # an input the fused kernels do not take (NCHW) goes through forward_relu + the stock pool
    xn = torch.randn(2, 128, 17, 19, device=DEV)
    out = bn_a.forward_relu_maxpool(xn, pool)
    assert out.shape == (2, 128, 9, 10)


@pytest.mark.parametrize("cfg", [(8, 1024, 256, 65, True), (2, 128, 128, 129, True), (2, 512, 128, 33, False)])
def test_frozen_bottleneck_blas_tail(cfg, monkeypatch):
    """Frozen-teacher Bottleneck on the GPU: the 1x1 reduce convolution + BN + ReLU (and the stride-1 down-sample branch)
    as library GEMMs with the folded BN epilogue vs MIOpen convolution + the in-place ABN pass (pspnet_combine.py:65-84),
    and vs plain torch ops in float64."""
    from structure_knowledge_distillation_amd.networks.pspnet_combine import Bottleneck, BatchNorm2d
    B, Cin, planes, HW, with_down = cfg
    torch.manual_seed(4)
    down = torch.nn.Sequential(torch.nn.Conv2d(Cin, planes * 4, 1, 1, bias=False), BatchNorm2d(planes * 4)) if with_down or Cin != planes * 4 else None
    blk = Bottleneck(Cin, planes, stride=1, dilation=2, downsample=down).eval()
    for mod in blk.modules():
        if getattr(mod, "running_mean", None) is not None:
            mod.running_mean.normal_(0, 0.5)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.normal_(0, 1.0)
            mod.bias.data.normal_(0, 0.5)
    blk = blk.to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(B, Cin, HW, HW, device=DEV).contiguous(memory_format=torch.channels_last)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setattr(PC_MOD, "BLAS_TAILS", flag == "1")
        with torch.no_grad():
            outs[flag] = blk(x.clone())
    close(outs["1"], outs["0"], 2e-5, "blas tail vs conv + abn")
    # the reduce layer alone against float64
    from structure_knowledge_distillation_amd import functional as SF
    with torch.no_grad():
        got = SF.conv1x1_bn_blas(x, blk.conv1, blk.bn1, relu=True)
        bn = blk.bn1
        s = ((bn.weight.abs() + bn.eps) / torch.sqrt(bn.running_var + bn.eps)).double().cpu()
        y = torch.nn.functional.conv2d(x.double().cpu().contiguous(), blk.conv1.weight.double().cpu())
        want = torch.relu((y - bn.running_mean.double().cpu().view(1, -1, 1, 1)) * s.view(1, -1, 1, 1) + bn.bias.double().cpu().view(1, -1, 1, 1))
    close(got, want, 2e-5, "conv1x1_bn_blas vs float64")
