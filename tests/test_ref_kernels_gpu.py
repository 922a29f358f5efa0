"""-m gpu: K1-K9 pinned to the REFERENCE'S OWN native kernels.

oracle/_ref/libbn_ref.so is libs/src/bn.cu + common.h + bn.h of the reference compiled for gfx950 by
oracle/build_ref.py (runtime-API names respelled, algorithm untouched; test-only, never imported by the product).
Three implementations run on the same seeded inputs:

    reference kernels (oracle/_ref)  vs  plain-C restatement (oracle/abn_ref.c)  vs  product (libskd_hip.so)

through the nine entry points that share the reference's argument lists (libs/src/bn.h:7-19 <-> include/skd.h
section 1).  Tolerances: the reference accumulates per-channel sums in fp32 (one thread strides a plane, then
32-wide shuffles), the oracle in double, the product in fp32 partials + double finalize, so reductions agree to
<= 2e-5 relative; element-wise results to <= 2e-5 of the tensor's maximum (dx: of the largest |dz| * gamma / sigma).
"""
import ctypes
import os

import pytest
import torch

from oracle import build_ref, cref
from structure_knowledge_distillation_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(2, 3, 1), (1, 5, 7), (3, 4, 36), (2, 19, 4225), (4, 64, 4225), (2, 7, 8193), (2, 16, 16641), (1, 3, 65536),
          (5, 130, 9), (2, 128, 4), (8, 128, 4225)]


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def libs():
    nat = build_ref.load()
    if nat is None:
        if build_ref.available():
            pytest.fail("oracle/_ref/libbn_ref.so is missing although the reference tree is present: run __graft_entry__.build()")
        pytest.skip("oracle/_ref/libbn_ref.so was not shipped (it is built where /root/reference exists)")
    return nat, cref.load(_lib.SIGNATURES), _lib.load()


def close(got, want, tol, what, floor=0.0):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = max(float(want.abs().max()), floor, 1e-30)
    err = float((got - want).abs().max()) / scale
    assert err <= tol, "%s: max err %.3e (rel to %.3e) > %.1e" % (what, err, scale, tol)


def inputs(N, C, S, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, S, generator=g) * 3.0 + torch.randn(1, C, 1, generator=g) * 5.0
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    if C >= 3:
        w[0] = 0.0            # bn.cu:217-223: zero weight -> no dweight contribution
        w[1] = -abs(w[1])     # gamma = |w| + eps
    dz = torch.randn(N, C, S, generator=g)
    return x, w, b, dz


@pytest.mark.parametrize("shape", SHAPES)
def test_bn_entries_three_way(libs, shape):
    nat, ref, hip = libs
    N, C, S = shape
    x, w, b, dz = inputs(N, C, S, N * 131 + C * 7 + S)
    g = lambda t: t.to(DEV)
    eps = 1e-5
    # K1 mean / biased variance (bn.cu:125-138)
    m_n, v_n, m_h, v_h = (torch.empty(C, device=DEV) for _ in range(4))
    m_r, v_r = torch.empty(C), torch.empty(C)
    xg = g(x)
    assert nat._bn_mean_var_cuda(N, C, S, P(xg), P(m_n), P(v_n), None)
    assert hip.skd_bn_mean_var(N, C, S, P(xg), P(m_h), P(v_h), None)
    assert ref.skd_bn_mean_var(N, C, S, P(x), P(m_r), P(v_r), None)
    xd = x.double()
    close(m_n, xd.mean((0, 2)).float(), 2e-5, "reference kernel mean vs fp64", floor=1.0)
    close(m_h, m_n, 2e-5, "mean: product vs reference kernel", floor=1.0)
    close(m_r, m_n, 2e-5, "mean: C oracle vs reference kernel", floor=1.0)
    close(v_h, v_n, 5e-5, "var: product vs reference kernel")
    close(v_r, v_n, 5e-5, "var: C oracle vs reference kernel")
    # K2 forward: y and z (bn.cu:140-165) from the SAME statistics
    mean, var = m_r.clone(), v_r.clone()
    y_n, z_n, y_h, z_h = (torch.empty(N, C, S, device=DEV) for _ in range(4))
    y_r, z_r = torch.empty(N, C, S), torch.empty(N, C, S)
    assert nat._bn_forward_cuda(N, C, S, P(xg), P(g(mean)), P(g(var)), P(g(w)), P(g(b)), P(y_n), P(z_n), eps, None)
    assert hip.skd_bn_forward(N, C, S, P(xg), P(g(mean)), P(g(var)), P(g(w)), P(g(b)), P(y_h), P(z_h), eps, None)
    assert ref.skd_bn_forward(N, C, S, P(x), P(mean), P(var), P(w), P(b), P(y_r), P(z_r), eps, None)
    close(y_h, y_n, 2e-6, "y: product vs reference kernel"); close(z_h, z_n, 2e-6, "z: product vs reference kernel")
    close(y_r, y_n, 2e-6, "y: C oracle vs reference kernel"); close(z_r, z_n, 2e-6, "z: C oracle vs reference kernel")
    # K3 edz / eydz (bn.cu:167-184) on the reference kernel's own z
    z = z_n.cpu()
    e_n, ey_n, e_h, ey_h = (torch.empty(C, device=DEV) for _ in range(4))
    e_r, ey_r = torch.empty(C), torch.empty(C)
    assert nat._bn_edz_eydz_cuda(N, C, S, P(z_n), P(g(dz)), P(g(w)), P(g(b)), P(e_n), P(ey_n), eps, None)
    assert hip.skd_bn_edz_eydz(N, C, S, P(z_n), P(g(dz)), P(g(w)), P(g(b)), P(e_h), P(ey_h), eps, None)
    assert ref.skd_bn_edz_eydz(N, C, S, P(z), P(dz), P(w), P(b), P(e_r), P(ey_r), eps, None)
    fl = float(dz.abs().max()) / max(1.0, (N * S) ** 0.5)          # a mean of N*S terms of magnitude |dz|
    close(e_h, e_n, 1e-4, "edz: product vs reference kernel", floor=fl); close(e_r, e_n, 1e-4, "edz: C oracle vs reference kernel", floor=fl)
    yfl = fl * float(y_r.abs().max())
    close(ey_h, ey_n, 1e-4, "eydz: product vs reference kernel", floor=yfl); close(ey_r, ey_n, 1e-4, "eydz: C oracle vs reference kernel", floor=yfl)
    # K4 backward (bn.cu:186-232): dx, dweight += , dbias +=   (same edz / eydz for all three)
    e, ey = e_r.clone(), ey_r.clone()
    seed_w, seed_b = torch.randn(C), torch.randn(C)                # the entries ACCUMULATE into dweight / dbias
    dx_n, dx_h = torch.empty(N, C, S, device=DEV), torch.empty(N, C, S, device=DEV)
    dw_n, db_n, dw_h, db_h = g(seed_w), g(seed_b), g(seed_w), g(seed_b)
    dx_r, dw_r, db_r = torch.empty(N, C, S), seed_w.clone(), seed_b.clone()
    assert nat._bn_backward_cuda(N, C, S, P(g(dz)), P(z_n), P(g(var)), P(g(w)), P(g(b)), P(g(e)), P(g(ey)), P(dx_n), P(dw_n), P(db_n), eps, None)
    assert hip.skd_bn_backward(N, C, S, P(g(dz)), P(z_n), P(g(var)), P(g(w)), P(g(b)), P(g(e)), P(g(ey)), P(dx_h), P(dw_h), P(db_h), eps, None)
    assert ref.skd_bn_backward(N, C, S, P(dz), P(z), P(var), P(w), P(b), P(e), P(ey), P(dx_r), P(dw_r), P(db_r), eps, None)
    mul = float(((w.abs() + eps) / torch.sqrt(var + eps)).max())
    close(dx_h, dx_n, 2e-5, "dx: product vs reference kernel", floor=float(dz.abs().max()) * mul)
    close(dx_r, dx_n, 2e-5, "dx: C oracle vs reference kernel", floor=float(dz.abs().max()) * mul)
    close(dw_h, dw_n, 1e-6, "dweight: product vs reference kernel", floor=1.0); close(db_h, db_n, 1e-6, "dbias: product vs reference kernel", floor=1.0)
    close(dw_r, dw_n, 1e-6, "dweight: C oracle vs reference kernel", floor=1.0); close(db_r, db_n, 1e-6, "dbias: C oracle vs reference kernel", floor=1.0)
    assert float(dw_n[0]) == float(seed_w[0]) if C >= 3 else True                 # weight == 0: untouched (bn.cu:219-222)
    # dx == NULL: only the parameter gradients (bn.cu:197)
    dw2_n, dw2_h = g(seed_w), g(seed_w)
    assert nat._bn_backward_cuda(N, C, S, P(g(dz)), P(z_n), P(g(var)), P(g(w)), P(g(b)), P(g(e)), P(g(ey)), None, P(dw2_n), None, eps, None)
    assert hip.skd_bn_backward(N, C, S, P(g(dz)), P(z_n), P(g(var)), P(g(w)), P(g(b)), P(g(e)), P(g(ey)), None, P(dw2_h), None, eps, None)
    close(dw2_h, dw2_n, 1e-6, "dweight only", floor=1.0)


@pytest.mark.parametrize("shape", [(4, 64, 4225), (8, 128, 4225), (2, 16, 16641), (3, 8, 36)])
def test_mean_var_with_an_outlier_where_the_old_pivot_sat(libs, shape):
    """K1 robustness: the product's statistics are ONE pass around a per-channel pivot; the reference
    is two-pass (bn.cu:125-138).  One element 300 sigma off the channel mean at (n, h, w) = (0, 0, 0) -- the element the
    single-sample pivot of rounds 1-2 used -- cost 6e-4 of relative accuracy on the variance; the median-of-three pivot
    (csrc/abn.hip: median3) keeps it at the fp32 level, for the NCHW entry and the channels-last entry alike."""
    nat, ref, hip = libs
    N, C, S = shape
    g = torch.Generator().manual_seed(N * 1000 + C)
    x = torch.randn(N, C, S, generator=g) * 3.0 + torch.randn(1, C, 1, generator=g) * 5.0
    x[0, :, 0] += 900.0                                             # 300 sigma
    want_m, want_v = x.double().mean((0, 2)), x.double().var((0, 2), unbiased=False)
    xg = x.to(DEV)
    m_n, v_n, m_h, v_h = (torch.empty(C, device=DEV) for _ in range(4))
    assert nat._bn_mean_var_cuda(N, C, S, P(xg), P(m_n), P(v_n), None)
    assert hip.skd_bn_mean_var(N, C, S, P(xg), P(m_h), P(v_h), None)
    rel = lambda a, b: float(((a.cpu().double() - b) / b).abs().max())
    print("outlier pivot %s: var rel err  reference kernel %.2e  product NCHW %.2e" % (shape, rel(v_n, want_v), rel(v_h, want_v)), end="")
    assert rel(v_n, want_v) < 2e-5, "reference kernel vs fp64"
    assert rel(v_h, want_v) < 2e-5 and float((m_h.cpu().double() - want_m).abs().max()) < 2e-5 * 10
    if C % 4 == 0 and not (C & (C - 1)):
        xl = x.permute(0, 2, 1).contiguous().to(DEV)                 # (rows = N*S, C): same element is row 0
        m_l, v_l = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ws = torch.empty(max(1, hip.skd_abn_nhwc_workspace_floats(N * S, C)), device=DEV)
        assert hip.skd_abn_stats_nhwc(N * S, C, P(xl), P(m_l), P(v_l), P(ws), None)
        print("  product NHWC %.2e" % rel(v_l, want_v), end="")
        assert rel(v_l, want_v) < 2e-5 and float((m_l.cpu().double() - want_m).abs().max()) < 2e-5 * 10
    print()


@pytest.mark.parametrize("n", [1, 63, 4097, 1 << 20])
def test_activation_entries_three_way(libs, n):
    """K5-K9 (bn.cu:302-377): in-place leaky-ReLU / ELU forward, their gradient rewrites, ELU inverse."""
    nat, ref, hip = libs
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * 2
    d = torch.randn(n, generator=g)
    for name, args in (("leaky_relu", (0.01,)), ("elu", ()), ("elu_inv", ())):
        src = x if name != "elu_inv" else x.clamp(min=-0.9)
        a, b, c = src.clone().to(DEV), src.clone().to(DEV), src.clone()
        assert getattr(nat, "_%s_cuda" % name)(n, P(a), *args, None)
        assert getattr(hip, "skd_%s" % name)(n, P(b), *args, None)
        assert getattr(ref, "skd_%s" % name)(n, P(c), *args, None)
        close(b, a, 1e-6, name + ": product vs reference kernel", floor=1.0); close(c, a, 1e-6, name + ": C oracle vs reference kernel", floor=1.0)
    for name, args in (("leaky_relu_backward", (0.01,)), ("elu_backward", ())):
        a, b, c = d.clone().to(DEV), d.clone().to(DEV), d.clone()
        assert getattr(nat, "_%s_cuda" % name)(n, P(x.to(DEV)), P(a), *args, None)
        assert getattr(hip, "skd_%s" % name)(n, P(x.to(DEV)), P(b), *args, None)
        assert getattr(ref, "skd_%s" % name)(n, P(x), P(c), *args, None)
        close(b, a, 1e-6, name + ": product vs reference kernel", floor=1.0); close(c, a, 1e-6, name + ": C oracle vs reference kernel", floor=1.0)


def test_ref_library_is_test_infrastructure_only():
    """The product never touches oracle/_ref: no product source names the library or its build recipe."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "structure_knowledge_distillation_amd")
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(d, f), errors="ignore").read()
                assert "libbn_ref" not in text and "build_ref" not in text and "oracle/_ref" not in text, os.path.join(d, f)
