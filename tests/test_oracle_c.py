"""Pins the plain-C restatement (oracle/libskd_ref.so) to the torch restatements that
tests/test_oracle_vs_reference.py pins to the reference's own Python: same closed forms, fp64."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from oracle import abn_torch, cref, step_torch as O
from structure_knowledge_distillation_amd import _lib


@pytest.fixture(scope="module")
def ref():
    return cref.load(_lib.SIGNATURES)


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("act", ["none", "leaky_relu", "elu"])
@pytest.mark.parametrize("shape", [(2, 3, 5), (4, 6, 37), (1, 2, 300)])
def test_abn_c_vs_autograd(ref, act, shape):
    N, C, S = shape
    code = {"none": 0, "leaky_relu": 1, "elu": 2}[act]
    g = torch.Generator().manual_seed(S)
    x = torch.randn(N, C, S, generator=g) * 2 + 1
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    w[0] = -w[0].abs()
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    xo = x.double().requires_grad_(True)
    wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
    rmo, rvo = rm.double(), rv.double()
    zo = abn_torch.abn_autograd(xo, wo, bo, rmo, rvo, True, 0.1, 1e-5, act, 0.01)
    dz = torch.randn(N, C, S, generator=g)
    zo.backward(dz.double())
    z, m, v = x.clone(), torch.empty(C), torch.empty(C)
    ws = torch.empty(max(1, ref.skd_abn_workspace_floats(N, C, S)))
    assert ref.skd_abn_forward_train(N, C, S, P(z), P(w), P(b), P(rm), P(rv), P(m), P(v), 0.1, 1e-5, code, 0.01, P(ws), None)
    assert rel(z, zo.detach()) < 1e-5 and rel(rm, rmo) < 1e-6 and rel(rv, rvo) < 1e-6
    dx, dw, db, e, ey = torch.empty_like(x), torch.zeros(C), torch.zeros(C), torch.empty(C), torch.empty(C)
    assert ref.skd_abn_backward(N, C, S, P(z), P(dz), P(v), P(w), P(b), P(e), P(ey), P(dx), P(dw), P(db), 1e-5, code, 0.01, 1, P(ws), None)
    assert rel(dx, xo.grad) < 2e-4 and rel(dw, wo.grad) < 2e-4 and rel(db, bo.grad) < 2e-4
    # and the hand-written backward formula restated in torch agrees with autograd
    fdx, fdw, fdb, _, _ = abn_torch.abn_backward_formula(zo.detach(), dz.double(), abn_torch.batch_stats(x.double())[1],
                                                         wo.detach(), bo.detach(), True, 1e-5, act, 0.01)
    assert rel(fdx, xo.grad) < 1e-9 and rel(fdw, wo.grad) < 1e-9 and rel(fdb, bo.grad) < 1e-9


def test_pixelwise_c_vs_torch(ref):
    g = torch.Generator().manual_seed(1)
    s, t = torch.randn(2, 19, 5, 7, generator=g) * 3, torch.randn(2, 19, 5, 7, generator=g) * 3
    so = s.double().requires_grad_(True)
    L = O.criterion_pixel_wise([so], [t.double()])
    L.backward()
    loss, grad = torch.empty(1), torch.empty_like(s)
    assert ref.skd_pixelwise_loss(2, 19, 35, P(s), P(t), P(loss), P(grad), P(torch.empty(1)), None)
    assert rel(loss, L.detach().reshape(1)) < 1e-6 and rel(grad, so.grad) < 1e-5


@pytest.mark.parametrize("H,W,scale", [(33, 33, 0.5), (65, 65, 0.5), (65, 65, 0.125), (46, 61, 0.5), (9, 9, 0.12)])
def test_pairwise_c_vs_torch(ref, H, W, scale):
    g = torch.Generator().manual_seed(H)
    B, Cs, Ct = 2, 6, 10
    fs, ft = torch.randn(B, Cs, H, W, generator=g), torch.randn(B, Ct, H, W, generator=g)
    fso = fs.double().requires_grad_(True)
    preds = lambda f: [None, None, f, None, None, None, None]
    L = O.criterion_pair_wise(preds(fso), preds(ft.double()), scale, -5)
    (0.5 * L).backward()
    kh, kw = O.pair_wise_pool_window(ft.shape, scale)
    kh, kw = max(kh, 1), max(kw, 1)
    OH, OW = -(-H // kh), -(-W // kw)
    M = OH * OW
    ps, pt = torch.empty(B, Cs, M), torch.empty(B, Ct, M)
    idx = torch.empty(B, Cs, M, dtype=torch.int32)
    assert ref.skd_maxpool_argmax(B * Cs, H, W, kh, kw, P(fs), P(ps), P(idx), None)
    assert ref.skd_maxpool_argmax(B * Ct, H, W, kh, kw, P(ft), P(pt), None, None)
    tp, ti = F.max_pool2d(fs, (kh, kw), (kh, kw), 0, ceil_mode=True, return_indices=True)
    assert torch.equal(ti.reshape(B, Cs, M).int(), idx) and torch.equal(tp.reshape(B, Cs, M), ps)
    ldm, ldc = ref.skd_pairwise_ldm(M), 128
    fhs, fht = torch.empty(B, Cs, ldm), torch.empty(B, Ct, ldm)
    fst, nrm = torch.empty(B, ldm, ldc), torch.empty(B, M)
    assert ref.skd_channel_l2_normalise(B, Cs, M, P(ps), P(fhs), ldm, P(fst), ldc, P(nrm), None)
    assert ref.skd_channel_l2_normalise(B, Ct, M, P(pt), P(fht), ldm, None, 0, None, None)
    G, loss = torch.empty(B, ldm, ldm), torch.empty(1)
    assert ref.skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, P(fhs), P(fht), P(G), P(loss), P(torch.empty(1)), None)
    assert rel(loss, L.detach().reshape(1)) < 1e-5
    dp, dx = torch.empty(B, Cs, ldm), torch.empty(B, Cs, H, W)
    assert ref.skd_pairwise_backward(B, Cs, M, ldm, ldc, P(fst), P(G), P(nrm), P(torch.tensor([0.5])), P(dp), None)
    assert ref.skd_maxunpool_scatter(B * Cs, H, W, kh, kw, P(dp), ldm, P(idx), P(dx), None)
    assert rel(dx, fso.grad) < 1e-4


def test_spectral_c_vs_torch(ref):
    g = torch.Generator().manual_seed(2)
    W = torch.randn(12, 40, generator=g) * 0.1
    u, v = torch.randn(12, generator=g), torch.randn(40, generator=g)
    Pd = {"weight_bar": W.double().reshape(12, 10, 2, 2).requires_grad_(True), "weight_u": (u / u.norm()).double(),
          "weight_v": (v / v.norm()).double()}
    uc, vc = (u / u.norm()).clone(), (v / v.norm()).clone()
    w_t = O.spectral_weight(Pd, "")
    gw = torch.randn(12, 40, generator=g)
    (w_t.reshape(12, 40) * gw.double()).sum().backward()
    sig, wo = torch.empty(1), torch.empty(12, 40)
    assert ref.skd_spectral_norm_forward(12, 40, P(W), P(uc), P(vc), P(sig), P(wo), P(torch.empty(1)), None)
    assert rel(wo, w_t.detach().reshape(12, 40)) < 1e-5 and rel(uc, Pd["weight_u"]) < 1e-5 and rel(vc, Pd["weight_v"]) < 1e-5
    gwb = torch.empty(12, 40)
    assert ref.skd_spectral_norm_backward(12, 40, P(W), P(uc), P(vc), P(sig), P(gw), P(gwb), P(torch.empty(1)), None)
    assert rel(gwb, Pd["weight_bar"].grad.reshape(12, 40)) < 1e-4
