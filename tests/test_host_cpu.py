"""Host-side logic on a box without a GPU: the product's autograd wiring, module surface and
NetModel step are run with oracle/libskd_ref.so installed as the C-ABI double (same entry points,
host pointers) and compared with the torch oracle.  No compute goes through libskd_hip.so here."""
import argparse
import os

import pytest
import torch

import structure_knowledge_distillation_amd.networks.pspnet_combine as PC_MOD  # noqa: E402  (its fused-form constants are patched by tests)

from oracle import abn_torch, cref, step_torch as O
from structure_knowledge_distillation_amd import _lib


@pytest.fixture(autouse=True)
def c_double():
    _lib.install_test_backend(cref.load(_lib.SIGNATURES))
    yield
    _lib.install_test_backend(None)


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_abn_module_autograd_and_inplace_contract():
    from structure_knowledge_distillation_amd import libs
    torch.manual_seed(0)
    for act in ("none", "leaky_relu", "elu"):
        mod = libs.InPlaceABNSync(6, activation=act).train()
        with torch.no_grad():
            mod.weight.copy_(torch.randn(6)); mod.bias.copy_(torch.randn(6))
        x = torch.randn(3, 6, 5, 4)
        gz = torch.randn(3, 6, 5, 4)
        xo = x.double().requires_grad_(True)
        wo, bo = mod.weight.detach().double().requires_grad_(True), mod.bias.detach().double().requires_grad_(True)
        rm, rv = torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
        zo = abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, act, 0.01)
        zo.backward(gz.double())
        xg = x.clone().requires_grad_(True)
        inp = xg * 1.0
        z = mod(inp)
        assert z.data_ptr() == inp.data_ptr()
        z.backward(gz)
        assert rel(z, zo) < 1e-5 and rel(xg.grad, xo.grad) < 1e-4
        assert rel(mod.weight.grad, wo.grad) < 1e-4 and rel(mod.bias.grad, bo.grad) < 1e-4
        assert rel(mod.running_mean, rm) < 1e-6 and rel(mod.running_var, rv) < 1e-6
        with pytest.raises(RuntimeError):   # once_differentiable: no double backward (functions.py:112)
            xg2 = x.clone().requires_grad_(True)
            out = mod(xg2 * 1.0)
            (g1,) = torch.autograd.grad(out.sum(), xg2, create_graph=True)
            g1.sum().backward()
    with pytest.raises(ValueError):
        libs.InPlaceABN(6)(torch.randn(2, 6, 4, 4).transpose(2, 3))
    # wrappers and state-dict keys of libs/bn.py
    wr = libs.InPlaceABNSyncWrapper(4)
    assert sorted(wr.state_dict().keys()) == ["bn.bias", "bn.running_mean", "bn.running_var", "bn.weight"]
    assert libs.InPlaceABNSync(4, devices=[0, 1]).devices == [0, 1]


def test_criteria_host_wiring():
    from structure_knowledge_distillation_amd.utils import criterion as C
    g = torch.Generator().manual_seed(1)
    S = [torch.randn(2, 19, 17, 17, generator=g).requires_grad_(True), torch.randn(2, 19, 17, 17, generator=g).requires_grad_(True),
         torch.randn(2, 12, 17, 17, generator=g).requires_grad_(True)] + [None] * 4
    T = [torch.randn(2, 19, 17, 17, generator=g), torch.randn(2, 19, 17, 17, generator=g), torch.randn(2, 20, 17, 17, generator=g)] + [None] * 4
    So = [t.detach().double().requires_grad_(True) for t in S[:3]] + [None] * 4
    To = [t.double() for t in T[:3]] + [None] * 4
    y = torch.randint(0, 19, (2, 65, 65), generator=g)
    y[1, 5:9] = 255
    got = C.CriterionDSN()(S, y) + 10 * C.CriterionPixelWise()(S, T) + 0.5 * C.CriterionPairWiseforWholeFeatAfterPool(0.5, -5)(S, T)
    want = O.criterion_dsn(So, y) + 10 * O.criterion_pixel_wise(So, To) + 0.5 * O.criterion_pair_wise(So, To, 0.5, -5)
    assert abs(float(got) - float(want)) < 1e-5 * abs(float(want))
    got.backward(); want.backward()
    for a, b in zip(S[:3], So[:3]):
        assert rel(a.grad, b.grad) < 1e-4
    for scale in (0.25, 0.1):
        a = C.CriterionPairWiseforWholeFeatAfterPool(scale, -5)(S, T)
        b = O.criterion_pair_wise(So, To, scale, -5)
        assert abs(float(a) - float(b)) < 1e-5 * abs(float(b))
    # channels-last features (what NetModel hands the pair-wise criterion since round 5): pooled as they are
    # (skd_maxpool_argmax_nhwc), same loss bits, the gradient comes back channels-last with the same values
    for scale in (0.5, 0.1):
        ref = C.CriterionPairWiseforWholeFeatAfterPool(scale, -5)
        fs0 = S[2].detach().clone().requires_grad_(True)
        l0 = ref([None, None, fs0] + [None] * 4, T)
        l0.backward()
        fs1 = S[2].detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ft1 = T[2].contiguous(memory_format=torch.channels_last)
        l1 = ref([None, None, fs1] + [None] * 4, [None, None, ft1] + [None] * 4)
        l1.backward()
        assert float(l1) == float(l0)
        assert fs1.grad.is_contiguous(memory_format=torch.channels_last) and torch.equal(fs1.grad.contiguous(), fs0.grad)


def test_spectral_norm_wrapper_state_and_quirk():
    from structure_knowledge_distillation_amd.networks.spectral import SpectralNorm
    torch.manual_seed(2)
    conv = torch.nn.Conv2d(5, 7, 4, 2, 1)
    sn = SpectralNorm(conv)
    assert sorted(sn.state_dict().keys()) == ["module.bias", "module.weight_bar", "module.weight_u", "module.weight_v"]
    assert not conv.weight_u.requires_grad and not conv.weight_v.requires_grad and conv.weight_bar.requires_grad
    Pd = {"weight_bar": conv.weight_bar.detach().double().requires_grad_(True), "weight_u": conv.weight_u.detach().double().clone(),
          "weight_v": conv.weight_v.detach().double().clone()}
    x1, x2 = torch.randn(2, 5, 9, 9), torch.randn(2, 5, 9, 9)
    y = sn(x1).sum() + (sn(x2) ** 2).sum()        # two forwards, one backward
    y.backward()
    F = torch.nn.functional
    yo = F.conv2d(x1.double(), O.spectral_weight(Pd, ""), conv.bias.detach().double(), 2, 1).sum() + \
        (F.conv2d(x2.double(), O.spectral_weight(Pd, ""), conv.bias.detach().double(), 2, 1) ** 2).sum()
    yo.backward()
    assert rel(conv.weight_u, Pd["weight_u"]) < 1e-5 and rel(conv.weight_v, Pd["weight_v"]) < 1e-5
    assert abs(float(y) - float(yo)) < 1e-4 * abs(float(yo))
    assert rel(conv.weight_bar.grad, Pd["weight_bar"].grad) < 1e-4


def test_discriminator_normalises_its_weights_together_with_the_same_numbers(monkeypatch):
    """networks/spectral.py::normalize_together (3 launches for the discriminator's four weights instead of 12) against one
    wrapper at a time: three forwards then one backward (kd_model.py:153-165) -- same outputs, same u / v, same gradients."""
    from structure_knowledge_distillation_amd.networks import sagan_models
    res = {}
    together = sagan_models.normalize_together
    for flag in ("1", "0"):
        monkeypatch.setattr(sagan_models, "normalize_together", together if flag == "1" else (lambda wrappers: False))
        torch.manual_seed(11)
        D = sagan_models.Discriminator(1, 19, 2, 65, 64).train()
        with torch.no_grad():
            D.attn1.gamma.fill_(0.3)
        g = torch.Generator().manual_seed(12)
        a, b = torch.randn(2, 19, 65, 65, generator=g), torch.randn(2, 19, 65, 65, generator=g)
        with torch.no_grad():
            D(a)
        loss = D(b)[0].mean() - D(a)[0].mean()
        loss.backward()
        res[flag] = (float(loss), {k: v.detach().clone() for k, v in D.state_dict().items()},
                     {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None})
        assert not any(getattr(blk[0], "_prepared", False) for blk in (D.l1, D.l2, D.l3, D.l4))
    assert res["1"][0] == res["0"][0]
    for which in (1, 2):
        for k, v in res["0"][which].items():
            assert torch.equal(v, res["1"][which][k]), k
    # a forward that dies between normalize_together and a layer must not leave a wrapper that skips its next power step
    monkeypatch.setattr(sagan_models, "normalize_together", together)
    D = sagan_models.Discriminator(1, 19, 2, 65, 64).train()
    boom = RuntimeError("interrupted")
    def _raise(*a):
        raise boom
    monkeypatch.setattr(D.l2[0].module, "forward", _raise)
    with pytest.raises(RuntimeError):
        D(torch.randn(2, 19, 65, 65))
    monkeypatch.undo()
    assert any(getattr(blk[0], "_prepared", False) for blk in (D.l2, D.l3, D.l4))     # the aborted forward left flags behind ...
    u0 = D.l3[0].module.weight_u.detach().clone()
    D(torch.randn(2, 19, 65, 65))
    assert not torch.equal(u0, D.l3[0].module.weight_u) and not any(getattr(blk[0], "_prepared", False) for blk in (D.l1, D.l2, D.l3, D.l4))


def test_inference_fusion_equals_unfused_graph():
    """Under no_grad + eval the networks take the fused BN->ReLU / BN->(+res)->ReLU passes; with grad enabled
    they run the reference's op sequence.  Same numbers either way (teacher = Bottleneck, student = BasicBlock)."""
    from structure_knowledge_distillation_amd.networks import pspnet_combine as PC
    torch.manual_seed(7)
    for block, layers in ((PC.Bottleneck, [3, 4, 23, 3]), (PC.BasicBlock, [2, 2, 2, 2])):
        net = PC.Res_pspnet(block, layers, 19).eval()
        for m in net.modules():     # non-trivial running statistics / affine parameters
            if isinstance(m, PC.InPlaceABNSync):
                with torch.no_grad():
                    m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
                    m.weight.normal_(0, 1); m.bias.normal_(0, 0.2)
        x = torch.randn(1, 3, 65, 49) * 57
        with torch.no_grad():
            fused = net(x)
        plain = net(x.clone().requires_grad_(True))
        for a, b in zip(fused, plain):
            assert rel(a, b) < 1e-5
        # the frozen stem runs eval BN + ReLU AFTER the max-pool (a monotone map commutes with max): the same BITS as before it
        with torch.no_grad():
            c3 = net.conv3(net.bn2.forward_relu(net.conv2(net.bn1.forward_relu(net.conv1(x)))))
            assert torch.equal(net.bn3.forward_relu(PC.SF.max_pool_stem(c3.clone(), net.maxpool)),
                               PC.SF.max_pool_stem(net.bn3.forward_relu(c3.clone()), net.maxpool))
    from structure_knowledge_distillation_amd import libs
    with pytest.raises(RuntimeError):
        libs.abn_eval_fused(torch.randn(1, 2, 3, 3, requires_grad=True) * 1.0, None, None, torch.zeros(2), torch.ones(2))


def _tiny_args(**kw):
    from structure_knowledge_distillation_amd.networks.kd_model import default_args
    return default_args(device=torch.device("cpu"), weight_decay=5e-4, lambda_pa=0.5, **kw)


def test_netmodel_surface_and_step_pi_pa():
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel
    torch.manual_seed(3)
    model = NetModel(_tiny_args(batch_size=2, ho=False))
    for attr in ("G_solver", "D_solver", "student", "adjust_learning_rate", "set_input", "optimize_parameters",
                 "print_info", "evalute_model", "save_ckpt", "G_loss", "mc_G_loss", "pi_G_loss", "pa_G_loss", "D_loss"):
        assert hasattr(model, attr), attr
    assert model.name() == "kd_seg"
    assert abs(model.adjust_learning_rate(1e-2, model.G_solver, 20000) - 1e-2 * 0.5 ** 0.9) < 1e-12
    assert model.G_solver.param_groups[0]["lr"] == model.lr_poly(1e-2, 20000, 40000, 0.9)
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    PS = {k: v.detach().clone() for k, v in model.student.state_dict().items()}
    PT = {k: v.detach().clone() for k, v in model.teacher.state_dict().items()}
    x, y = O.synthetic_batch(2, 96, 96)
    model.adjust_learning_rate(1e-2, model.G_solver, 0)
    model.set_input((x, y.float(), None, None))      # the loader hands float labels (datasets.py), .long() in set_input
    model.optimize_parameters()
    cfg = O.StepConfig(ho=False, weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    out = O.distillation_step(PS, PT, None, x, y, cfg)
    for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss"):
        assert abs(getattr(model, k) - out[k]) <= 2e-5 * abs(out[k]), k
    assert model.D_loss == 0.0
    assert all(not p.requires_grad for p in model.teacher.parameters())
    model.print_info(0, 0)


@pytest.mark.timeout(600)
def test_netmodel_full_step_with_ho_cpu_double():
    """Pi+Pa+Ho at B=2, 512x512 (D needs 65x65 logits) through the C double vs the torch oracle."""
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel
    torch.manual_seed(4)
    model = NetModel(_tiny_args(batch_size=2))
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    with torch.no_grad():
        model.D_model.attn1.gamma.fill_(0.25)
        model.D_model.attn2.gamma.fill_(-0.5)
    snap = lambda mod: {k: v.detach().clone() for k, v in mod.state_dict().items()}
    PS, PT, PD = snap(model.student), snap(model.teacher), snap(model.D_model)
    x, y = O.synthetic_batch(2, 512, 512)
    alpha = torch.rand(2, 1, 1, 1, generator=torch.Generator().manual_seed(5))
    model.gp_alpha = alpha
    model.set_input((x, y, None, None))
    model.optimize_parameters()
    assert all(p.requires_grad for p in model._d_params)
    out = O.distillation_step(PS, PT, PD, x, y, O.StepConfig(weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0), alpha=alpha)
    for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss"):
        assert abs(getattr(model, k) - out[k]) <= 5e-5 * abs(out[k]), (k, getattr(model, k), out[k])
    assert abs(model.D_loss - out["D_loss"]) <= 1e-3 * abs(out["D_loss"]), (model.D_loss, out["D_loss"])
    after = model.D_model.state_dict()
    for k in PD:
        if k.endswith(("weight_u", "weight_v")):
            assert rel(after[k], PD[k]) < 1e-4, k


def test_abn_relu_training_fusion_vs_autograd():
    """relu(bn(x) [+ residual]) fused (C double) == the reference's op sequence under autograd, fp64."""
    from structure_knowledge_distillation_amd import libs
    torch.manual_seed(11)
    for with_res in (False, True):
        mod = libs.InPlaceABNSync(5, activation="none").train()
        with torch.no_grad():
            mod.weight.copy_(torch.randn(5)); mod.bias.copy_(torch.randn(5) * 0.3)
        x, r, g = torch.randn(3, 5, 7, 6) * 2 + 0.5, torch.randn(3, 5, 7, 6), torch.randn(3, 5, 7, 6)
        xo, ro = x.double().requires_grad_(True), r.double().requires_grad_(True)
        wo, bo = mod.weight.detach().double().requires_grad_(True), mod.bias.detach().double().requires_grad_(True)
        rm, rv = torch.zeros(5, dtype=torch.float64), torch.ones(5, dtype=torch.float64)
        z = abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, "none", 0.01)
        want = torch.relu(z + ro if with_res else z)
        want.backward(g.double())
        xg, rg = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
        xin = xg * 1.0
        got = mod.forward_relu(xin, rg if with_res else None)
        assert got.data_ptr() != xin.data_ptr()            # out of place: the conv output is kept
        got.backward(g)
        assert rel(got, want) < 1e-5 and rel(xg.grad, xo.grad) < 1e-4
        assert rel(mod.weight.grad, wo.grad) < 1e-4 and rel(mod.bias.grad, bo.grad) < 1e-4
        assert rel(mod.running_mean, rm) < 1e-6 and rel(mod.running_var, rv) < 1e-6
        if with_res:
            assert rel(rg.grad, ro.grad) < 1e-5
    with pytest.raises(ValueError):
        libs.InPlaceABNSync(5, activation="leaky_relu").forward_relu(torch.randn(2, 5, 3, 3))


def test_channels_last_inference_abn_and_teacher():
    """NHWC inference path (skd_abn_apply_nhwc through the C double): same numbers as the NCHW path."""
    from structure_knowledge_distillation_amd import libs
    from structure_knowledge_distillation_amd.networks import pspnet_combine as PC
    torch.manual_seed(12)
    mod = libs.InPlaceABNSync(8, activation="none").eval()
    with torch.no_grad():
        mod.running_mean.normal_(0, 0.3); mod.running_var.uniform_(0.5, 1.5); mod.weight.normal_(0, 1); mod.bias.normal_(0, 0.2)
        x, r = torch.randn(2, 8, 5, 7), torch.randn(2, 8, 5, 7)
        for act, res in (("relu", None), ("relu", r), ("none", None), ("leaky_relu", r)):
            want = mod.fused_eval(x.clone(), act, res)
            xl = x.clone().contiguous(memory_format=torch.channels_last)
            got = mod.fused_eval(xl, act, res)          # residual given in NCHW: converted internally
            assert got.data_ptr() == xl.data_ptr() and got.is_contiguous(memory_format=torch.channels_last)
            assert rel(got, want) < 1e-6
        net = PC.Res_pspnet(PC.Bottleneck, [3, 4, 23, 3], 19).eval()
        for m in net.modules():
            if isinstance(m, PC.InPlaceABNSync):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.normal_(0, 1); m.bias.normal_(0, 0.2)
        img = torch.randn(1, 3, 65, 49) * 57
        want = net(img)
        net.to(memory_format=torch.channels_last)
        got = net(img.contiguous(memory_format=torch.channels_last))
        for a, b in zip(got, want):
            assert a.shape == b.shape and rel(a, b) < 1e-5


@pytest.mark.parametrize("kind", ["leaky_inplace", "relu_fused", "relu_fused_residual", "none_inplace"])
def test_channels_last_training_abn_vs_nchw(kind):
    """Channels-last training ABN (skd_abn_*_nhwc through the C double) == the NCHW path on the same numbers."""
    from structure_knowledge_distillation_amd import libs
    torch.manual_seed(21)
    act = {"leaky_inplace": "leaky_relu", "none_inplace": "none"}.get(kind, "none")
    x, r, g = torch.randn(3, 8, 6, 5) * 2 + 0.3, torch.randn(3, 8, 6, 5), torch.randn(3, 8, 6, 5)
    res = {}
    for fmt in ("nchw", "nhwc"):
        mod = libs.InPlaceABNSync(8, activation=act).train()
        with torch.no_grad():
            mod.weight.copy_(torch.linspace(-1, 1.5, 8)); mod.bias.copy_(torch.linspace(0.5, -0.5, 8))
        mf = torch.channels_last if fmt == "nhwc" else torch.contiguous_format
        xg = x.clone().contiguous(memory_format=mf).requires_grad_(True)
        rg = r.clone().contiguous(memory_format=mf).requires_grad_(True)
        xin = (xg * 1.0).contiguous(memory_format=mf)
        if kind.startswith("relu"):
            out = mod.forward_relu(xin, rg if kind.endswith("residual") else None)
        else:
            out = mod(xin)
            assert out.data_ptr() == xin.data_ptr()
        assert out.is_contiguous(memory_format=mf)
        out.backward(g.contiguous(memory_format=mf))
        res[fmt] = (out.detach(), xg.grad, mod.weight.grad, mod.bias.grad, mod.running_mean.clone(), mod.running_var.clone(),
                    rg.grad if kind.endswith("residual") else torch.zeros(1))
    for a, b in zip(res["nhwc"], res["nchw"]):
        assert rel(a, b) < 1e-5
    # 6 channels: not a power of two -> no channels-last training kernels; the reference takes any width, so the call
    # goes through the NCHW kernels on a copy (same numbers, result handed back channels-last) instead of raising
    x6 = torch.randn(2, 6, 4, 4)
    m6 = libs.InPlaceABN(6).train()
    a = m6(x6.clone().contiguous(memory_format=torch.channels_last))
    b = libs.InPlaceABN(6).train()(x6.clone())
    assert a.is_contiguous(memory_format=torch.channels_last) and rel(a, b) < 1e-6


def test_student_channels_last_training_graph():
    """The student network in channels-last (every ABN through the NHWC training entries, PPM / criteria fed NCHW
    copies) gives the same outputs and parameter gradients as the NCHW network."""
    from structure_knowledge_distillation_amd.networks import pspnet_combine as PC
    torch.manual_seed(31)
    net = PC.Res_pspnet(PC.BasicBlock, [2, 2, 2, 2], 19).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    import copy
    net2 = copy.deepcopy(net).to(memory_format=torch.channels_last)
    x = torch.randn(2, 3, 97, 65) * 57
    outs1 = net(x)
    outs2 = net2(x.contiguous(memory_format=torch.channels_last))
    g = [torch.randn_like(o) for o in outs1[:3]]
    sum((o * gg).sum() for o, gg in zip(outs1[:3], g)).backward()
    sum((o * gg).sum() for o, gg in zip(outs2[:3], g)).backward()
    for a, b in zip(outs2, outs1):
        assert a.shape == b.shape and rel(a, b) < 2e-5
    p1, p2 = dict(net.named_parameters()), dict(net2.named_parameters())
    for k in p1:
        # fp32 reassociation through the BN chains; a conv bias that feeds a BN has a zero true gradient (noise only)
        assert float((p2[k].grad - p1[k].grad).norm()) <= 5e-3 * float(p1[k].grad.norm()) + 1e-4, k
    b1, b2 = dict(net.named_buffers()), dict(net2.named_buffers())
    for k in b1:
        assert rel(b2[k], b1[k]) < 1e-5, k


def test_evaluate_main_whole_image_miou():
    """networks/evaluate.py:156-206 (whole=True) through the fused kernel (C double) vs the reference's numpy recipe."""
    import numpy as np
    from structure_knowledge_distillation_amd.networks import evaluate as E
    torch.manual_seed(41)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 7, 3, 2, 1)

        def forward(self, x):
            return [self.conv(x), None]

    net = Tiny()
    H, W = 24, 40
    batches = []
    for i in range(3):
        img = torch.randn(1, 3, H, W) * 10
        lab = torch.randint(0, 7, (1, H, W))
        lab[0, :3] = 255
        size = torch.tensor([[H - 2 * i, W - i, 3]])
        batches.append((img, lab, size, ["im%d" % i]))
    mean_iu, iu = E.evaluate_main(net, batches, "0", "512,512", 7, whole=True)
    cm = np.zeros((7, 7))
    for img, lab, size, _ in batches:
        with torch.no_grad():
            up = torch.nn.functional.interpolate(net(img)[0], size=(H, W), mode="bilinear", align_corners=True)
        pred = up[0].permute(1, 2, 0).numpy().argmax(2).astype(np.uint8)          # evaluate.py:112, 186
        hh, ww = int(size[0][0]), int(size[0][1])
        gt = lab[0].numpy()[:hh, :ww]
        keep = gt != 255
        cm += E.get_confusion_matrix(gt[keep], pred[:hh, :ww][keep], 7)           # evaluate.py:193-198
    want_mean, want_iu = E.iou_from_confusion(cm)
    assert np.allclose(iu, want_iu) and abs(mean_iu - want_mean) < 1e-12
    with pytest.raises(NotImplementedError):
        E.evaluate_main(net, batches, "0", "512,512", 7, whole=False)


def test_miopen_db_is_private_writable_copy_and_version_guard(tmp_path, monkeypatch):
    """The shipped find-db is never handed to MIOpen directly (it appends to its user db): a per-user / per-rank copy
    is; and a MIOpen build other than the one the db was tuned on is reported loudly instead of silently running
    untuned kernels."""
    import importlib
    import warnings
    import structure_knowledge_distillation_amd as S
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("MIOPEN_CUSTOM_CACHE_DIR", raising=False)
    monkeypatch.setenv("SKD_MIOPEN_CACHE", str(tmp_path))
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("MASTER_PORT", "29611")
    S = importlib.reload(S)
    assert "MIOPEN_USER_DB_PATH" not in os.environ, "importing the package must not touch the process environment (ADVICE r02)"
    stale = tmp_path / "miopen_db_000000000000_r0"          # a copy of a superseded database, untouched for 8 days: pruned
    stale.mkdir()
    os.utime(stale, (0, __import__("time").time() - 8 * 86400))
    assert S.configure_miopen() == os.environ["MIOPEN_USER_DB_PATH"]
    dst = os.environ["MIOPEN_USER_DB_PATH"]
    assert dst.startswith(str(tmp_path)) and dst.endswith("_r3_j29611") and os.path.isdir(os.path.join(dst, "cache"))
    assert not stale.exists() and os.environ["MIOPEN_DEBUG_CONV_WINOGRAD"] == "0"
    assert sorted(os.listdir(dst)) == sorted(os.listdir(S.MIOPEN_DB_DIR))
    assert S.MIOPEN_DB_VERSION == (3, 5, 0) and S.check_miopen_db() is True
    monkeypatch.setattr(S, "MIOPEN_DB_VERSION", (9, 9, 9))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert S.check_miopen_db() is False
    assert any("find-db" in str(x.message) for x in w)


@pytest.mark.parametrize("shape", [(2, 16, 8, 9, 11), (2, 8, 4, 6, 6), (3, 12, 4, 13, 7)])
def test_psp_fold_equals_concat_then_convolution(shape, monkeypatch):
    """PSPModule's folded bottleneck (conv3x3(feats) + fold(priors x W), csrc/ppm.hip) against the reference sequence
    pspnet_combine.py:104-111 -- upsample, cat, conv3x3 -- on the same module: outputs and every gradient."""
    import torch.nn.functional as F
    from structure_knowledge_distillation_amd.networks.pspnet_combine import PSPModule
    b, cf, cm, h, w = shape
    torch.manual_seed(5)
    m = PSPModule(cf, cm).train()
    for p in m.parameters():
        torch.nn.init.normal_(p, 0.0, 0.3)
    x = torch.randn(b, cf, h, w).contiguous(memory_format=torch.channels_last)

    def run(fold):
        monkeypatch.setattr(PC_MOD, "PSP_FOLD", bool(fold))
        for p in m.parameters():
            p.grad = None
        for mod in m.modules():                                   # same running statistics on both runs
            if hasattr(mod, "running_mean") and mod.running_mean is not None:
                mod.running_mean.zero_()
                mod.running_var.fill_(1.0)
        xx = x.clone().requires_grad_(True)
        torch.manual_seed(11)                                     # Dropout2d mask
        out = m(xx)
        torch.manual_seed(12)
        (out * torch.randn(out.shape)).sum().backward()
        return out.detach(), xx.grad, {k: v.grad.clone() for k, v in m.named_parameters()}

    o1, g1, p1 = run(True)
    o0, g0, p0 = run(False)
    assert rel(o1, o0) < 2e-5, rel(o1, o0)
    assert rel(g1, g0) < 2e-5, rel(g1, g0)
    for k in p0:
        assert rel(p1[k], p0[k]) < 5e-5, (k, rel(p1[k], p0[k]))
    # and the bottleneck convolution alone against plain torch ops in float64 (the reference graph itself)
    from structure_knowledge_distillation_amd import functional as SF
    with torch.no_grad():
        m.eval()
        pooled = [F.adaptive_avg_pool2d(x, st[0].output_size).contiguous(memory_format=torch.channels_last) for st in m.stages]
        priors = [st[2](st[1](p)) for st, p in zip(m.stages, pooled)]
        pri = [F.interpolate(p.double().contiguous(), size=(h, w), mode="bilinear", align_corners=True) for p in priors]
        want = F.conv2d(torch.cat(pri + [x.double().contiguous()], 1), m.bottleneck[0].weight.double(), None, 1, 1)
        got = SF.ppm_fold_bottleneck(priors, x, m.bottleneck[0].weight, {})
    assert rel(got, want) < 2e-5, rel(got, want)


@pytest.mark.parametrize("shape,ceil", [((2, 8, 16, 16), True), ((1, 4, 9, 12), True), ((2, 4, 7, 5), False), ((1, 12, 1, 2), True)])
def test_max_pool_stem_matches_torch(shape, ceil):
    """functional.max_pool_stem (the stem's MaxPool2d(3, 2, 1, ceil_mode), pspnet_combine.py:135) == the stock operator:
    values and gradient, with ties, -inf and NaN in the input (first maximum wins, NaN propagates)."""
    from structure_knowledge_distillation_amd import functional as SF
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randint(-3, 4, shape, generator=g).float()             # many ties
    x[0, 0, 0, 0] = float("-inf")
    if shape[2] > 2:
        x[0, 1, 2, 1] = float("nan")
    pool = torch.nn.MaxPool2d(3, 2, 1, ceil_mode=ceil)
    xa = x.contiguous(memory_format=torch.channels_last).clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya, yb = SF.max_pool_stem(xa, pool), pool(xb)
    assert ya.shape == yb.shape and ya.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(torch.nan_to_num(ya.detach(), nan=7e7), torch.nan_to_num(yb.detach(), nan=7e7))
    gy = torch.randn(yb.shape, generator=g)
    ya.backward(gy.contiguous(memory_format=torch.channels_last))
    yb.backward(gy)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-6, atol=1e-6)
    # plain NCHW tensors keep the stock operator
    assert torch.equal(torch.nan_to_num(SF.max_pool_stem(x, pool), nan=7e7), torch.nan_to_num(pool(x), nan=7e7))


def test_frozen_bottleneck_blas_tail_equals_conv_plus_abn(monkeypatch):
    """The frozen teacher's 1x1 reduce convolution + BN + ReLU and its stride-1 down-sample branch as library GEMMs with the
    folded BN in the epilogue (functional.conv1x1_bn_blas) == convolution + eval-mode InPlace-ABN (pspnet_combine.py:65-84)."""
    from structure_knowledge_distillation_amd.networks.pspnet_combine import Bottleneck, BatchNorm2d
    torch.manual_seed(2)
    down = torch.nn.Sequential(torch.nn.Conv2d(128, 512, 1, 1, bias=False), BatchNorm2d(512))
    blk = Bottleneck(128, 128, stride=1, dilation=2, downsample=down).eval()
    for mod in blk.modules():
        if getattr(mod, "running_mean", None) is not None:
            mod.running_mean.normal_(0, 0.5)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.normal_(0, 1.0)             # negative gammas too: the module uses |gamma| + eps
            mod.bias.data.normal_(0, 0.5)
    x = torch.randn(2, 128, 9, 7).contiguous(memory_format=torch.channels_last)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setattr(PC_MOD, "BLAS_TAILS", flag == "1")
        with torch.no_grad():
            outs[flag] = blk(x.clone())
    assert rel(outs["1"], outs["0"]) < 2e-6, rel(outs["1"], outs["0"])
    # the folded operands follow the parameters: change a running statistic -> the cache is rebuilt
    blk.bn1.running_var.mul_(2.0)
    for flag in ("1", "0"):
        monkeypatch.setattr(PC_MOD, "BLAS_TAILS", flag == "1")
        with torch.no_grad():
            outs[flag] = blk(x.clone())
    assert rel(outs["1"], outs["0"]) < 2e-6
    # with a graph (eval + grad) the reference sequence is used
    monkeypatch.setattr(PC_MOD, "BLAS_TAILS", True)
    y = blk(x.clone().requires_grad_(True))
    assert y.requires_grad


def test_frozen_bottleneck_fused_tail_equals_conv_plus_two_abn_passes(monkeypatch):
    """pspnet_combine.FUSED_TAIL: conv2 -> ONE GEMM with bn2 + ReLU in its prologue and bn3 + residual + ReLU in its epilogue
    (functional.conv1x1_abn_eval(pro=...), here on the C double) == convolution + the two in-place ABN passes
    (pspnet_combine.py:71-82); the packed bn2 constants are cached on the module and follow its tensors."""
    from structure_knowledge_distillation_amd import functional as SF
    from structure_knowledge_distillation_amd.networks.pspnet_combine import Bottleneck
    torch.manual_seed(4)
    blk = Bottleneck(128, 32, stride=1, dilation=2).eval()          # conv3: 32 -> 128 channels (K % 16 == 0, N % 128 == 0)
    for mod in blk.modules():
        if getattr(mod, "running_mean", None) is not None:
            mod.running_mean.normal_(0, 0.5)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.normal_(0, 1.0)
            mod.bias.data.normal_(0, 0.5)
    x = torch.randn(2, 128, 9, 7).contiguous(memory_format=torch.channels_last)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setattr(PC_MOD, "FUSED_TAIL", flag == "1")
        with torch.no_grad():
            outs[flag] = blk(x.clone(memory_format=torch.channels_last))
    assert rel(outs["1"], outs["0"]) < 2e-6, rel(outs["1"], outs["0"])
    pack = SF.abn_pack_eval_params(blk.bn2)
    assert SF.abn_pack_eval_params(blk.bn2) is pack                       # cached
    assert torch.equal(pack[0], blk.bn2.running_mean) and torch.allclose(pack[2], blk.bn2.weight.abs() + blk.bn2.eps)
    blk.bn2.running_var.mul_(3.0)                                         # in-place write -> version bump -> repacked
    assert SF.abn_pack_eval_params(blk.bn2) is not pack
    for flag in ("1", "0"):
        monkeypatch.setattr(PC_MOD, "FUSED_TAIL", flag == "1")
        with torch.no_grad():
            outs[flag] = blk(x.clone(memory_format=torch.channels_last))
    assert rel(outs["1"], outs["0"]) < 2e-6
    monkeypatch.setattr(PC_MOD, "FUSED_TAIL", True)                           # with a graph the reference sequence is used
    assert blk(x.clone(memory_format=torch.channels_last).requires_grad_(True)).requires_grad


def test_teacher_dsn_head_is_optional_and_everything_else_unchanged():
    """``model.teacher.skip_dsn = True`` (bench.py --dsn-ab's informative figure; never the default) skips the frozen teacher's
    deep-supervision head -- read by nothing but the teacher CE the reference computes and discards (kd_model.py:129):
    preds_T[1] is None, every loss of the step is bit-identical to the default.  ``model.log_teacher_ce`` computes that CE."""
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel
    vals = {}
    for skip in (False, True):
        torch.manual_seed(3)
        model = NetModel(_tiny_args(batch_size=2, ho=False))
        assert model.teacher.skip_dsn is False and model.log_teacher_ce is False        # the defaults: the reference's forward
        model.teacher.skip_dsn = skip
        x, y = O.synthetic_batch(2, 96, 96)
        torch.manual_seed(17)
        model.set_input((x, y, None, None))
        model.optimize_parameters()
        assert (model.preds_T[1] is None) == skip
        vals[skip] = [model.G_loss, model.mc_G_loss, model.pi_G_loss, model.pa_G_loss]
    assert vals[False] == vals[True], vals
    model = NetModel(_tiny_args(batch_size=2, ho=False))
    model.log_teacher_ce = True
    model.set_input((x, y, None, None))
    model.optimize_parameters()
    assert model.preds_T[1] is not None and model.mc_T_loss > 0


@pytest.mark.parametrize("channels_last", [False, True])
def test_classifier_head_is_conv2d_with_a_two_stage_bias_gradient(channels_last):
    """The 19-class heads (pspnet_combine.py:138-154) are ``nn.Conv2d`` with the same parameters and keys; only the REDUCTION of the
    bias gradient is split in two (rows first) -- same numbers as the stock module's backward up to summation order."""
    torch.manual_seed(0)
    head = PC_MOD.ClassifierConv(32, 19, 1, 1, 0, bias=True).double()
    stock = torch.nn.Conv2d(32, 19, 1, 1, 0, bias=True).double()
    stock.load_state_dict(head.state_dict())
    assert isinstance(head, torch.nn.Conv2d) and list(head.state_dict()) == ["weight", "bias"]
    x = torch.randn(3, 32, 9, 11, dtype=torch.double)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(3, 19, 9, 11, dtype=torch.double)
    ya, yb = head(xa), stock(xb)
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad) and torch.equal(head.weight.grad, stock.weight.grad)
    assert rel(head.bias.grad, stock.bias.grad) < 1e-14
    with torch.no_grad():
        assert torch.equal(head(x), stock(x))
    head.weight.requires_grad_(False)                     # frozen weight, live bias: no weight gradient is computed or returned
    head.zero_grad()
    head(x.clone().requires_grad_(True)).backward(g)
    assert head.weight.grad is None and rel(head.bias.grad, stock.bias.grad) < 1e-14
    net = PC_MOD.Res_pspnet(PC_MOD.BasicBlock, [2, 2, 2, 2], 19)
    assert isinstance(net.head, PC_MOD.ClassifierConv) and isinstance(net.dsn[3], PC_MOD.ClassifierConv)
    # constructions the two-stage form does not cover take the stock operator: same gradients as nn.Conv2d
    for kw in ({"groups": 2}, {"padding": "same", "kernel_size": 3}, {"padding": 1, "kernel_size": 3, "padding_mode": "reflect"}):
        args = dict(in_channels=32, out_channels=20, kernel_size=1, bias=True)
        args.update(kw)
        odd, ref = PC_MOD.ClassifierConv(**args).double(), torch.nn.Conv2d(**args).double()
        ref.load_state_dict(odd.state_dict())
        assert not odd._plain()
        xo, xr = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        go = torch.randn(3, 20, 9, 11, dtype=torch.double)
        odd(xo).backward(go)
        ref(xr).backward(go)
        assert torch.equal(xo.grad, xr.grad) and torch.equal(odd.weight.grad, ref.weight.grad) and torch.equal(odd.bias.grad, ref.bias.grad), kw


def test_device_identity_tells_physical_gpus_apart():
    """utils.parallel.device_identity: what SyncMailbox uses to count the ranks that share ITS device (their grid-barrier launches
    are capped at 1 / N of the compute units).  Eight ranks on eight GPUs must never look like eight ranks on one -- whether or not
    the runtime reports a UUID, whether or not the ranks mask their devices (every masked rank calls its GPU index 0)."""
    from structure_knowledge_distillation_amd.utils.parallel import device_identity

    class Props:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    zero = "00000000-0000-0000-0000-000000000000"
    gpus = [Props(pci_domain_id=0, pci_bus_id=0x05 + 0x10 * i, pci_device_id=0, uuid=zero) for i in range(8)]
    assert len({device_identity(p, 0) for p in gpus}) == 8                  # masked ranks, all-zero UUIDs: the PCI address decides
    assert device_identity(gpus[3], 0) == device_identity(gpus[3], 0)      # two ranks on ONE device agree
    with_uuid = [Props(uuid="58f3e8c2-%04d-0000-0000-000000000000" % i) for i in range(2)]
    assert device_identity(with_uuid[0], 0) != device_identity(with_uuid[1], 0)
    bare = Props()
    assert device_identity(bare, 2) == "index 2" and device_identity(bare, 2) != device_identity(bare, 3)


def test_reserved_fused_cap_leaves_compute_units_for_rccl(monkeypatch):
    """utils.parallel.reserved_fused_cap: with RCCL's kernels beside the step the grid-barrier launches must never cover the chip
    (a cross-rank cycle ABN pass -> peer's ABN pass -> peer's RCCL kernel -> own RCCL kernel -> own ABN pass, see
    reserve_for_collectives)."""
    from structure_knowledge_distillation_amd.utils.parallel import reserved_fused_cap
    monkeypatch.delenv("SKD_ABN_RCCL_RESERVE_CUS", raising=False)
    assert reserved_fused_cap(256) == 192 and reserved_fused_cap(304) == 192 and reserved_fused_cap(32) == 4
    monkeypatch.setenv("SKD_ABN_RCCL_RESERVE_CUS", "32")
    assert reserved_fused_cap(256) == 224
    monkeypatch.setenv("SKD_ABN_RCCL_RESERVE_CUS", "0")
    assert reserved_fused_cap(256) == 256


def test_sharded_evaluation_refuses_loaders_that_shard_or_shuffle():
    """networks/evaluate.py: with world > 1 every rank walks the SAME batch sequence and scores batch i on rank i % world; a loader
    that shuffles or already shards would be scored only in part, silently -- it must be refused with a message that
    names the sampler (the refusal itself used to die on `type(sampler)`: `type` is an argument of evaluate_main)."""
    from torch.utils.data import DataLoader, TensorDataset
    from torch.utils.data.distributed import DistributedSampler
    from structure_knowledge_distillation_amd.networks.evaluate import evaluate_main
    ds = TensorDataset(torch.zeros(4, 3, 8, 8), torch.zeros(4, 8, 8), torch.tensor([[8, 8, 3]] * 4))
    model = torch.nn.Conv2d(3, 19, 1)
    with pytest.raises(ValueError, match="RandomSampler"):
        evaluate_main(model, DataLoader(ds, batch_size=1, shuffle=True), "0", "8,8", 19, whole=True, rank=0, world=2)
    with pytest.raises(ValueError, match="DistributedSampler"):
        evaluate_main(model, DataLoader(ds, batch_size=1, sampler=DistributedSampler(ds, num_replicas=2, rank=0)), "0", "8,8", 19,
                      whole=True, rank=0, world=2)


def test_dist_timeout_and_miopen_find_switches(monkeypatch):
    """Two operational switches named in INTEGRATION.md: SKD_DIST_TIMEOUT_S becomes the process group's collective time-out
    (utils.parallel.init_distributed; bench.py asks for 300 s), SKD_MIOPEN_FIND=1 re-enables MIOpen's find mode for shapes the
    shipped find-db does not hold (NetModel sets torch.backends.cudnn.benchmark from it)."""
    import datetime
    import torch.distributed as dist
    from structure_knowledge_distillation_amd.utils import parallel as P
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel
    seen = {}
    monkeypatch.setattr(dist, "init_process_group", lambda **kw: seen.update(kw))
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("SKD_DIST_TIMEOUT_S", "7.5")
    assert P.init_distributed("gloo") == (1, 2, 1)
    assert seen["backend"] == "gloo" and seen["world_size"] == 2 and seen["timeout"] == datetime.timedelta(seconds=7.5)
    seen.clear()
    monkeypatch.delenv("SKD_DIST_TIMEOUT_S")
    P.init_distributed("gloo")
    assert "timeout" not in seen                                   # torch's default
    monkeypatch.undo()
    keep = torch.backends.cudnn.benchmark
    try:
        for flag, want in (("1", True), ("0", False)):
            monkeypatch.setenv("SKD_MIOPEN_FIND", flag)
            NetModel(_tiny_args(batch_size=2, ho=False))
            assert torch.backends.cudnn.benchmark is want
    finally:
        torch.backends.cudnn.benchmark = keep


@pytest.mark.parametrize("geom", [(2, 8, 9, 11), (1, 16, 8, 8), (2, 4, 5, 6), (1, 4, 1, 2)])
def test_fused_stem_on_the_double_equals_forward_relu_then_pool(geom):
    """libs.modules.forward_relu_maxpool (round 6; what ResNet.forward calls for the training student) on the C double: output bits,
    running statistics and gradients of forward_relu followed by the stem pool; inputs the fused form does not take (NCHW, eval,
    another pool) fall back to that sequence."""
    from structure_knowledge_distillation_amd import libs
    from structure_knowledge_distillation_amd import functional as SF
    B, C, H, W = geom
    torch.manual_seed(B + C + H)
    pool = torch.nn.MaxPool2d(3, 2, 1, ceil_mode=True)
    bn_a, bn_b = libs.InPlaceABNSync(C, activation="none").train(), libs.InPlaceABNSync(C, activation="none").train()
    with torch.no_grad():
        bn_a.weight.normal_(); bn_a.bias.normal_()
        bn_b.load_state_dict(bn_a.state_dict())
    x = (torch.randn(B, C, H, W) * 2 + 0.5).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone(memory_format=torch.channels_last).requires_grad_(True), x.clone(memory_format=torch.channels_last).requires_grad_(True)
    ya = bn_a.forward_relu_maxpool(xa * 1.0, pool)
    yb = SF.max_pool_stem(bn_b.forward_relu(xb * 1.0), pool)
    assert torch.equal(ya, yb) and torch.equal(bn_a.running_mean, bn_b.running_mean) and torch.equal(bn_a.running_var, bn_b.running_var)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    assert rel(xa.grad, xb.grad) < 1e-6 and rel(bn_a.weight.grad, bn_b.weight.grad) < 1e-6 and rel(bn_a.bias.grad, bn_b.bias.grad) < 1e-6
    # fall-backs: NCHW input, a pool that is not the stem's, eval mode
    xn = torch.randn(B, C, H, W)
    assert torch.equal(bn_a.forward_relu_maxpool(xn.clone(), pool), pool(bn_b.forward_relu(xn.clone())))
    other = torch.nn.MaxPool2d(2, 2)
    if H >= 2 and W >= 2:
        assert torch.equal(bn_a.forward_relu_maxpool(x.clone(memory_format=torch.channels_last), other),
                           other(bn_b.forward_relu(x.clone(memory_format=torch.channels_last))))
    with torch.no_grad():
        bn_a.eval(); bn_b.eval()
        assert torch.equal(bn_a.forward_relu_maxpool(x.clone(memory_format=torch.channels_last), pool),
                           SF.max_pool_stem(bn_b.forward_relu(x.clone(memory_format=torch.channels_last)), pool))
