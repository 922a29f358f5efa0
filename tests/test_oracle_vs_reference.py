"""Pin oracle/step_torch.py against the reference's OWN Python (imported from /root/reference with
the stubs of oracle/ref_import.py).  Skipped where the tree is absent (the GPU box); there the same
pin is carried by the fixtures under tests/golden/ (tests/test_oracle_golden.py)."""
import copy

import pytest
import torch

from oracle import abn_torch, ref_import, step_torch as O

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    return ref_import.load_reference(abn_torch)


def _sd(module, dtype=torch.float64):
    return {k: v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone()
            for k, v in module.state_dict().items()}


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_student_forward_and_keys(ref):
    torch.manual_seed(1)
    net = ref.pspnet.Res_pspnet(ref.pspnet.BasicBlock, [2, 2, 2, 2], 19).double()
    mine = O.pspnet_init(O.STUDENT, 19, dtype=torch.float64)
    sd = net.state_dict()
    assert sorted(mine.keys()) == sorted(sd.keys())
    assert all(mine[k].shape == sd[k].shape for k in sd)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    P = _sd(net)
    x = torch.randn(2, 3, 97, 81, dtype=torch.float64) * 57
    net.train()
    want = net(x)
    got = O.pspnet_forward(P, x, O.STUDENT, True, dropout_p=0.0)
    for a, b in zip(got, want):
        assert a.shape == b.shape and _rel(a, b) < 1e-10
    after = _sd(net)
    for k in after:  # running statistics were updated identically
        assert _rel(P[k].double(), after[k].double()) < 1e-10, k
    net.eval()
    want = net(x)
    got = O.pspnet_forward(P, x, O.STUDENT, False)
    for a, b in zip(got, want):
        assert _rel(a, b) < 1e-10


def test_teacher_forward_and_keys(ref):
    torch.manual_seed(2)
    net = ref.pspnet.Res_pspnet(ref.pspnet.Bottleneck, [3, 4, 23, 3], 19).double().eval()
    mine = O.pspnet_init(O.TEACHER, 19, dtype=torch.float64)
    sd = net.state_dict()
    assert sorted(mine.keys()) == sorted(sd.keys())
    assert all(mine[k].shape == sd[k].shape for k in sd)
    x = torch.randn(1, 3, 65, 65, dtype=torch.float64) * 57
    with torch.no_grad():
        want = net(x)
        got = O.pspnet_forward(_sd(net), x, O.TEACHER, False)
    for a, b in zip(got, want):
        assert a.shape == b.shape and _rel(a, b) < 1e-9


def test_criteria(ref):
    g = torch.Generator().manual_seed(3)
    S = [torch.randn(2, 19, 33, 33, generator=g, dtype=torch.float64), torch.randn(2, 19, 33, 33, generator=g, dtype=torch.float64),
         torch.randn(2, 16, 33, 33, generator=g, dtype=torch.float64)] + [torch.zeros(1)] * 4
    T = [torch.randn(2, 19, 33, 33, generator=g, dtype=torch.float64), torch.randn(2, 19, 33, 33, generator=g, dtype=torch.float64),
         torch.randn(2, 40, 33, 33, generator=g, dtype=torch.float64)] + [torch.zeros(1)] * 4
    y = torch.randint(0, 19, (2, 129, 129), generator=g)
    y[0, :9] = 255
    C = ref.criterion
    assert _rel(O.criterion_dsn(S, y), C.CriterionDSN()(S, y)) < 1e-12
    assert _rel(O.criterion_pixel_wise(S, T), C.CriterionPixelWise()(S, T)) < 1e-12
    for scale in (0.5, 0.25, 0.1, 0.04):
        want = C.CriterionPairWiseforWholeFeatAfterPool(scale, -5)(S, T)
        # the reference force-casts to fp32 in similarity() (utils.py:174): compare at fp32 resolution
        assert _rel(O.criterion_pair_wise(S, T, scale, -5), want.double()) < 2e-6
    d_s, d_t = [torch.randn(2, 1, 1, 1, generator=g)], [torch.randn(2, 1, 1, 1, generator=g)]
    for kind in ("wgan-gp", "hinge"):
        assert _rel(O.criterion_adv(d_s, d_t, kind), C.CriterionAdv(kind)(d_s, d_t)) < 1e-6
        assert _rel(O.criterion_adv_for_g(d_s, kind), C.CriterionAdvForG(kind)(d_s, d_s)) < 1e-6
    with pytest.raises(ValueError):
        O.criterion_adv(d_s, d_t, "lsgan")
    with pytest.raises(ValueError):
        C.CriterionAdv("lsgan")


def test_discriminator_step_incl_gp_and_uv_quirk(ref):
    """kd_model.py:153-165: three D forwards, then ONE backward; u, v advance on every forward."""
    torch.manual_seed(4)
    D = ref.sagan.Discriminator(1, 19, 2, 65, 64).double()
    with torch.no_grad():
        D.attn1.gamma.fill_(0.3)
        D.attn2.gamma.fill_(-0.2)
    assert sorted(O.discriminator_init(dtype=torch.float64).keys()) == sorted(D.state_dict().keys())
    P = _sd(D)
    g = torch.Generator().manual_seed(5)
    pS = [torch.randn(2, 19, 65, 65, generator=g, dtype=torch.float64)]
    pT = [torch.randn(2, 19, 65, 65, generator=g, dtype=torch.float64)]
    alpha = torch.rand(2, 1, 1, 1, generator=g, dtype=torch.float64)
    C = ref.criterion
    # reference
    D.train()
    dT, dS = D(pT[0]), D(pS[0])
    loss = 0.1 * C.CriterionAdv("wgan-gp")(dS, dT)
    orig_rand = torch.rand
    try:
        torch.rand = lambda *a, **k: alpha.clone()
        with ref_import.cpu_cuda_identity():
            loss = loss + 0.1 * C.CriterionAdditionalGP(D, 10.0)(pS, pT)
    finally:
        torch.rand = orig_rand
    loss.backward()
    want = {k: p.grad for k, p in D.named_parameters() if p.requires_grad}
    # oracle
    O.require_grad(P)
    oT, oS = O.discriminator_forward(P, pT[0]), O.discriminator_forward(P, pS[0])
    mine = 0.1 * O.criterion_adv(oS, oT) + 0.1 * O.criterion_gp(P, pS, pT, 10.0, alpha)
    keys = O.learnable_keys(P)
    got = dict(zip(keys, torch.autograd.grad(mine, [P[k] for k in keys], allow_unused=True)))
    assert _rel(mine, loss) < 1e-10
    assert sorted(want.keys()) == sorted(keys)
    for k in keys:
        assert _rel(got[k], want[k]) < 1e-8, k
    after = _sd(D)
    for k in ("l1.0.module.weight_u", "l4.0.module.weight_v", "preprocess_additional.running_var"):
        assert _rel(P[k].detach(), after[k]) < 1e-10


def test_full_step_config1_plus_pa(ref):
    """BASELINE config 1 shape (B=2, 256x256, Pi) plus Pa, fp64, against the reference classes
    composed in the order of kd_model.py:119-151."""
    torch.manual_seed(6)
    S = ref.pspnet.Res_pspnet(ref.pspnet.BasicBlock, [2, 2, 2, 2], 19).double().train()
    T = ref.pspnet.Res_pspnet(ref.pspnet.Bottleneck, [3, 4, 23, 3], 19).double().eval()
    for m in S.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    PS, PT = _sd(S), _sd(T)
    x, y = O.synthetic_batch(2, 256, 256, dtype=torch.float64)
    C = ref.criterion
    with torch.no_grad():
        pT = T(x)
    pS = S(x)
    mc = C.CriterionDSN()(pS, y)
    pi = 10.0 * C.CriterionPixelWise()(pS, pT)
    pa = C.CriterionPairWiseforWholeFeatAfterPool(0.5, -5)(pS, pT)
    G = mc + pi + 0.5 * pa
    opt = torch.optim.SGD(S.parameters(), 1e-2, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()
    G.backward()
    want_g = {k: p.grad.clone() for k, p in S.named_parameters()}
    opt.step()
    cfg = O.StepConfig(pi=True, pa=True, ho=False, lambda_pa=0.5, weight_decay=5e-4, dropout_p=0.0)
    out = O.distillation_step(PS, PT, None, x, y, cfg)
    assert abs(out["mc_G_loss"] - float(mc)) < 1e-9 * abs(float(mc))
    assert abs(out["pi_G_loss"] - float(pi)) < 1e-9 * abs(float(pi))
    assert abs(out["pa_G_loss"] - float(pa)) < 1e-5 * abs(float(pa))  # reference computes Pa in fp32
    for k, gw in want_g.items():
        gg = out["grads_S"][k]
        assert gg is not None and float((gg - gw).norm()) <= 1e-5 * float(gw.norm()) + 1e-12, k
    after = _sd(S)
    for k in after:
        assert _rel(PS[k].double(), after[k].double()) < 1e-6, k
