"""Generates tests/golden/reference_vectors.pt by running the REFERENCE's own Python
(/root/reference, imported through oracle/ref_import.py) on seeded inputs.  Only runnable in the
build container; the fixture it writes travels with the repo so that the GPU box (which has no
reference tree) can still pin the oracle (tests/test_oracle_golden.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Weights come from oracle.step_torch.*_init(seed) (plain seeded torch CPU RNG) loaded into the
reference modules, so nothing large needs to be stored: the fixture holds inputs' seeds, the
reference's outputs (small tensors in full, large ones as strided samples + sums) and gradients'
norms / samples.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import abn_torch, ref_import, step_torch as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.pt")


def sample(t, n=64):
    """Deterministic strided sample + sum + L2 norm of a tensor."""
    f = t.detach().double().reshape(-1)
    step = max(1, f.numel() // n)
    return {"shape": list(t.shape), "step": step, "sample": f[::step][:n].clone(), "sum": f.sum().clone(), "norm": f.norm().clone()}


def load_into(module, P):
    module.load_state_dict({k: v.clone() for k, v in P.items()})
    return module


def main():
    ref = ref_import.load_reference(abn_torch)
    C = ref.criterion
    G = {}
    dt = torch.float64

    # 1. criteria on seeded tensors --------------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    S = [torch.randn(2, 19, 33, 33, generator=g, dtype=dt).requires_grad_(True), torch.randn(2, 19, 33, 33, generator=g, dtype=dt).requires_grad_(True),
         torch.randn(2, 24, 33, 33, generator=g, dtype=dt).requires_grad_(True)] + [None] * 4
    T = [torch.randn(2, 19, 33, 33, generator=g, dtype=dt), torch.randn(2, 19, 33, 33, generator=g, dtype=dt),
         torch.randn(2, 40, 33, 33, generator=g, dtype=dt)] + [None] * 4
    y = torch.randint(0, 19, (2, 129, 129), generator=g)
    y[0, :9] = 255
    crit = {"dsn": C.CriterionDSN()(S, y), "pixelwise": C.CriterionPixelWise()(S, T)}
    for scale in (0.5, 0.25, 0.1, 0.04):
        crit["pairwise_%g" % scale] = C.CriterionPairWiseforWholeFeatAfterPool(scale, -5)(S, T).double()
    total = crit["dsn"] + 10 * crit["pixelwise"] + 0.5 * crit["pairwise_0.5"] + 2.0 * crit["pairwise_0.1"]
    total.backward()
    G["criteria"] = {"seed": 11, "losses": {k: v.detach().clone() for k, v in crit.items()},
                     "grads": [sample(S[i].grad, 256) for i in range(3)]}

    # 2. networks ---------------------------------------------------------------------------------
    for name, arch, block, layers, hw in (("student", O.STUDENT, ref.pspnet.BasicBlock, [2, 2, 2, 2], (97, 81)),
                                          ("teacher", O.TEACHER, ref.pspnet.Bottleneck, [3, 4, 23, 3], (65, 65))):
        P = O.pspnet_init(arch, 19, seed=21, dtype=dt)
        net = load_into(ref.pspnet.Res_pspnet(block, layers, 19).double(), P)
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        x = torch.randn(2, 3, *hw, generator=torch.Generator().manual_seed(22), dtype=dt) * 57
        rec = {"init_seed": 21, "input_seed": 22, "hw": hw}
        if name == "student":
            net.train()
            outs = net(x)
            rec["train"] = [sample(o, 128) for o in outs]
            rec["running"] = {k: sample(v, 32) for k, v in net.state_dict().items() if "running" in k and k.startswith(("bn1", "layer4.1.bn2", "pspmodule.stages.0"))}
        net.eval()
        with torch.no_grad():
            outs = net(x)
        rec["eval"] = [sample(o, 128) for o in outs]
        G[name] = rec

    # 3. discriminator step (adv + WGAN-GP; three forwards, one backward) ------------------------
    PD = O.discriminator_init(seed=31, dtype=dt)
    PD["attn1.gamma"].fill_(0.3)
    PD["attn2.gamma"].fill_(-0.2)
    D = load_into(ref.sagan.Discriminator(1, 19, 2, 65, 64).double(), PD).train()
    g = torch.Generator().manual_seed(32)
    pS, pT = [torch.randn(2, 19, 65, 65, generator=g, dtype=dt)], [torch.randn(2, 19, 65, 65, generator=g, dtype=dt)]
    alpha = torch.rand(2, 1, 1, 1, generator=g, dtype=dt)
    dT, dS = D(pT[0]), D(pS[0])
    loss = 0.1 * C.CriterionAdv("wgan-gp")(dS, dT)
    orig = torch.rand
    try:
        torch.rand = lambda *a, **k: alpha.clone()
        with ref_import.cpu_cuda_identity():
            gp = C.CriterionAdditionalGP(D, 10.0)(pS, pT)
    finally:
        torch.rand = orig
    (loss + 0.1 * gp).backward()
    G["discriminator"] = {"init_seed": 31, "input_seed": 32, "d_out_T": dT[0].detach().clone(), "d_out_S": dS[0].detach().clone(),
                          "attn1_T": sample(dT[1]), "adv": loss.detach().clone(), "gp": gp.detach().clone(),
                          "hinge": C.CriterionAdv("hinge")(dS, dT).detach().clone(),
                          "grads": {k: sample(p.grad, 32) for k, p in D.named_parameters() if p.requires_grad},
                          "uv_after": {k: v.detach().clone() for k, v in D.state_dict().items() if k.endswith(("weight_u",)) }}

    # 4. one G step, BASELINE config 1 shape (B=2, 256x256) with Pi + Pa -------------------------
    PS, PT = O.pspnet_init(O.STUDENT, 19, seed=41, dtype=dt), O.pspnet_init(O.TEACHER, 19, seed=42, dtype=dt)
    Snet = load_into(ref.pspnet.Res_pspnet(ref.pspnet.BasicBlock, [2, 2, 2, 2], 19).double(), PS).train()
    Tnet = load_into(ref.pspnet.Res_pspnet(ref.pspnet.Bottleneck, [3, 4, 23, 3], 19).double(), PT).eval()
    for m in Snet.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    x, yy = O.synthetic_batch(2, 256, 256, seed=43, dtype=dt)
    with torch.no_grad():
        pT = Tnet(x)
    pS = Snet(x)
    mc, pi, pa = C.CriterionDSN()(pS, yy), 10.0 * C.CriterionPixelWise()(pS, pT), C.CriterionPairWiseforWholeFeatAfterPool(0.5, -5)(pS, pT)
    opt = torch.optim.SGD(Snet.parameters(), 1e-2, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()
    (mc + pi + 0.5 * pa).backward()
    grads = {k: sample(p.grad, 16) for k, p in Snet.named_parameters()}
    opt.step()
    G["step_config1_pa"] = {"seeds": (41, 42, 43), "mc": mc.detach().clone(), "pi": pi.detach().clone(), "pa": pa.detach().double().clone(),
                            "grads": grads, "after": {k: sample(v, 16) for k, v in Snet.state_dict().items() if k in ("conv1.weight", "head.weight", "layer3.0.conv1.weight", "bn1.running_var")}}
    torch.save(G, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
