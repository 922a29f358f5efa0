"""GPU diagnostic (test infrastructure, may use the oracle; not product code): per-parameter gradient error of the discriminator step, split into
the adversarial part and the WGAN-GP part, product (cuda fp32) vs CPU oracle (fp32 and fp64)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")
import torch  # noqa: E402

from oracle import step_torch as O  # noqa: E402
from structure_knowledge_distillation_amd.networks import sagan_models  # noqa: E402
from structure_knowledge_distillation_amd.utils import criterion as C  # noqa: E402

DEV = torch.device("cuda", 0)


def cpu_sd(mod, dtype):
    return {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in mod.state_dict().items()}


B = int(os.environ.get("DIAG_B", "2"))


def run(part):
    torch.manual_seed(3)
    D = sagan_models.Discriminator(1, 19, B, 65, 64).to(DEV).train()
    with torch.no_grad():
        D.attn1.gamma.fill_(0.3)
        D.attn2.gamma.fill_(-0.2)
    P32, P64 = cpu_sd(D, torch.float32), cpu_sd(D, torch.float64)
    gen = torch.Generator().manual_seed(4)
    pS, pT = torch.randn(B, 19, 65, 65, generator=gen), torch.randn(B, 19, 65, 65, generator=gen)
    alpha = torch.rand(B, 1, 1, 1, generator=gen)

    def oracle(P, dt):
        O.require_grad(P)
        loss = 0.0
        if part in ("adv", "both"):
            t_out = O.discriminator_forward(P, pT.to(dt))
            s_out = O.discriminator_forward(P, pS.to(dt))
            loss = loss + 0.1 * O.criterion_adv(s_out, t_out)
        if part in ("gp", "both"):
            loss = loss + 0.1 * O.criterion_gp(P, [pS.to(dt)], [pT.to(dt)], 10.0, alpha.to(dt))
        keys = O.learnable_keys(P)
        return float(loss), dict(zip(keys, torch.autograd.grad(loss, [P[k] for k in keys], allow_unused=True)))

    l64, g64 = oracle(P64, torch.float64)
    l32, g32 = oracle(P32, torch.float32)
    loss = 0.0
    if part in ("adv", "both"):
        d_t = D(pT.to(DEV))
        d_s = D(pS.to(DEV))
        loss = loss + 0.1 * C.CriterionAdv("wgan-gp")(d_s, d_t)
    if part in ("gp", "both"):
        loss = loss + 0.1 * C.CriterionAdditionalGP(D, 10.0)([pS.to(DEV)], [pT.to(DEV)], alpha=alpha.to(DEV))
    loss.backward()
    print("== %s: loss gpu %.8g cpu32 %.8g cpu64 %.8g" % (part, float(loss), l32, l64))
    named = dict(D.named_parameters())
    for k, gw in g64.items():
        if gw is None:
            print("%-34s oracle grad None; gpu %s" % (k, None if named[k].grad is None else float(named[k].grad.norm())))
            continue
        n = float(gw.norm()) + 1e-30
        eg = float((named[k].grad.detach().cpu().double() - gw).norm()) / n if named[k].grad is not None else float("nan")
        ec = float((g32[k].double() - gw).norm()) / n
        if n > 1e-12:
            print("%-34s |g| %.3e  rel err gpu %.2e  cpu32 %.2e" % (k, n, eg, ec))


if __name__ == "__main__":
    print("B =", B, "MIOPEN_DEBUG_CONV_WINOGRAD =", os.environ.get("MIOPEN_DEBUG_CONV_WINOGRAD"))
    for part in (sys.argv[1:] or ("adv", "gp", "both")):
        run(part)
