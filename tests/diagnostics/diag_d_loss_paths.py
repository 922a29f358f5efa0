"""GPU diagnostic: the critic step's LOSS on fixed logits through the four ways this package can evaluate it --
{spectral norms together, one wrapper at a time} x {MIOpen convolutions, deterministic im2col + rocBLAS} -- against the fp64 CPU
oracle on the same logits.  (profiles/r04l_d_loss_paths.txt: why test_full_step_b8_vs_golden's replica check has the tolerance it has.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import step_torch as O  # noqa: E402
from structure_knowledge_distillation_amd.networks import sagan_models  # noqa: E402
_together = sagan_models.normalize_together
from structure_knowledge_distillation_amd.utils import criterion as C  # noqa: E402
import structure_knowledge_distillation_amd as S  # noqa: E402

S.configure_miopen()
dev = torch.device("cuda", 0)
B = 8
PD = O.discriminator_init(seed=143)
PD["attn1.gamma"].fill_(0.25)
PD["attn2.gamma"].fill_(-0.5)
cfg = O.StepConfig(weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
for trial in range(3):
    g = torch.Generator().manual_seed(trial)
    pS, pT = torch.randn(B, 19, 65, 65, generator=g) * 4, torch.randn(B, 19, 65, 65, generator=g) * 4
    alpha = torch.rand(B, 1, 1, 1, generator=g)
    P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in PD.items()}
    ref, _ = O.discriminator_step(P, pS.double(), pT.double(), cfg, alpha.double())
    P32 = {k: v.clone() for k, v in PD.items()}
    ref32, _ = O.discriminator_step(P32, pS, pT, cfg, alpha)
    row = ["trial %d  f64 %.8f  cpu-fp32 rel %.1e" % (trial, ref, abs(ref32 - ref) / abs(ref))]
    for together in ("1", "0"):
        for det in (False, True):
            # (the SKD_SN_TOGETHER switch is gone since round 5: the arm is selected the way tests/test_host_cpu.py does it)
            sagan_models.normalize_together = _together if together == "1" else (lambda wrappers: False)
            D = sagan_models.Discriminator(1, 19, B, 65, 64).to(dev).train()
            D.load_state_dict({k: v.clone() for k, v in PD.items()})
            import contextlib
            ctx = torch.backends.cudnn.flags(enabled=False) if det else contextlib.nullcontext()
            # NOTE: det=True here means im2col convolutions WITHOUT torch.use_deterministic_algorithms, so that the spectral-norm arm decides
            with ctx:
                with torch.no_grad():
                    D(pS.to(dev))
                d_t, d_s = D(pT.to(dev)), D(pS.to(dev))
                loss = cfg.lambda_d * C.CriterionAdv("wgan-gp")(d_s, d_t) + cfg.lambda_d * C.CriterionAdditionalGP(D, cfg.lambda_gp)(
                    [pS.to(dev)], [pT.to(dev)], alpha=alpha.to(dev))
            row.append("together=%s %s rel %.1e" % (together, "im2col" if det else "miopen", abs(float(loss) - ref) / abs(ref)))
    print("   ".join(row), flush=True)
