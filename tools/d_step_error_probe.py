"""GPU tool: where does the discriminator step's fp32 gradient error come from?

    python tools/d_step_error_probe.py [batch] > gpurun_out/d_step_error.jsonl

One critic step (G-step forward, D(T), D(S), WGAN loss + gradient penalty, backward; kd_model.py:148-165) on seeded random
logits.  Reference: the same modules on the CPU in fp64 with the spectral normalisation written in stock torch ops.
Variants on the GPU (fp32): the product path; convolutions on im2col + rocBLAS (cudnn off); spectral normalisation in stock
torch ops instead of csrc/spectral.hip; both; and the CPU in fp32.  Per variant: loss, worst and median err / |g| over the
parameter gradients.  One JSON line per variant."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def torch_sn(w_bar, u, v):
    h = w_bar.shape[0]
    W = w_bar.view(h, -1)
    l2n = lambda t: t / (t.norm() + 1e-12)
    v.copy_(l2n(torch.mv(W.data.t(), u)))
    u.copy_(l2n(torch.mv(W.data, v)))
    sigma = u.dot(W.mv(v))
    return w_bar / sigma.expand_as(w_bar)


def step(D, pS, pT, alpha, C):
    for p in D.parameters():
        p.grad = None
    with torch.no_grad():
        D(pS)
    d_t, d_s = D(pT), D(pS)
    loss = 0.1 * C.CriterionAdv("wgan-gp")(d_s, d_t) + 0.1 * C.CriterionAdditionalGP(D, 10.0)([pS], [pT], alpha=alpha)
    loss.backward()
    return float(loss), {k: p.grad.detach().double().cpu() for k, p in D.named_parameters() if p.grad is not None}


if __name__ == "__main__":
    import torch
    from structure_knowledge_distillation_amd import functional as SF
    from structure_knowledge_distillation_amd.networks import sagan_models
    from structure_knowledge_distillation_amd.utils import criterion as C
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    D0 = sagan_models.Discriminator(1, 19, B, 65, 64).train()
    with torch.no_grad():
        D0.attn1.gamma.fill_(0.25)
        D0.attn2.gamma.fill_(-0.5)
    g = torch.Generator().manual_seed(1)
    pS, pT = torch.randn(B, 19, 65, 65, generator=g) * 3, torch.randn(B, 19, 65, 65, generator=g) * 3
    alpha = torch.rand(B, 1, 1, 1, generator=g)
    hip_sn = SF.spectral_normalize

    def run(name, device, dtype, sn, cudnn):
        SF.spectral_normalize = sn
        D = copy.deepcopy(D0).to(device=device, dtype=dtype)
        with torch.backends.cudnn.flags(enabled=cudnn):
            loss, grads = step(D, pS.to(device=device, dtype=dtype), pT.to(device=device, dtype=dtype), alpha.to(device=device, dtype=dtype), C)
        SF.spectral_normalize = hip_sn
        return name, loss, grads

    ref = run("cpu fp64", "cpu", torch.float64, torch_sn, True)
    for v in (run("cpu fp32, torch spectral", "cpu", torch.float32, torch_sn, True),
              run("gpu: product path (MIOpen + spectral.hip)", dev, torch.float32, hip_sn, True),
              run("gpu: im2col + rocBLAS convolutions, spectral.hip", dev, torch.float32, hip_sn, False),
              run("gpu: MIOpen, spectral in torch ops", dev, torch.float32, torch_sn, True),
              run("gpu: im2col + rocBLAS, spectral in torch ops", dev, torch.float32, torch_sn, False)):
        errs = sorted(((float((v[2][k] - gr).norm() / (gr.norm() + 1e-30)), k) for k, gr in ref[2].items() if float(gr.norm()) > 1e-12), reverse=True)
        print(json.dumps({"variant": v[0], "loss": v[1], "loss_rel_err": abs(v[1] - ref[1]) / abs(ref[1]),
                          "worst_err_over_norm": errs[0][0], "worst_key": errs[0][1], "median": errs[len(errs) // 2][0],
                          "top5": [(k, float("%.3g" % e)) for e, k in errs[:5]]}), flush=True)
