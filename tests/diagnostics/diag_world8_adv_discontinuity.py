"""CPU analysis (round 4) of gpurun_out/r04q_world8_grads.pt (written by diag_world8_grads.py on the GPU box): why the world-8
Pi + Pa + Ho step's student gradients sit 10 % from the recorded 8-shard oracle although every kernel and every exchange is right.

  1. the sharded oracle (fp32, this host) with the gradients AT the student's outputs kept, per loss term;
  2. per rank: product's logits / d logits / d (post-pyramid feature) against the oracle's;
  3. per rank: the product's adversarial part of d logits against the fp64 critic differentiated (a) at the product's OWN logits and
     (b) at the oracle's logits.

Result kept in profiles/r04q_world8_adv_discontinuity.txt: (a) agrees to 2e-5 on all eight ranks; (a) and (b) -- the same fp64 function
at two sets of logits 1.4e-5 apart -- differ by 4e-3 .. 1.3e-2 on five ranks and by 1e-6 on the other three: LeakyReLU slopes of
the critic flipping for units within rounding distance of zero.

    python tests/diagnostics/diag_world8_adv_discontinuity.py [gpurun_out/r04q_world8_grads.pt] [batch seed of the dump: 13]

(The dump of round 4 was taken with the fixture's FIRST batch seed, 13; the fixture has since moved to a seed that keeps the pyramid
stages away from their leaky ReLU's kink -- part 4 below shows the unit that sat on it.)
"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import step_torch as O  # noqa: E402

spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tests", "golden", "make_golden_gpu_suite.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def oracle_outputs():
    x, y, alpha, shards = gen.sharded8_inputs()
    cfg = O.StepConfig(weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    PS, PT, PD = gen.init_nets("sharded8", torch.float32)
    O.require_grad(PS, True)
    with torch.no_grad():
        preds_T = O.pspnet_forward(PT, x, O.TEACHER, False)
    preds_S = O.pspnet_forward(PS, x, O.STUDENT, True, cfg.dropout_p)
    parts = []
    for sl in shards:
        P = O.require_grad({k: v.detach().clone() for k, v in PD.items()}, True)
        s, t = [p[sl] for p in preds_S], [p[sl] for p in preds_T]
        parts.append((O.criterion_dsn(s, y[sl]), cfg.lambda_pi * O.criterion_pixel_wise(s, t),
                      cfg.lambda_pa * O.criterion_pair_wise(s, t, cfg.pool_scale, -5),
                      cfg.lambda_d * O.criterion_adv_for_g(O.discriminator_forward(P, s[0]), cfg.adv_loss_type)))
    res = {}
    for i, name in enumerate(("mc", "pi", "pa", "adv")):
        res[name] = [None if g is None else g.detach() for g in
                     torch.autograd.grad(sum(p[i] for p in parts), preds_S[:3], retain_graph=True, allow_unused=True)]
    res["all"] = [sum(res[n][j] for n in res if res[n][j] is not None) for j in range(3)]
    return {"res": res, "logits": preds_S[0].detach(), "dsn": preds_S[1].detach(), "PD": PD, "cfg": cfg}


def pyramid_flip(g):
    """4. the pyramid's 1 x 1 stage: y in front of the leaky ReLU from the oracle's pooled features and from the product's."""
    x, _, _, _ = gen.sharded8_inputs()
    PS, _, _ = gen.init_nets("sharded8", torch.float32)
    with torch.no_grad():
        x4 = O.pspnet_forward(PS, x, O.STUDENT, True, 0.0)[3]
    W = PS["pspmodule.stages.0.1.weight"].double().reshape(128, 512)

    def y_of(pooled):
        t = pooled.double() @ W.t()
        return (t - t.mean(0)) / torch.sqrt(t.var(0, unbiased=False) + 1e-5), t.var(0, unbiased=False)

    y_o, var = y_of(x4.double().mean((2, 3)))
    y_p, _ = y_of(torch.cat([k["x4_mean"] for k in g]))
    print("pyramid 1 x 1 stage: variance over the 16 samples, median %.2e (eps 1e-5); smallest |y| of the oracle: %s" % (
        float(var.median()), ["%.1e" % float(v) for v in y_o.abs().flatten().sort().values[:5]]))
    print("y from the product's pooled x4 vs from the oracle's: max |difference| %.2e" % float((y_p - y_o).abs().max()))
    for i, j in ((y_p < 0) != (y_o < 0)).nonzero().tolist():
        print("SIGN FLIP at (sample %d, channel %d): oracle y = %.2e, product y = %.2e -> leaky slope 0.01 <-> 1" % (i, j, float(y_o[i, j]), float(y_p[i, j])))
    print("pyramid margins (min |y| of stages 1, 2, 3, 6) of this batch:", ["%.1e" % m for m in gen.pyramid_margins(PS, x)])


def main(path):
    g = torch.load(path)
    pyramid_flip(g)
    o = oracle_outputs()
    allg = o["res"]["all"]
    for r in range(8):
        sl = slice(2 * r, 2 * r + 2)
        k = g[r]
        print("rank %d: logits %.1e  dsn logits %.1e | d logits %.2e  d dsn %.2e  d feat_psp (fp16 copy) %.2e" % (
            r, rel(k["logits"], o["logits"][sl]), rel(k["dsn"], o["dsn"][sl]), rel(k["d_logits"], allg[0][sl]),
            rel(k["d_dsn"], allg[1][sl]), rel(k["d_feat_psp_half"].float(), allg[2][sl])))
    PD64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in o["PD"].items()}

    def adv_grad(logits):
        l = logits.double().clone().requires_grad_(True)
        P = {k: v.detach().clone() for k, v in PD64.items()}
        return torch.autograd.grad(o["cfg"].lambda_d * O.criterion_adv_for_g(O.discriminator_forward(P, l), o["cfg"].adv_loss_type), l)[0]

    for r in range(8):
        sl = slice(2 * r, 2 * r + 2)
        own, theirs = adv_grad(g[r]["logits"]), adv_grad(o["logits"][sl])
        smooth = allg[0][sl].double() - o["res"]["adv"][0][sl].double()
        prod_adv = g[r]["d_logits"].double() - smooth
        print("rank %d: fp64 critic gradient at the product's logits vs at the oracle's %.2e | product's adversarial part vs the fp64 critic "
              "at ITS OWN logits %.2e, at the oracle's %.2e" % (r, rel(own, theirs), rel(prod_adv, own), rel(prod_adv, theirs)))


if __name__ == "__main__":
    gen.SEEDS["sharded8"]["batch"] = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04q_world8_grads.pt"))
