"""CPU diagnostic (fp64 oracle only, no GPU): how smooth is the discriminator step in its inputs?

    python tests/diagnostics/diag_dstep_sensitivity.py

The batch-8 fixture's logits are recomputed with the oracle, the critic step (G-step forward, D(T), D(S), WGAN loss + gradient
penalty, backward) is evaluated in fp64, and again with the logits multiplied by (1 + 1e-6 * N(0, 1)) for six seeds.  Result
(profiles/r03_d_step_sensitivity.txt): the loss moves by 5e-9 ... 4e-6, the parameter gradients by 3e-6 in one seed and by
2e-4 ... 3e-3 in the other five (the input-side bias gradients and the first attention block first) -- the gradient penalty
differentiates through LeakyReLU's slope, a step function, so the gradient is discontinuous in the logits at the 1e-6
scale.  This is the bimodal "replica D step" error of tests/test_step_gpu.py (IM2COL_D_FLOOR): a property of the function
being compared, not of a kernel."""
import torch, os, sys, importlib.util, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import step_torch as O
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'golden')
gold=torch.load(os.path.join(G,'step_b8_oracle.pt'), weights_only=False)
spec=importlib.util.spec_from_file_location("gen", os.path.join(G,"make_golden_step_b8.py")); gen=importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
PS,PT,PD=gen.init(torch.float32)
B,H,W=gold['shape']
images, labels = O.synthetic_batch(B,H,W,seed=gold['seeds']['batch'])
t0=time.time()
with torch.no_grad():
    pT=O.pspnet_forward(PT, images, O.TEACHER, False)[0]
    pS=O.pspnet_forward(PS, images, O.STUDENT, True, 0.0)[0]
print("forward", time.time()-t0, pS.shape)
alpha=torch.rand(B,1,1,1,generator=torch.Generator().manual_seed(gold['seeds']['alpha']))
cfg=O.StepConfig(weight_decay=gold['cfg']['weight_decay'], lambda_pa=gold['cfg']['lambda_pa'], dropout_p=0.0)
def dstep(ps, pt, dt):
    P={k:(v.to(dt, copy=True) if v.is_floating_point() else v.clone()) for k,v in PD.items()}
    loss, grads = O.discriminator_step(P, ps.to(dt), pt.to(dt), cfg, alpha.to(dt))
    return loss, {k:g.double() for k,g in grads.items() if g is not None}
l0,g0=dstep(pS,pT,torch.float64)
for seed in range(6):
    gen_=torch.Generator().manual_seed(seed)
    ps2=pS*(1+1e-6*torch.randn(pS.shape,generator=gen_))
    pt2=pT*(1+1e-6*torch.randn(pT.shape,generator=gen_))
    l1,g1=dstep(ps2,pt2,torch.float64)
    rels={k:float((g1[k]-g0[k]).norm()/(g0[k].norm()+1e-30)) for k in g0 if float(g0[k].norm())>1e-12}
    w=sorted(rels.items(), key=lambda kv:-kv[1])[:3]
    print("perturbation seed %d: loss rel %.2e; worst gradient changes:"%(seed, abs(l1-l0)/abs(l0)), [(k,"%.2e"%v) for k,v in w], flush=True)
