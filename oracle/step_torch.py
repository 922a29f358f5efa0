"""CPU restatement of the reference's distillation step as pure functions over flat
state-dicts (reference key names), in plain torch ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for the HIP path, and the
``cpu_baseline`` leg of bench.py.  Never imported by the product package.

Everything is a function of ``(P, ...)`` where ``P`` is a dict {reference state-dict key ->
tensor}; learnable entries are leaf tensors with ``requires_grad`` so ``torch.autograd`` gives
the gradients.  Running statistics / spectral-norm u, v are updated IN PLACE in ``P`` exactly
where the reference mutates its buffers.  Works in fp32 or fp64 (dtype follows ``P``).

What each function follows (all paths relative to the reference root):
  pspnet_forward        networks/pspnet_combine.py:114-189 (ResNet), 19-45 (BasicBlock),
                        47-84 (Bottleneck), 86-112 (PSPModule)
  abn                   libs/functions.py:70-110 + libs/src/bn.cu:125-165,302-315
                        (oracle/abn_torch.py holds the formula-level restatement)
  spectral_weight       networks/spectral.py:10-35
  self_attn             networks/sagan_models.py:22-41
  discriminator_forward networks/sagan_models.py:105-168
  criterion_dsn         utils/criterion.py:179-188
  criterion_pixel_wise  utils/criterion.py:219-226
  criterion_pair_wise   utils/criterion.py:236-245 + utils/utils.py:170-183
  criterion_adv_for_g   utils/criterion.py:129-137
  criterion_adv         utils/criterion.py:146-166
  criterion_gp          utils/criterion.py:98-120
  distillation_step     networks/kd_model.py:119-173 (+ SGD of kd_model.py:74-75)
  distillation_step_sharded   the same step under the reference's multi-GPU semantics: utils/parallel.py:106,142,155
                        (scatter, per-replica criteria, mean of the replica losses), libs/functions.py:185-209
                        (whole-batch student BN statistics), sagan_models.py:148 (D's BatchNorm stays local)
Pinned against the reference's own Python by tests/test_oracle_vs_reference.py (runs where
/root/reference exists) and by the fixtures in tests/golden/ (made by tests/golden/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F

from . import abn_torch

STUDENT = ("basic", (2, 2, 2, 2))      # Res_pspnet(BasicBlock, [2,2,2,2]) kd_model.py:57
TEACHER = ("bottleneck", (3, 4, 23, 3))  # Res_pspnet(Bottleneck, [3,4,23,3]) kd_model.py:62


# --------------------------------------------------------------------------------------------
# parameter construction (shapes of pspnet_combine.py / sagan_models.py; default torch init)
# --------------------------------------------------------------------------------------------
def _conv_init(P, key, cout, cin, k, bias, gen, dtype):
    # nn.Conv2d default init: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    fan_in = cin * k * k
    bound = 1.0 / math.sqrt(fan_in)
    P[key + ".weight"] = (torch.rand(cout, cin, k, k, generator=gen, dtype=dtype) * 2 - 1) * bound
    if bias:
        P[key + ".bias"] = (torch.rand(cout, generator=gen, dtype=dtype) * 2 - 1) * bound


def _abn_init(P, key, c, dtype):
    P[key + ".weight"] = torch.ones(c, dtype=dtype)          # libs/bn.py:86-91
    P[key + ".bias"] = torch.zeros(c, dtype=dtype)
    P[key + ".running_mean"] = torch.zeros(c, dtype=dtype)
    P[key + ".running_var"] = torch.ones(c, dtype=dtype)


def _layer_plan(arch):
    """[(layer name, planes, blocks, stride, dilation)] pspnet_combine.py:132-135."""
    _, layers = arch
    return [("layer1", 64, layers[0], 1, 1), ("layer2", 128, layers[1], 2, 1),
            ("layer3", 256, layers[2], 1, 2), ("layer4", 512, layers[3], 1, 4)]


def _expansion(arch):
    return 1 if arch[0] == "basic" else 4


def pspnet_init(arch, num_classes, seed=0, dtype=torch.float32):
    """Random-init state dict with the reference's key names (150 tensors student / 565 teacher)."""
    gen = torch.Generator().manual_seed(seed)
    P = {}
    _conv_init(P, "conv1", 64, 3, 3, False, gen, dtype); _abn_init(P, "bn1", 64, dtype)
    _conv_init(P, "conv2", 64, 64, 3, False, gen, dtype); _abn_init(P, "bn2", 64, dtype)
    _conv_init(P, "conv3", 128, 64, 3, False, gen, dtype); _abn_init(P, "bn3", 128, dtype)
    exp = _expansion(arch)
    inplanes = 128
    for name, planes, blocks, stride, _dil in _layer_plan(arch):
        for b in range(blocks):
            pre = "%s.%d" % (name, b)
            cin = inplanes if b == 0 else planes * exp
            if arch[0] == "basic":
                _conv_init(P, pre + ".conv1", planes, cin, 3, False, gen, dtype); _abn_init(P, pre + ".bn1", planes, dtype)
                _conv_init(P, pre + ".conv2", planes, planes, 3, False, gen, dtype); _abn_init(P, pre + ".bn2", planes, dtype)
            else:
                _conv_init(P, pre + ".conv1", planes, cin, 1, False, gen, dtype); _abn_init(P, pre + ".bn1", planes, dtype)
                _conv_init(P, pre + ".conv2", planes, planes, 3, False, gen, dtype); _abn_init(P, pre + ".bn2", planes, dtype)
                _conv_init(P, pre + ".conv3", planes * 4, planes, 1, False, gen, dtype); _abn_init(P, pre + ".bn3", planes * 4, dtype)
            if b == 0 and (stride != 1 or inplanes != planes * exp):
                _conv_init(P, pre + ".downsample.0", planes * exp, inplanes, 1, False, gen, dtype)
                _abn_init(P, pre + ".downsample.1", planes * exp, dtype)
        inplanes = planes * exp
    feat, out = (512, 128) if arch[0] == "basic" else (2048, 512)
    for i in range(4):
        _conv_init(P, "pspmodule.stages.%d.1" % i, out, feat, 1, False, gen, dtype)
        _abn_init(P, "pspmodule.stages.%d.2" % i, out, dtype)
    _conv_init(P, "pspmodule.bottleneck.0", out, feat + 4 * out, 3, False, gen, dtype)
    _abn_init(P, "pspmodule.bottleneck.1", out, dtype)
    _conv_init(P, "head", num_classes, out, 1, True, gen, dtype)
    _conv_init(P, "dsn.0", out, feat // 2, 3, True, gen, dtype)
    _abn_init(P, "dsn.1", out, dtype)
    _conv_init(P, "dsn.3", num_classes, out, 1, True, gen, dtype)
    return P


def discriminator_init(in_ch=19, conv_dim=64, seed=0, dtype=torch.float32):
    """sagan_models.py:108-153 (imsize 65, preprocess mode 1) -- 37 tensors."""
    gen = torch.Generator().manual_seed(seed)
    P = {}
    dims = [(in_ch, conv_dim), (conv_dim, conv_dim * 2), (conv_dim * 2, conv_dim * 4), (conv_dim * 4, conv_dim * 8)]
    for i, (ci, co) in enumerate(dims):
        tmp = {}
        _conv_init(tmp, "c", co, ci, 4, True, gen, dtype)
        pre = "l%d.0.module." % (i + 1)
        P[pre + "bias"] = tmp["c.bias"]
        u = torch.randn(co, generator=gen, dtype=dtype)          # spectral.py:53-56
        v = torch.randn(ci * 16, generator=gen, dtype=dtype)
        P[pre + "weight_u"] = u / (u.norm() + 1e-12)
        P[pre + "weight_v"] = v / (v.norm() + 1e-12)
        P[pre + "weight_bar"] = tmp["c.weight"]
    _conv_init(P, "last.0", 1, conv_dim * 8, 4, True, gen, dtype)
    for name, c in (("attn1", conv_dim * 4), ("attn2", conv_dim * 8)):
        _conv_init(P, name + ".query_conv", c // 8, c, 1, True, gen, dtype)
        _conv_init(P, name + ".key_conv", c // 8, c, 1, True, gen, dtype)
        _conv_init(P, name + ".value_conv", c, c, 1, True, gen, dtype)
        P[name + ".gamma"] = torch.zeros(1, dtype=dtype)           # sagan_models.py:19
    P["preprocess_additional.weight"] = torch.ones(in_ch, dtype=dtype)
    P["preprocess_additional.bias"] = torch.zeros(in_ch, dtype=dtype)
    P["preprocess_additional.running_mean"] = torch.zeros(in_ch, dtype=dtype)
    P["preprocess_additional.running_var"] = torch.ones(in_ch, dtype=dtype)
    P["preprocess_additional.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return P


NON_LEARNABLE_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v")


def learnable_keys(P):
    return [k for k in P if not k.endswith(NON_LEARNABLE_SUFFIXES)]


def require_grad(P, on=True):
    for k in learnable_keys(P):
        P[k].requires_grad_(on)
    return P


# --------------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------------
def abn(P, key, x, training, activation, momentum=0.1, eps=1e-5, slope=0.01, replicas=1, stats_fn=None):
    """InPlaceABNSync forward (libs/bn.py:165-184 -> functions.py:167-227), out of place.
    ``stats_fn(mean, var) -> (mean, var)`` models the cross-replica combine (functions.py:185-205)."""
    w, b = P[key + ".weight"], P[key + ".bias"]
    rm, rv = P[key + ".running_mean"], P[key + ".running_var"]
    stats = None
    if training and stats_fn is not None:
        m, v = abn_torch.batch_stats(x)
        stats = stats_fn(m, v)
    return abn_torch.abn_autograd(x, w, b, rm, rv, training, momentum, eps, activation, slope,
                                  replicas=replicas, stats=stats)


def _dropout2d(x, p, training, masks, key):
    """nn.Dropout2d(0.1) (pspnet_combine.py:99,143,153). ``masks`` {key: (N,C) keep mask} pins the RNG."""
    if not training or p == 0.0:
        return x
    if masks is not None and key in masks:
        keep = masks[key].to(x.dtype).view(x.shape[0], x.shape[1], 1, 1)
        return x * keep / (1.0 - p)
    return F.dropout2d(x, p, True)


def pspnet_forward(P, x, arch, training, dropout_p=0.1, dropout_masks=None, abn_kw=None):
    """Returns the 7-element list of pspnet_combine.py:189."""
    kw = abn_kw or {}
    bn = lambda key, t, act="none": abn(P, key, t, training, act, **kw)  # BatchNorm2d = ABNSync(activation='none'), :12
    x = F.relu(bn("bn1", F.conv2d(x, P["conv1.weight"], None, 2, 1)))
    x = F.relu(bn("bn2", F.conv2d(x, P["conv2.weight"], None, 1, 1)))
    x = F.relu(bn("bn3", F.conv2d(x, P["conv3.weight"], None, 1, 1)))
    x = F.max_pool2d(x, 3, 2, 1, ceil_mode=True)                         # :131
    feats = []
    for name, _planes, blocks, stride, dil in _layer_plan(arch):
        for b in range(blocks):
            pre = "%s.%d" % (name, b)
            s = stride if b == 0 else 1
            res = x
            if arch[0] == "basic":                                       # :34-45
                out = F.relu(bn(pre + ".bn1", F.conv2d(x, P[pre + ".conv1.weight"], None, s, dil, dil)))
                out = bn(pre + ".bn2", F.conv2d(out, P[pre + ".conv2.weight"], None, 1, dil, dil))
            else:                                                        # :65-84
                out = F.relu(bn(pre + ".bn1", F.conv2d(x, P[pre + ".conv1.weight"])))
                out = F.relu(bn(pre + ".bn2", F.conv2d(out, P[pre + ".conv2.weight"], None, s, dil, dil)))
                out = bn(pre + ".bn3", F.conv2d(out, P[pre + ".conv3.weight"]))
            if (pre + ".downsample.0.weight") in P:
                res = bn(pre + ".downsample.1", F.conv2d(x, P[pre + ".downsample.0.weight"], None, s))
            x = F.relu(out + res)
        feats.append(x)
        if name == "layer3":                                             # :183 dsn on x3
            d = F.conv2d(x, P["dsn.0.weight"], P["dsn.0.bias"], 1, 1)
            d = bn("dsn.1", d, "leaky_relu")
            d = _dropout2d(d, dropout_p, training, dropout_masks, "dsn.2")
            x_dsn = F.conv2d(d, P["dsn.3.weight"], P["dsn.3.bias"])
    x1, x2, x3, x4 = feats
    h, w = x4.shape[2], x4.shape[3]
    priors = []
    for i, size in enumerate((1, 2, 3, 6)):                              # :94, :108-110
        t = F.adaptive_avg_pool2d(x4, (size, size))
        t = bn("pspmodule.stages.%d.2" % i, F.conv2d(t, P["pspmodule.stages.%d.1.weight" % i]), "leaky_relu")
        priors.append(F.interpolate(t, size=(h, w), mode="bilinear", align_corners=True))
    cat = torch.cat(priors + [x4], 1)
    t = bn("pspmodule.bottleneck.1", F.conv2d(cat, P["pspmodule.bottleneck.0.weight"], None, 1, 1), "leaky_relu")
    feat_psp = _dropout2d(t, dropout_p, training, dropout_masks, "pspmodule.bottleneck.2")
    logits = F.conv2d(feat_psp, P["head.weight"], P["head.bias"])
    return [logits, x_dsn, feat_psp, x4, x3, x2, x1]


def l2normalize(v, eps=1e-12):
    return v / (v.norm() + eps)                                          # spectral.py:10-11


def spectral_weight(P, pre):
    """One power iteration on .data, differentiable sigma (spectral.py:23-35). Mutates u, v in P."""
    w = P[pre + "weight_bar"]
    h = w.shape[0]
    wm = w.detach().reshape(h, -1)
    with torch.no_grad():
        v = l2normalize(torch.mv(wm.t(), P[pre + "weight_u"]))
        u = l2normalize(torch.mv(wm, v))
    # spectral.py:30-31 assigns through ``.data``: no autograd version bump, and graphs recorded by
    # EARLIER forwards see the new u, v when they are finally back-propagated (the D step runs three
    # forwards before one backward, kd_model.py:156-164).  Reproduced, not "fixed".
    P[pre + "weight_v"].data = v
    P[pre + "weight_u"].data = u
    sigma = P[pre + "weight_u"].dot(w.reshape(h, -1).mv(P[pre + "weight_v"]))
    return w / sigma


def self_attn(P, name, x):
    B, C, W, H = x.shape
    q = F.conv2d(x, P[name + ".query_conv.weight"], P[name + ".query_conv.bias"]).view(B, -1, W * H).permute(0, 2, 1)
    k = F.conv2d(x, P[name + ".key_conv.weight"], P[name + ".key_conv.bias"]).view(B, -1, W * H)
    attn = F.softmax(torch.bmm(q, k), dim=-1)
    v = F.conv2d(x, P[name + ".value_conv.weight"], P[name + ".value_conv.bias"]).view(B, -1, W * H)
    out = torch.bmm(v, attn.permute(0, 2, 1)).view(B, C, W, H)
    return P[name + ".gamma"] * out + x, attn


class LeakyMasks:
    """The critic's LeakyReLU decisions of ANOTHER evaluation of the same inputs, prescribed to this one (kink-aware parity).

    The critic is piecewise smooth: its LeakyReLU slopes (sagan_models.py:116-134) are step functions of the pre-activations, the
    WGAN-GP term differentiates THROUGH them, and two evaluations of the same logits in different arithmetic (fp32 on the GPU, fp64
    here) disagree about the sign of a pre-activation that lies within rounding of zero -- one flipped first-layer unit moves the
    gradient by ~ 1 / sqrt(units).  No bound on a gradient holds ACROSS a flipped unit, so the comparison is made on the same linear
    piece: ``masks`` are the (pre-activation > 0) decisions the evaluation under test actually took, in call order (4 per critic
    forward); ``leaky`` applies them instead of this evaluation's own signs and records every disagreement with the distance of
    that pre-activation from the kink (relative to the layer's rms) -- the test asserts that only units within rounding distance
    of zero were overridden, and only a handful of them."""

    def __init__(self, masks):
        self.masks = [m.bool() for m in masks]
        self.pos = 0
        self.units = 0
        self.flipped = 0              # units whose prescribed decision differs from this evaluation's own sign
        self.worst = 0.0              # max over those of |pre-activation| / rms(layer pre-activations)

    def leaky(self, pre, slope):
        assert self.pos < len(self.masks), "more LeakyReLU calls than recorded masks"
        m = self.masks[self.pos].to(pre.device)
        self.pos += 1
        assert m.shape == pre.shape, (tuple(m.shape), tuple(pre.shape))
        d = pre.detach()
        diff = m != (d > 0)
        self.units += m.numel()
        n = int(diff.sum())
        if n:
            self.flipped += n
            self.worst = max(self.worst, float(d[diff].abs().max() / d.pow(2).mean().sqrt()))
        return pre * torch.where(m, torch.ones((), dtype=pre.dtype), torch.full((), slope, dtype=pre.dtype))

    def done(self):
        assert self.pos == len(self.masks), "recorded %d LeakyReLU masks, %d consumed" % (len(self.masks), self.pos)
        return self


def discriminator_forward(P, x, training=True, masks=None):
    """[out (B,1,1,1), attn1 (B,64,64), attn2 (B,16,16)] for 65x65 inputs, preprocess mode 1 (BN).
    ``masks``: a LeakyMasks whose recorded decisions replace this evaluation's own LeakyReLU signs (see the class)."""
    leaky = (lambda t: F.leaky_relu(t, 0.1)) if masks is None else (lambda t: masks.leaky(t, 0.1))
    pa = "preprocess_additional."
    if training:
        P[pa + "num_batches_tracked"] += 1
    x = F.batch_norm(x, P[pa + "running_mean"], P[pa + "running_var"], P[pa + "weight"], P[pa + "bias"],
                     training, 0.1, 1e-5)                                # sagan_models.py:148,158
    out = x
    for i in (1, 2, 3):
        pre = "l%d.0.module." % i
        out = leaky(F.conv2d(out, spectral_weight(P, pre), P[pre + "bias"], 2, 1))
    out, p1 = self_attn(P, "attn1", out)
    pre = "l4.0.module."
    out = leaky(F.conv2d(out, spectral_weight(P, pre), P[pre + "bias"], 2, 1))
    out, p2 = self_attn(P, "attn2", out)
    out = F.conv2d(out, P["last.0.weight"], P["last.0.bias"])
    return [out, p1, p2]


# --------------------------------------------------------------------------------------------
# criteria
# --------------------------------------------------------------------------------------------
def criterion_dsn(preds, target, ignore_index=255):
    h, w = target.shape[1], target.shape[2]
    l1 = F.cross_entropy(F.interpolate(preds[0], size=(h, w), mode="bilinear", align_corners=True), target,
                         ignore_index=ignore_index)
    l2 = F.cross_entropy(F.interpolate(preds[1], size=(h, w), mode="bilinear", align_corners=True), target,
                         ignore_index=ignore_index)
    return l1 + l2 * 0.4


def criterion_pixel_wise(preds_S, preds_T):
    assert preds_S[0].shape == preds_T[0].shape, "the output dim of teacher and student differ"
    N, C, W, H = preds_S[0].shape
    pt = F.softmax(preds_T[0].permute(0, 2, 3, 1).reshape(-1, C), dim=1)
    ls = F.log_softmax(preds_S[0].permute(0, 2, 3, 1).reshape(-1, C), dim=1)
    return torch.sum(-pt * ls) / W / H                                   # NOT divided by N


def similarity(feat):
    norm = (((feat ** 2).sum(dim=1)) ** 0.5).reshape(feat.shape[0], 1, feat.shape[2], feat.shape[3]) + 1e-8
    feat = feat / norm.detach()                                          # utils.py:175: detached
    feat = feat.reshape(feat.shape[0], feat.shape[1], -1)
    return torch.einsum("icm,icn->imn", feat, feat)


def pair_wise_pool_window(shape, scale):
    """criterion.py:241-242: (int(H*scale), int(W*scale)) on feat.shape[2], feat.shape[3]."""
    return int(shape[2] * scale), int(shape[3] * scale)


def criterion_pair_wise(preds_S, preds_T, scale=0.5, feat_ind=-5):
    fs, ft = preds_S[feat_ind], preds_T[feat_ind]
    kh, kw = pair_wise_pool_window(ft.shape, scale)
    ps = F.max_pool2d(fs, (kh, kw), (kh, kw), 0, ceil_mode=True)
    pt = F.max_pool2d(ft, (kh, kw), (kh, kw), 0, ceil_mode=True)
    err = ((similarity(pt) - similarity(ps)) ** 2) / ((pt.shape[-1] * pt.shape[-2]) ** 2) / pt.shape[0]
    return err.sum()


def _check_adv(adv_type):
    if adv_type not in ("wgan-gp", "hinge"):
        raise ValueError("adv_type should be wgan-gp or hinge")


def criterion_adv_for_g(d_out_S, adv_type="wgan-gp"):
    _check_adv(adv_type)
    return -d_out_S[0].mean()


def criterion_adv(d_out_S, d_out_T, adv_type="wgan-gp"):
    _check_adv(adv_type)
    assert d_out_S[0].shape == d_out_T[0].shape
    if adv_type == "wgan-gp":
        return -d_out_T[0].mean() + d_out_S[0].mean()
    return F.relu(1.0 - d_out_T[0]).mean() + F.relu(1.0 + d_out_S[0]).mean()


def criterion_gp(PD, preds_S, preds_T, lambda_gp, alpha, masks=None):
    """alpha: (B,1,1,1) uniform samples (the reference draws them with torch.rand, criterion.py:104)."""
    real, fake = preds_T[0].detach(), preds_S[0].detach()
    assert real.shape == fake.shape
    x = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    out = discriminator_forward(PD, x, masks=masks)
    grad = torch.autograd.grad(out[0], x, torch.ones_like(out[0]), retain_graph=True, create_graph=True)[0]
    grad = grad.reshape(grad.shape[0], -1)
    return lambda_gp * torch.mean((torch.sqrt(torch.sum(grad ** 2, dim=1)) - 1) ** 2)


# --------------------------------------------------------------------------------------------
# the step
# --------------------------------------------------------------------------------------------
class StepConfig:
    """Subset of utils/train_options.py:18-63 that shapes the step."""

    def __init__(self, pi=True, pa=True, ho=True, lambda_pi=10.0, lambda_pa=1.0, lambda_d=0.1, lambda_gp=10.0,
                 pool_scale=0.5, adv_loss_type="wgan-gp", lr_g=1e-2, lr_d=4e-4, momentum=0.9, weight_decay=1e-4,
                 dropout_p=0.1):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def sgd_step(P, grads, bufs, lr, momentum, weight_decay):
    """torch.optim.SGD semantics (kd_model.py:74-75): d = g + wd*p; buf = mu*buf + d (first step: buf = d); p -= lr*buf."""
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            d = g + weight_decay * P[k] if weight_decay != 0 else g.clone()
            if k in bufs:
                bufs[k].mul_(momentum).add_(d)
            else:
                bufs[k] = d.clone()
            P[k].sub_(lr * bufs[k])


def _zero_like_grads(P):
    return {k: None for k in learnable_keys(P)}


def distillation_step(PS, PT, PD, images, labels, cfg, state=None, alpha=None, dropout_masks=None,
                      lr_g=None, lr_d=None, abn_kw=None, apply_updates=True):
    """One NetModel.optimize_parameters() (kd_model.py:167-173).  Returns a dict with the logged
    scalars (kd_model.py:183-187), the student / D gradients, and the tensors the step produced."""
    state = state if state is not None else {"G": {}, "D": {}}
    lr_g = cfg.lr_g if lr_g is None else lr_g
    lr_d = cfg.lr_d if lr_d is None else lr_d
    require_grad(PS, True)
    if PD is not None:
        require_grad(PD, True)
    with torch.no_grad():                                                # kd_model.py:121-122
        preds_T = pspnet_forward(PT, images, TEACHER, False)
    preds_S = pspnet_forward(PS, images, STUDENT, True, cfg.dropout_p, dropout_masks, abn_kw)
    out = {}
    mc = criterion_dsn(preds_S, labels)
    out["mc_G_loss"] = float(mc)
    G = mc
    out["pi_G_loss"] = out["pa_G_loss"] = 0.0
    if cfg.pi:
        pi = cfg.lambda_pi * criterion_pixel_wise(preds_S, preds_T)
        out["pi_G_loss"] = float(pi)
        G = G + pi
    if cfg.pa:
        pa = criterion_pair_wise(preds_S, preds_T, cfg.pool_scale, -5)
        out["pa_G_loss"] = float(pa)
        G = G + cfg.lambda_pa * pa
    if cfg.ho:
        d_out_S = discriminator_forward(PD, preds_S[0])                  # kd_model.py:148
        G = G + cfg.lambda_d * criterion_adv_for_g(d_out_S, cfg.adv_loss_type)
    s_keys = learnable_keys(PS)
    g = torch.autograd.grad(G, [PS[k] for k in s_keys], allow_unused=True)
    out["G_loss"] = float(G)
    out["grads_S"] = dict(zip(s_keys, g))
    out["preds_S"] = [t.detach() for t in preds_S]
    out["preds_T"] = preds_T
    if apply_updates:
        sgd_step(PS, out["grads_S"], state["G"], lr_g, cfg.momentum, cfg.weight_decay)
    out["D_loss"] = 0.0
    if cfg.ho:                                                           # kd_model.py:153-165
        d_out_T = discriminator_forward(PD, preds_T[0].detach())
        d_out_S = discriminator_forward(PD, preds_S[0].detach())
        d_loss = cfg.lambda_d * criterion_adv(d_out_S, d_out_T, cfg.adv_loss_type)
        if cfg.adv_loss_type == "wgan-gp":
            if alpha is None:
                alpha = torch.rand(images.shape[0], 1, 1, 1, dtype=images.dtype)
            d_loss = d_loss + cfg.lambda_d * criterion_gp(PD, preds_S, preds_T, cfg.lambda_gp, alpha)
        d_keys = learnable_keys(PD)
        gd = torch.autograd.grad(d_loss, [PD[k] for k in d_keys], allow_unused=True)
        out["D_loss"] = float(d_loss)
        out["grads_D"] = dict(zip(d_keys, gd))
        if apply_updates:
            sgd_step(PD, out["grads_D"], state["D"], lr_d, cfg.momentum, cfg.weight_decay)
    require_grad(PS, False)
    if PD is not None:
        require_grad(PD, False)
    return out


def distillation_step_sharded(PS, PT, PD, images, labels, cfg, shards, alphas=None, lr_g=None, lr_d=None,
                              apply_updates=True):
    """BASELINE configs[3]: the step of ``distillation_step`` as the reference's multi-GPU mode defines it.

    ``nn.DataParallel`` scatters the minibatch (utils/parallel.py:106,142); every replica computes ITS shard's
    criteria and ``Reduce.apply(*outputs) / len(outputs)`` averages them (parallel.py:155), so the quantity that
    is differentiated is the MEAN over shards of the per-shard losses.  What crosses shards:
      * the student's InPlaceABNSync statistics (libs/functions.py:185-209): whole-batch mean / variance, which
        for equal shards is exactly batch norm over the concatenated batch -- so the student runs ONCE here on
        the whole batch and its outputs are sliced;
      * the gradient sum (ReduceAddCoalesced of the replicas), i.e. autograd of the mean loss.
    What does NOT: the discriminator's ``nn.BatchNorm2d`` (sagan_models.py:148) sees only its replica's shard,
    Pi is a per-shard sum (criterion.py:225), Pa divides by the shard's batch (utils.py:181), CE averages over
    the shard's valid pixels.  Spectral-norm u, v advance once per D forward of a replica (spectral.py:30-31):
    they depend on the weights only, so every replica holds the same values -- each shard gets its own copy of
    ``PD`` here and the copies' u, v must come out identical (checked by the caller).

    ``shards``: list of slices of the batch dimension; ``alphas``: per-shard (b,1,1,1) WGAN-GP coefficients.
    Returns per-shard scalars, the averaged student / D gradients, and (apply_updates) updates PS / PD in place
    with them; ``PD_shards`` are the per-replica D states after the step (BN running statistics differ)."""
    lr_g = cfg.lr_g if lr_g is None else lr_g
    lr_d = cfg.lr_d if lr_d is None else lr_d
    G = len(shards)
    require_grad(PS, True)
    with torch.no_grad():
        preds_T = pspnet_forward(PT, images, TEACHER, False)
    preds_S = pspnet_forward(PS, images, STUDENT, True, cfg.dropout_p)
    PDs = []
    if cfg.ho:
        for _ in range(G):
            P = {k: v.detach().clone() for k, v in PD.items()}
            PDs.append(require_grad(P, True))
    total = 0.0
    out = {"shards": []}
    for r, sl in enumerate(shards):
        s, t = [p[sl] for p in preds_S], [p[sl] for p in preds_T]
        rec = {}
        mc = criterion_dsn(s, labels[sl])
        g_loss = mc
        rec["mc_G_loss"] = float(mc.detach())
        rec["pi_G_loss"] = rec["pa_G_loss"] = 0.0
        if cfg.pi:
            pi = cfg.lambda_pi * criterion_pixel_wise(s, t)
            rec["pi_G_loss"] = float(pi.detach())
            g_loss = g_loss + pi
        if cfg.pa:
            pa = criterion_pair_wise(s, t, cfg.pool_scale, -5)
            rec["pa_G_loss"] = float(pa.detach())
            g_loss = g_loss + cfg.lambda_pa * pa
        if cfg.ho:
            g_loss = g_loss + cfg.lambda_d * criterion_adv_for_g(discriminator_forward(PDs[r], s[0]), cfg.adv_loss_type)
        rec["G_loss"] = float(g_loss.detach())
        out["shards"].append(rec)
        total = total + g_loss / G
    s_keys = learnable_keys(PS)
    out["grads_S"] = dict(zip(s_keys, torch.autograd.grad(total, [PS[k] for k in s_keys], allow_unused=True)))
    out["preds_S"] = [t.detach() for t in preds_S]
    out["preds_T"] = preds_T
    state = {"G": {}, "D": {}}
    if apply_updates:
        sgd_step(PS, out["grads_S"], state["G"], lr_g, cfg.momentum, cfg.weight_decay)
    if cfg.ho:
        d_keys = learnable_keys(PD)
        acc = {k: None for k in d_keys}
        for r, sl in enumerate(shards):
            P = PDs[r]
            pS, pT = preds_S[0][sl].detach(), preds_T[0][sl].detach()
            d_t = discriminator_forward(P, pT)
            d_s = discriminator_forward(P, pS)
            d_loss = cfg.lambda_d * criterion_adv(d_s, d_t, cfg.adv_loss_type)
            if cfg.adv_loss_type == "wgan-gp":
                d_loss = d_loss + cfg.lambda_d * criterion_gp(P, [pS], [pT], cfg.lambda_gp, alphas[r])
            out["shards"][r]["D_loss"] = float(d_loss.detach())
            for k, g in zip(d_keys, torch.autograd.grad(d_loss, [P[k] for k in d_keys], allow_unused=True)):
                if g is not None:
                    acc[k] = g / G if acc[k] is None else acc[k] + g / G
            require_grad(P, False)
        out["grads_D"] = acc
        out["PD_shards"] = PDs
        if apply_updates:
            require_grad(PD, True)
            sgd_step(PD, acc, state["D"], lr_d, cfg.momentum, cfg.weight_decay)
            require_grad(PD, False)
            with torch.no_grad():
                for k in PD:
                    if k.endswith(("weight_u", "weight_v")):
                        PD[k].copy_(PDs[0][k])
    require_grad(PS, False)
    return out


def discriminator_step(P, logits_S, logits_T, cfg, alpha, g_step_forward=True, masks=None, terms=None):
    """The discriminator's part of one step on GIVEN logits (kd_model.py:148, 153-165): the G step's critic forward
    (advances u, v only), D(T), D(S), adversarial loss + WGAN-GP.  Mutates u, v / BN statistics in ``P``.
    Returns (d_loss as float, {key: gradient}).  ``masks``: LeakyMasks over the step's critic forwards in this order
    (kink-aware comparison, see the class); ``terms``: a dict that receives the magnitudes of the cancelling summands."""
    require_grad(P, True)
    pS, pT = logits_S.detach(), logits_T.detach()
    if g_step_forward:
        discriminator_forward(P, pS, masks=masks)
    d_t, d_s = discriminator_forward(P, pT, masks=masks), discriminator_forward(P, pS, masks=masks)
    d_loss = cfg.lambda_d * criterion_adv(d_s, d_t, cfg.adv_loss_type)
    gp = None
    if cfg.adv_loss_type == "wgan-gp":
        gp = cfg.lambda_d * criterion_gp(P, [pS], [pT], cfg.lambda_gp, alpha, masks=masks)
        d_loss = d_loss + gp
    if terms is not None:
        terms.update(d_T=float(cfg.lambda_d * d_t[0].detach().mean().abs()), d_S=float(cfg.lambda_d * d_s[0].detach().mean().abs()),
                     gp=float(gp.detach().abs()) if gp is not None else 0.0)
    keys = learnable_keys(P)
    grads = dict(zip(keys, torch.autograd.grad(d_loss, [P[k] for k in keys], allow_unused=True)))
    require_grad(P, False)
    if masks is not None:
        masks.done()
    return float(d_loss.detach()), grads


def synthetic_batch(B, H, W, num_classes=19, seed=0, dtype=torch.float32):
    """SURVEY.md 8d synthetic inputs: images randn*57, labels randint with a 255 stripe in sample 0."""
    gen = torch.Generator().manual_seed(seed)
    images = (torch.randn(B, 3, H, W, generator=gen) * 57.0).to(dtype)
    labels = torch.randint(0, num_classes, (B, H, W), generator=gen)
    labels[0, : max(1, H // 16)] = 255
    return images, labels
