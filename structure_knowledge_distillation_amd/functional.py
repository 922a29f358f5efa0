"""Autograd front-ends of the loss / spectral-norm kernels in libskd_hip.so.

Each function here is the differentiable form of one reference computation, executed by the
hand-written gfx950 kernels behind the C ABI of include/skd.h (no eager fallback: CPU tensors
raise, see _lib.require_device):

    cross_entropy_dsn(m, d, y)   utils/criterion.py:179-188   csrc/ce_dsn.hip
    ppm_pool / ppm_concat        networks/pspnet_combine.py:102-111   csrc/ppm.hip
    seg_confusion(logits, y)     networks/evaluate.py:106-113,186-198 csrc/evaluate.hip  (no autograd)
    pixel_wise_loss(S, T)        utils/criterion.py:219-226   csrc/pixelwise.hip
    max_pool_argmax(x, kh, kw)   nn.MaxPool2d(k=s, ceil_mode=True) of criterion.py:243   csrc/pairwise.hip
    sim_dis(f_S, f_T)            utils/utils.py:170-183 (L2, similarity, sim_dis_compute)  csrc/pairwise.hip
    spectral_normalize(W, u, v)  networks/spectral.py:23-35    csrc/spectral.hip

All are first-order only (``once_differentiable``), like the reference's native op
(libs/functions.py:112,230).  WGAN-GP's double backward (criterion.py:110-115) only needs second
derivatives of the discriminator's stock convolutions; the spectral-norm node is traversed once,
by the final ``d_loss.backward()``.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib


def _f32c(t, what):
    if t.dtype != torch.float32:
        raise TypeError("%s: fp32 tensors only (got %s)" % (what, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


class _PixelWise(Function):
    @staticmethod
    def forward(ctx, logits_s, logits_t):
        _lib.require_device(logits_s, logits_t)
        assert logits_s.shape == logits_t.shape, "the output dim of teacher and student differ"  # criterion.py:221
        s, t = _f32c(logits_s, "pixel_wise_loss"), _f32c(logits_t, "pixel_wise_loss")
        n, c = s.shape[0], s.shape[1]
        hw = s.numel() // max(1, n * c)
        lib, st = _lib.get(), _lib.stream_of(s)
        loss = s.new_empty(())
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(s) if need_grad else None
        ws = s.new_empty((max(1, lib.skd_pixelwise_workspace_floats(n, hw)),))
        _lib.check(lib.skd_pixelwise_loss(n, c, hw, s.data_ptr(), t.data_ptr(), loss.data_ptr(),
                                          _lib.ptr(grad), ws.data_ptr(), st), "skd_pixelwise_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad is not None else None), None


def pixel_wise_loss(logits_s, logits_t):
    """sum(-softmax(T) * log_softmax(S)) / W / H over the class dim of (N,C,W,H) logits; no grad to T."""
    return _PixelWise.apply(logits_s, logits_t.detach())


class _CrossEntropyDSN(Function):
    @staticmethod
    def forward(ctx, logits_main, logits_dsn, target, ignore_index, aux_weight):
        _lib.require_device(logits_main, logits_dsn, target)
        lm = _f32c(logits_main, "cross_entropy_dsn")
        ld = _f32c(logits_dsn, "cross_entropy_dsn") if logits_dsn is not None else None
        if target.dtype != torch.int64:
            raise TypeError("cross_entropy_dsn: int64 target expected (got %s)" % target.dtype)
        tg = target if target.is_contiguous() else target.contiguous()
        b, c, h, w = lm.shape
        if ld is not None and ld.shape != lm.shape:
            raise ValueError("main and dsn logits differ in shape")
        if tg.dim() != 3 or tg.shape[0] != b:
            raise ValueError("target must be (B, H, W)")
        H, W = tg.shape[1], tg.shape[2]
        lib, st = _lib.get(), _lib.stream_of(lm)
        need_m = ctx.needs_input_grad[0]
        need_d = ld is not None and ctx.needs_input_grad[1]
        loss = lm.new_empty(())
        gm = torch.empty_like(lm) if need_m else None
        gd = torch.empty_like(ld) if need_d else None
        ws = lm.new_empty((max(8, lib.skd_ce_dsn_workspace_floats(b, c, h, w, H, W)),))
        _lib.check(lib.skd_ce_dsn_forward(b, c, h, w, H, W, lm.data_ptr(), _lib.ptr(ld), tg.data_ptr(),
                                          int(ignore_index), float(aux_weight), loss.data_ptr(), _lib.ptr(gm),
                                          _lib.ptr(gd), ws.data_ptr(), st), "skd_ce_dsn_forward")
        ctx.save_for_backward(gm, gd)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        gm, gd = ctx.saved_tensors
        return (gm * g if gm is not None else None), (gd * g if gd is not None else None), None, None, None


def cross_entropy_dsn(logits_main, logits_dsn, target, ignore_index=255, aux_weight=0.4):
    """CE(up(main), target) + aux_weight * CE(up(dsn), target), up = bilinear align_corners upsample to the
    target's size, CE = mean over non-ignored pixels (utils/criterion.py:179-188) -- one fused kernel chain
    that never materialises a (B, C, H, W) tensor.  ``logits_dsn`` may be None (single CE)."""
    return _CrossEntropyDSN.apply(logits_main, logits_dsn, target, ignore_index, aux_weight)


def _is_cl(t):
    """4-D, channels-last memory (and not also plain-contiguous), channel count a multiple of 4."""
    return (t.dim() == 4 and t.shape[1] % 4 == 0 and not t.is_contiguous()
            and t.is_contiguous(memory_format=torch.channels_last))


def _cl(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _new_cl(ref, b, c, h, w):
    """Uninitialised (b, c, h, w) tensor with channels-last memory (b, h, w, c)."""
    return ref.new_empty((b, h, w, c), dtype=torch.float32).permute(0, 3, 1, 2)


class _PPMPool(Function):
    """All pyramid levels of AdaptiveAvgPool2d from one read of the feature map (NCHW or channels-last)."""

    @staticmethod
    def forward(ctx, x, sizes):
        _lib.require_device(x)
        if x.dtype != torch.float32:
            raise TypeError("ppm_pool: fp32 tensors only (got %s)" % x.dtype)
        ctx.cl = _is_cl(x)
        if ctx.cl:
            b, c, h, w = x.shape
            lib, st = _lib.get(), _lib.stream_of(x)
            arr = _lib.int_array(sizes)
            total = lib.skd_ppm_pooled_floats(b * c, len(sizes), arr)
            if total <= 0:
                raise ValueError("ppm_pool: bad pyramid sizes %r" % (sizes,))
            buf = x.new_empty((total,))
            ws = x.new_empty((max(1, lib.skd_ppm_nhwc_workspace_floats(b, c, 0, h, w, len(sizes), arr)),))
            _lib.check(lib.skd_ppm_pool_nhwc(b, c, h, w, len(sizes), arr, x.data_ptr(), buf.data_ptr(), ws.data_ptr(), st),
                       "skd_ppm_pool_nhwc")
            ctx.geom = (b, c, h, w, tuple(sizes))
            outs, off = [], 0
            for s_ in sizes:
                n = b * c * s_ * s_
                outs.append(buf[off:off + n].view(b, s_, s_, c).permute(0, 3, 1, 2))     # (b, c, s, s), channels-last memory
                off += n
            return tuple(outs)
        x = _f32c(x, "ppm_pool")
        b, c, h, w = x.shape
        lib, st = _lib.get(), _lib.stream_of(x)
        arr = _lib.int_array(sizes)
        total = lib.skd_ppm_pooled_floats(b * c, len(sizes), arr)
        if total <= 0:
            raise ValueError("ppm_pool: bad pyramid sizes %r" % (sizes,))
        buf = x.new_empty((total,))
        _lib.check(lib.skd_ppm_pool(b * c, h, w, len(sizes), arr, x.data_ptr(), buf.data_ptr(), st), "skd_ppm_pool")
        ctx.geom = (b, c, h, w, tuple(sizes))
        outs, off = [], 0
        for s_ in sizes:
            n = b * c * s_ * s_
            outs.append(buf[off:off + n].view(b, c, s_, s_))
            off += n
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        b, c, h, w, sizes = ctx.geom
        ref = next(g for g in grads if g is not None)
        if ctx.cl:
            flat = torch.cat([(_cl(g).permute(0, 2, 3, 1) if g is not None else ref.new_zeros((b, s_, s_, c))).reshape(-1)
                              for g, s_ in zip(grads, sizes)]).to(torch.float32)
            dx = _new_cl(ref, b, c, h, w)
            lib, st = _lib.get(), _lib.stream_of(flat)
            _lib.check(lib.skd_ppm_pool_backward_nhwc(b, c, h, w, len(sizes), _lib.int_array(sizes), flat.data_ptr(),
                                                      dx.data_ptr(), st), "skd_ppm_pool_backward_nhwc")
            return dx, None
        flat = torch.cat([(g if g is not None else ref.new_zeros((b, c, s_, s_))).reshape(-1)
                          for g, s_ in zip(grads, sizes)]).to(torch.float32)
        dx = ref.new_empty((b, c, h, w), dtype=torch.float32)
        lib, st = _lib.get(), _lib.stream_of(flat)
        _lib.check(lib.skd_ppm_pool_backward(b * c, h, w, len(sizes), _lib.int_array(sizes), flat.data_ptr(),
                                             dx.data_ptr(), st), "skd_ppm_pool_backward")
        return dx, None


def ppm_pool(x, sizes=(1, 2, 3, 6)):
    """[AdaptiveAvgPool2d(s)(x) for s in sizes] (pspnet_combine.py:102) -- one kernel, one read of x."""
    return _PPMPool.apply(x, tuple(int(s_) for s_ in sizes))


class _PPMConcat(Function):
    """cat([upsample(p, (H, W), bilinear, align_corners=True) for p in priors] + [feats], 1)."""

    @staticmethod
    def forward(ctx, feats, *priors):
        _lib.require_device(feats, *priors)
        ctx.cl = _is_cl(feats) and feats.dtype == torch.float32 and all(p.dtype == torch.float32 and p.shape[1] % 4 == 0 for p in priors)
        if ctx.cl:
            b, cf, h, w = feats.shape
            cout = priors[0].shape[1]
            sizes = []
            for p in priors:
                if p.shape[0] != b or p.shape[1] != cout or p.shape[2] != p.shape[3]:
                    raise ValueError("ppm_concat: priors must be (B, Cout, s, s)")
                sizes.append(p.shape[2])
            priors = [_cl(p) for p in priors]                     # memory (b, s, s, cout)
            lib, st = _lib.get(), _lib.stream_of(feats)
            cat = _new_cl(feats, b, len(priors) * cout + cf, h, w)
            _lib.check(lib.skd_ppm_concat_nhwc(b, cout, cf, h, w, len(sizes), _lib.int_array(sizes), _lib.ptr_array(priors),
                                               feats.data_ptr(), cat.data_ptr(), st), "skd_ppm_concat_nhwc")
            ctx.geom = (b, cout, cf, h, w, tuple(sizes))
            return cat
        feats = _f32c(feats, "ppm_concat")
        priors = [_f32c(p, "ppm_concat") for p in priors]
        b, cf, h, w = feats.shape
        cout = priors[0].shape[1]
        sizes = []
        for p in priors:
            if p.shape[0] != b or p.shape[1] != cout or p.shape[2] != p.shape[3]:
                raise ValueError("ppm_concat: priors must be (B, Cout, s, s)")
            sizes.append(p.shape[2])
        lib, st = _lib.get(), _lib.stream_of(feats)
        cat = feats.new_empty((b, len(priors) * cout + cf, h, w))
        _lib.check(lib.skd_ppm_concat(b, cout, cf, h, w, len(sizes), _lib.int_array(sizes), _lib.ptr_array(priors),
                                      feats.data_ptr(), cat.data_ptr(), st), "skd_ppm_concat")
        ctx.geom = (b, cout, cf, h, w, tuple(sizes))
        return cat

    @staticmethod
    @once_differentiable
    def backward(ctx, gcat):
        b, cout, cf, h, w, sizes = ctx.geom
        if ctx.cl:
            gcat = _cl(gcat.to(torch.float32))
            lib, st = _lib.get(), _lib.stream_of(gcat)
            arr = _lib.int_array(sizes)
            need_p = any(ctx.needs_input_grad[1:])
            gfeats = _new_cl(gcat, b, cf, h, w) if ctx.needs_input_grad[0] else None
            gpriors = [_new_cl(gcat, b, cout, s_, s_) for s_ in sizes] if need_p else None
            ws = gcat.new_empty((max(1, lib.skd_ppm_nhwc_workspace_floats(b, 0, cout, h, w, len(sizes), arr)),)) if need_p else None
            _lib.check(lib.skd_ppm_concat_backward_nhwc(b, cout, cf, h, w, len(sizes), arr, gcat.data_ptr(),
                                                        _lib.ptr_array(gpriors) if need_p else None, _lib.ptr(gfeats),
                                                        _lib.ptr(ws), st), "skd_ppm_concat_backward_nhwc")
            return (gfeats,) + (tuple(gpriors) if need_p else (None,) * len(sizes))
        gcat = _f32c(gcat, "ppm_concat backward")
        lib, st = _lib.get(), _lib.stream_of(gcat)
        gfeats = gcat[:, len(sizes) * cout:] if ctx.needs_input_grad[0] else None
        gpriors = [None] * len(sizes)
        if any(ctx.needs_input_grad[1:]):
            gpriors = [gcat.new_empty((b, cout, s_, s_)) for s_ in sizes]
            _lib.check(lib.skd_ppm_concat_backward(b, cout, cf, h, w, len(sizes), _lib.int_array(sizes),
                                                   gcat.data_ptr(), _lib.ptr_array(gpriors), st),
                       "skd_ppm_concat_backward")
        return (gfeats,) + tuple(gpriors)


def ppm_concat(priors, feats):
    """The concatenated input of the PSP bottleneck (pspnet_combine.py:110-111): up-sampled priors followed by
    the feature map, written directly into one (B, L*Cout + Cfeat, H, W) tensor."""
    return _PPMConcat.apply(feats, *priors)


def _fold_block_ptrs(z_all, b, cout, sizes):
    """Device addresses of the diagonal blocks of Z_all: rows of level k start at B * sum_{j<k} s_j^2, its columns at
    k * 9 * Cout."""
    ptrs, row = [], 0
    ld = z_all.shape[1]
    for k, s_ in enumerate(sizes):
        ptrs.append(z_all.data_ptr() + 4 * (row * ld + k * 9 * cout))
        row += b * s_ * s_
    return (ctypes.c_void_p * len(ptrs))(*ptrs), ld


class _PPMFold(Function):
    """base += fold(Z_1 .. Z_L) (include/skd.h section 8), in place on the channels-last convolution output `base`.
    z_all: (B * sum_k s_k^2, L * 9 * Cout), the product of the stacked priors with the stacked weight blocks; level k's Z
    is its k-th diagonal block (the off-diagonal blocks are never read, and receive zero gradient)."""

    @staticmethod
    def forward(ctx, base, sizes, z_all):
        _lib.require_device(base, z_all)
        b, cout, h, w = base.shape
        if cout % 4 or not base.is_contiguous(memory_format=torch.channels_last) or base.dtype != torch.float32:
            raise ValueError("ppm_fold: base must be a float32 channels-last (B, Cout, H, W) tensor")
        rows = b * sum(s_ * s_ for s_ in sizes)
        if z_all.dtype != torch.float32 or not z_all.is_contiguous() or tuple(z_all.shape) != (rows, len(sizes) * 9 * cout):
            raise ValueError("ppm_fold: Z must be a contiguous float32 (B * sum s^2, L * 9 * Cout) tensor")
        lib, st = _lib.get(), _lib.stream_of(base)
        ptrs, ld = _fold_block_ptrs(z_all, b, cout, sizes)
        _lib.check(lib.skd_ppm_fold_nhwc(b, cout, h, w, len(sizes), _lib.int_array(sizes), ptrs, ld, base.data_ptr(), st),
                   "skd_ppm_fold_nhwc")
        ctx.geom = (b, cout, h, w, tuple(sizes))
        ctx.zshape = z_all.shape
        ctx.mark_dirty(base)
        return base

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        b, cout, h, w, sizes = ctx.geom
        gout = _cl(gout.to(torch.float32))
        gz = None
        if ctx.needs_input_grad[2]:
            lib, st = _lib.get(), _lib.stream_of(gout)
            arr = _lib.int_array(sizes)
            gz = gout.new_zeros(ctx.zshape)
            ptrs, ld = _fold_block_ptrs(gz, b, cout, sizes)
            ws = gout.new_empty((max(1, lib.skd_ppm_fold_nhwc_workspace_floats(b, cout, h, w, len(sizes), arr)),))
            _lib.check(lib.skd_ppm_fold_backward_nhwc(b, cout, h, w, len(sizes), arr, gout.data_ptr(), ptrs, ld,
                                                      ws.data_ptr(), st), "skd_ppm_fold_backward_nhwc")
        return gout if ctx.needs_input_grad[0] else None, None, gz


def ppm_fold_supported(feats, sizes):
    """True when the folded evaluation of the PSP bottleneck (ppm_fold_bottleneck) takes this feature map."""
    return (feats.is_cuda or _lib.test_backend_active()) and feats.dtype == torch.float32 and feats.dim() == 4 \
        and _is_cl(feats) and 3 * sum(sizes) <= 64 and len(sizes) <= 4


def ppm_fold_bottleneck(priors, feats, weight, cache=None):
    """conv3x3(cat([upsample(p) for p in priors] + [feats], 1), weight, padding=1) (pspnet_combine.py:104-111) without
    the concatenated tensor and without convolving the priors' channels: the feature-map slice of the weight goes through
    the convolution, the priors through a (B s^2) x Cm x 9 Cout GEMM each and the fold kernel (csrc/ppm.hip).
    priors: (B, Cm, s, s) channels-last; feats: (B, Cf, H, W) channels-last; weight: (Cout, L*Cm + Cf, 3, 3).
    `cache` (a dict) keeps the rearranged weight slices of a frozen network between calls."""
    import torch.nn.functional as F
    cm = priors[0].shape[1]
    n_prior = len(priors) * cm
    cout = weight.shape[0]
    sizes = tuple(int(p.shape[2]) for p in priors)
    key = (weight.data_ptr(), weight._version, tuple(weight.shape))
    mats = cache.get("mats") if cache is not None and cache.get("key") == key else None
    if mats is None:
        wf = weight[:, n_prior:].contiguous(memory_format=torch.channels_last)
        # (Cout, L, Cm, 3, 3) -> (Cm, L, 3, 3, Cout): column block k of the (Cm, L * 9 * Cout) matrix is level k's weights
        w_all = weight[:, :n_prior].reshape(cout, len(priors), cm, 3, 3).permute(2, 1, 3, 4, 0).reshape(cm, len(priors) * 9 * cout)
        mats = (wf, w_all)
        if cache is not None and not (torch.is_grad_enabled() and weight.requires_grad):
            cache["key"], cache["mats"] = key, mats
    wf, w_all = mats
    base = F.conv2d(feats, wf, None, 1, 1)
    # ONE GEMM of all levels' priors against all levels' weight blocks (only the diagonal blocks are used: 4x the
    # necessary flops of a 2 GFLOP product, but a single well-shaped library call instead of four skinny ones)
    p_all = torch.cat([_cl(p).permute(0, 2, 3, 1).reshape(-1, cm) for p in priors], 0)
    return _PPMFold.apply(base, sizes, torch.mm(p_all, w_all))


class _MaxPool3x3s2(Function):
    """MaxPool2d(3, 2, 1, ceil_mode) on a channels-last tensor (csrc/maxpool.hip): one byte of argmax per element."""

    @staticmethod
    def forward(ctx, x, oh, ow):
        _lib.require_device(x)
        b, c, h, w = x.shape
        lib, st = _lib.get(), _lib.stream_of(x)
        y = _new_cl(x, b, c, oh, ow)
        need = ctx.needs_input_grad[0]
        arg = torch.empty((b, oh, ow, c), dtype=torch.uint8, device=x.device) if need else None
        _lib.check(lib.skd_maxpool3x3s2_nhwc(b, c, h, w, oh, ow, x.data_ptr(), y.data_ptr(), _lib.ptr(arg), st), "skd_maxpool3x3s2_nhwc")
        ctx.geom = (b, c, h, w, oh, ow)
        if need:
            ctx.save_for_backward(arg)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        b, c, h, w, oh, ow = ctx.geom
        (arg,) = ctx.saved_tensors
        gy = _cl(gy.to(torch.float32))
        lib, st = _lib.get(), _lib.stream_of(gy)
        dx = _new_cl(gy, b, c, h, w)
        _lib.check(lib.skd_maxpool3x3s2_backward_nhwc(b, c, h, w, oh, ow, gy.data_ptr(), arg.data_ptr(), dx.data_ptr(), st),
                   "skd_maxpool3x3s2_backward_nhwc")
        return dx, None, None


def _pool_out(n, ceil_mode):
    o = (n + 2 - 3 + (1 if ceil_mode else 0)) // 2 + 1
    if ceil_mode and (o - 1) * 2 >= n + 1:
        o -= 1
    return o


stem_pool_out = _pool_out


def is_stem_pool(pool):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1[, ceil_mode]) without dilation / returned indices: what csrc/maxpool.hip and
    the fused stem kernels implement."""
    def _pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    return (isinstance(pool, torch.nn.MaxPool2d) and _pair(pool.kernel_size) == (3, 3) and _pair(pool.stride) == (2, 2)
            and _pair(pool.padding) == (1, 1) and _pair(pool.dilation) == (1, 1) and not pool.return_indices)


def max_pool_stem(x, pool):
    """``pool(x)`` for the stem's nn.MaxPool2d(3, 2, 1, ceil_mode=True) (pspnet_combine.py:135): channels-last fp32
    tensors take csrc/maxpool.hip, anything else the stock operator."""
    if (x.dtype == torch.float32 and _is_cl(x) and (x.is_cuda or _lib.test_backend_active()) and is_stem_pool(pool)):
        return _MaxPool3x3s2.apply(x, _pool_out(x.shape[2], pool.ceil_mode), _pool_out(x.shape[3], pool.ceil_mode))
    return pool(x)


class _Head1x1(Function):
    """The 19-class 1x1 classifier head on a channels-last feature map (csrc/head.hip): logits come out in the reference's NCHW layout
    (what the criteria and the critic consume: no layout copy), the feature gradient goes back channels-last (what the InPlace-ABN
    backward wants: no copy either), dW / db in a fixed summation order."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _lib.require_device(x, weight, bias)
        b, k, h, w = x.shape
        c = weight.shape[0]
        lib, st = _lib.get(), _lib.stream_of(x)
        wt = weight.reshape(c, k)
        wt = wt if wt.is_contiguous() else wt.contiguous()
        out = x.new_empty((b, c, h, w))
        _lib.check(lib.skd_head1x1_forward_nhwc(b, h * w, k, c, x.data_ptr(), wt.data_ptr(), _lib.ptr(bias), out.data_ptr(), st),
                   "skd_head1x1_forward_nhwc")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        b, k, h, w = x.shape
        c = weight.shape[0]
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        g = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.to(torch.float32).contiguous()
        lib, st = _lib.get(), _lib.stream_of(x)
        wt = weight.reshape(c, k)
        wt = wt if wt.is_contiguous() else wt.contiguous()
        gx = _new_cl(x, b, k, h, w) if need_x else None
        gw = torch.empty_like(wt) if need_w else None
        gb = x.new_empty((c,)) if need_b else None
        ws = x.new_empty((max(1, lib.skd_head1x1_backward_workspace_floats(b, h * w, k, c)),))
        _lib.check(lib.skd_head1x1_backward_nhwc(b, h * w, k, c, x.data_ptr(), wt.data_ptr(), g.data_ptr(), _lib.ptr(gx), _lib.ptr(gw),
                                                 _lib.ptr(gb), ws.data_ptr(), st), "skd_head1x1_backward_nhwc")
        return gx, (gw.view_as(weight) if gw is not None else None), gb


def head1x1_supported(x, conv):
    """True when csrc/head.hip takes this classifier: an fp32 channels-last input, a plain 1x1 / stride 1 / no padding / ungrouped
    convolution with <= 20 outputs, Cin a multiple of 128 (forward) and exactly 128 when a gradient is wanted."""
    if not (x.dtype == torch.float32 and x.dim() == 4 and _is_cl(x) and (x.is_cuda or _lib.test_backend_active())):
        return False
    if not (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and conv.weight.dtype == torch.float32):
        return False
    grad = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad
                                        or (conv.bias is not None and conv.bias.requires_grad))
    return bool(_lib.get().skd_head1x1_supported(conv.in_channels, conv.out_channels, 1 if grad else 0))


def head1x1(x, conv):
    """``conv(x)`` for a classifier head that head1x1_supported() accepted: (B, C, H, W) logits, NCHW-contiguous."""
    return _Head1x1.apply(x, conv.weight, conv.bias)


def blas_1x1_bn_supported(x, conv):
    """True when conv1x1_bn_blas takes this call: an fp32 channels-last input of a frozen network (no graph) and a plain
    stride-1 1x1 convolution without bias."""
    return (not torch.is_grad_enabled() and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None
            and conv.in_channels >= 128 and conv.out_channels >= 128)   # 256 -> 64 at 129 x 129: 97 us vs MIOpen's 64


def conv1x1_bn_blas(x, conv, bn, relu):
    """relu?(bn(conv1x1(x))) of a FROZEN network (eval-mode InPlace-ABN, pspnet_combine.py:65-72) as one library GEMM with
    the bias (+ ReLU) epilogue: a channels-last 1x1 convolution IS the plain GEMM (B*H*W, Cin) x (Cin, Cout), the
    eval-mode normalisation folds into it -- W' = W * s, b' = beta - mean * s, s = (|gamma| + eps) / sqrt(var + eps) -- and
    rocBLAS / hipBLASLt run that GEMM faster than MIOpen's implicit-GEMM convolution on the teacher's reduce layers
    (1024 -> 256 at 65 x 65, batch 8: 144 us with the epilogue vs 175 us convolution + 27 us ABN pass;
    profiles/r02d_conv1x1_blas.jsonl), exact fp32 (no xf32 on gfx950).  The folded operands are cached on the BN module."""
    key = tuple((t.data_ptr(), t._version) for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None)
    fold = getattr(bn, "_blas_fold", None)
    if fold is None or fold[0] != key:
        gamma = (bn.weight.abs() + bn.eps) if bn.weight is not None else torch.ones_like(bn.running_var)
        scale = gamma / torch.sqrt(bn.running_var + bn.eps)
        w2 = (conv.weight.reshape(conv.out_channels, conv.in_channels) * scale.view(-1, 1)).contiguous()
        b2 = (bn.bias if bn.bias is not None else torch.zeros_like(scale)) - bn.running_mean * scale
        fold = (key, w2, b2.contiguous())
        bn._blas_fold = fold
    _, w2, b2 = fold
    b, _, h, w = x.shape
    x2 = x.permute(0, 2, 3, 1).reshape(b * h * w, conv.in_channels)
    out = torch._addmm_activation(b2, x2, w2.t()) if relu else torch.addmm(b2, x2, w2.t())
    return out.view(b, h, w, conv.out_channels).permute(0, 3, 1, 2)


def conv1x1_abn_supported(x, conv):
    """True when the fused 1x1-convolution + eval-ABN GEMM of csrc/conv1x1.hip takes this call: fp32 channels-last
    input, a plain stride-1 1x1 convolution without bias, Cin a multiple of 16 and Cout of 128."""
    if not ((x.is_cuda or _lib.test_backend_active()) and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)):      # (CPU tensors only under the tests' C-ABI double)
        return False
    if not (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.bias is None and conv.weight.dtype == torch.float32):
        return False
    m = x.shape[0] * x.shape[2] * x.shape[3]
    return bool(_lib.get().skd_conv1x1_abn_supported(m, conv.in_channels, conv.out_channels))


def abn_pack_eval_params(bn):
    """(4, C) = [running_mean | 1 / sqrt(running_var + eps) | |weight| + eps | bias] of an eval-mode InPlace-ABN module (the
    constants of bn.cu:146-159), cached on the module and rebuilt when any of its tensors is written (their autograd version
    counters) or moved: the frozen teacher packs each BatchNorm once.  (A write through ``tensor.data`` does not bump the
    version counter autograd keeps for ``tensor``: code that edits statistics that way must drop ``bn._skd_eval_pack`` itself;
    ``load_state_dict``, optimizers and ordinary in-place ops are seen.)"""
    key = tuple((t.data_ptr(), t._version) if t is not None else None
                for t in (bn.running_mean, bn.running_var, bn.weight, bn.bias)) + (float(bn.eps),)
    cached = getattr(bn, "_skd_eval_pack", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    c = bn.running_mean.numel()
    pack = bn.running_mean.new_empty((4, c))
    _lib.check(_lib.get().skd_abn_pack_eval_params(c, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), _lib.ptr(bn.weight),
                                                   _lib.ptr(bn.bias), float(bn.eps), pack.data_ptr(), _lib.stream_of(pack)),
               "skd_abn_pack_eval_params")
    bn._skd_eval_pack = (key, pack)
    return pack


def conv1x1_abn_eval(x, conv_weight, running_mean, running_var, weight, bias, eps=1e-5, activation="relu", slope=0.01,
                     residual=None, pro=None):
    """act(bn_running(conv1x1(x)) [+ residual]) as ONE fp32-MFMA GEMM with the normalisation in its epilogue (inference
    only; networks/pspnet_combine.py:65-84 for the frozen teacher).  x (B, Cin, H, W) and residual / result
    (B, Cout, H, W) in channels-last memory; conv_weight (Cout, Cin, 1, 1).
    ``pro`` = ``abn_pack_eval_params(bn)`` of an eval-mode BatchNorm + ReLU that PRECEDES the convolution
    (bn2 -> relu -> conv3, pspnet_combine.py:71-75): applied to x on its way into the GEMM, x itself is left untouched."""
    if torch.is_grad_enabled() and (x.requires_grad or conv_weight.requires_grad):
        raise RuntimeError("conv1x1_abn_eval is inference-only")
    _lib.require_device(x, conv_weight, running_mean, running_var, weight, bias, residual)
    act = {"none": 0, "leaky_relu": 1, "relu": 3}[activation]
    b, k, h, w = x.shape
    n = conv_weight.shape[0]
    out = _new_cl(x, b, n, h, w)
    if residual is not None:
        residual = _cl(residual)
        if tuple(residual.shape) != (b, n, h, w):
            raise ValueError("residual shape %s != output shape %s" % (tuple(residual.shape), (b, n, h, w)))
    wt = conv_weight.reshape(n, k)      # (N, K, 1, 1): the same memory in NCHW and channels-last
    if not wt.is_contiguous():
        wt = wt.contiguous()
    if pro is not None:
        _lib.require_device(pro)
        if pro.dtype != torch.float32 or tuple(pro.shape) != (4, k) or not pro.is_contiguous():
            raise ValueError("pro must be the (4, Cin) fp32 pack of abn_pack_eval_params")
        _lib.check(_lib.get().skd_conv1x1_abn_pro_nhwc(b * h * w, k, n, x.data_ptr(), wt.data_ptr(), _lib.ptr(residual), out.data_ptr(),
                                                       running_mean.data_ptr(), running_var.data_ptr(), _lib.ptr(weight),
                                                       _lib.ptr(bias), float(eps), pro.data_ptr(), act, float(slope),
                                                       _lib.stream_of(x)), "skd_conv1x1_abn_pro_nhwc")
        return out
    _lib.check(_lib.get().skd_conv1x1_abn_nhwc(b * h * w, k, n, x.data_ptr(), wt.data_ptr(), _lib.ptr(residual), out.data_ptr(),
                                               running_mean.data_ptr(), running_var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                                               float(eps), act, float(slope), _lib.stream_of(x)), "skd_conv1x1_abn_nhwc")
    return out


def seg_confusion(logits, target=None, ignore_index=255, confusion=None, want_pred=True):
    """Evaluation tail (networks/evaluate.py:106-113, 186-198): bilinear (align_corners) upsample of ``logits``
    (B, C, h, w) to the size of ``target`` (B, H, W) [or of ``want_pred`` = (H, W) when no target], argmax over the
    classes (first maximum, uint8) and accumulation of the (C, C) int64 ``confusion`` matrix over the pixels whose
    label is not ``ignore_index`` -- one fused kernel, bit-exact integer results.  Returns (pred or None, confusion)."""
    _lib.require_device(logits, target, confusion)
    lg = _f32c(logits.detach(), "seg_confusion")
    b, c, h, w = lg.shape
    if target is not None:
        if target.dtype != torch.int64:
            raise TypeError("seg_confusion: int64 target expected (got %s)" % target.dtype)
        tg = target if target.is_contiguous() else target.contiguous()
        H, W = tg.shape[1], tg.shape[2]
        if confusion is None:
            confusion = torch.zeros((c, c), dtype=torch.int64, device=lg.device)
    else:
        tg = None
        H, W = want_pred
    pred = torch.empty((b, H, W), dtype=torch.uint8, device=lg.device) if want_pred else None
    _lib.check(_lib.get().skd_seg_confusion(b, c, h, w, H, W, lg.data_ptr(), _lib.ptr(tg), int(ignore_index), _lib.ptr(pred),
                                            _lib.ptr(confusion), _lib.stream_of(lg)), "skd_seg_confusion")
    return pred, confusion


def pool_out_size(n, k):
    """ceil_mode=True, stride = kernel, no padding."""
    return -(-n // k)


class _MaxPoolArgmax(Function):
    @staticmethod
    def forward(ctx, x, kh, kw):
        _lib.require_device(x)
        x = _f32c(x, "max_pool_argmax")
        b, c, h, w = x.shape
        oh, ow = pool_out_size(h, kh), pool_out_size(w, kw)
        lib, st = _lib.get(), _lib.stream_of(x)
        pooled = x.new_empty((b, c, oh, ow))
        need = ctx.needs_input_grad[0]
        index = torch.empty((b, c, oh, ow), dtype=torch.int32, device=x.device) if need else None
        _lib.check(lib.skd_maxpool_argmax(b * c, h, w, kh, kw, x.data_ptr(), pooled.data_ptr(),
                                          _lib.ptr(index), st), "skd_maxpool_argmax")
        ctx.geom = (b, c, h, w, kh, kw)
        ctx.save_for_backward(index)
        return pooled

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        if index is None:
            return None, None, None
        b, c, h, w, kh, kw = ctx.geom
        g = _f32c(g, "max_pool_argmax backward")
        m = g.shape[2] * g.shape[3]
        dx = g.new_empty((b, c, h, w))
        lib, st = _lib.get(), _lib.stream_of(g)
        _lib.check(lib.skd_maxunpool_scatter(b * c, h, w, kh, kw, g.data_ptr(), m, index.data_ptr(),
                                             dx.data_ptr(), st), "skd_maxunpool_scatter")
        return dx, None, None


def max_pool_argmax(x, kh, kw, return_indices=False):
    """MaxPool2d(kernel=stride=(kh,kw), padding=0, ceil_mode=True).  Indices (bit-exact with
    PyTorch's: flat h*W+w of the first maximum, int32) are available through max_pool_indices()."""
    return _MaxPoolArgmax.apply(x, int(kh), int(kw))


def max_pool_indices(x, kh, kw):
    """(pooled, int32 argmax) without autograd -- used by the parity tests."""
    _lib.require_device(x)
    x = _f32c(x, "max_pool_indices")
    b, c, h, w = x.shape
    oh, ow = pool_out_size(h, kh), pool_out_size(w, kw)
    pooled = x.new_empty((b, c, oh, ow))
    index = torch.empty((b, c, oh, ow), dtype=torch.int32, device=x.device)
    _lib.check(_lib.get().skd_maxpool_argmax(b * c, h, w, int(kh), int(kw), x.data_ptr(), pooled.data_ptr(),
                                             index.data_ptr(), _lib.stream_of(x)), "skd_maxpool_argmax")
    return pooled, index


class _SimDis(Function):
    """sim_dis_compute(f_S, f_T) on pooled features (B, C, oh, ow)."""

    @staticmethod
    def forward(ctx, f_s, f_t):
        _lib.require_device(f_s, f_t)
        f_s, f_t = _f32c(f_s, "sim_dis"), _f32c(f_t, "sim_dis")
        b, cs = f_s.shape[0], f_s.shape[1]
        ct = f_t.shape[1]
        m = f_t.shape[-1] * f_t.shape[-2]                      # utils.py:181
        assert f_s.shape[0] == f_t.shape[0] and f_s.shape[2:] == f_t.shape[2:]
        lib, st = _lib.get(), _lib.stream_of(f_s)
        ldm = lib.skd_pairwise_ldm(m)
        need = ctx.needs_input_grad[0]
        fh_s = f_s.new_empty((b, cs, ldm))
        fh_t = f_s.new_empty((b, ct, ldm))
        norm_s = f_s.new_empty((b, m)) if need else None
        _lib.check(lib.skd_channel_l2_normalise(b, cs, m, f_s.data_ptr(), fh_s.data_ptr(), ldm,
                                                None, 0, _lib.ptr(norm_s), st),
                   "skd_channel_l2_normalise")
        _lib.check(lib.skd_channel_l2_normalise(b, ct, m, f_t.data_ptr(), fh_t.data_ptr(), ldm,
                                                None, 0, None, st), "skd_channel_l2_normalise")
        g = f_s.new_empty((b, ldm, ldm)) if need else None
        loss = f_s.new_empty(())
        ws = f_s.new_empty((max(1, lib.skd_pairwise_workspace_floats(b, m)),))
        _lib.check(lib.skd_pairwise_gram_loss(b, cs, ct, m, ldm, fh_s.data_ptr(), fh_t.data_ptr(),
                                              _lib.ptr(g), loss.data_ptr(), ws.data_ptr(), st),
                   "skd_pairwise_gram_loss")
        ctx.geom = (tuple(f_s.shape), m, ldm)
        ctx.save_for_backward(fh_s if need else None, g, norm_s)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        fh_s, g, norm_s = ctx.saved_tensors
        if g is None:
            return None, None
        shape, m, ldm = ctx.geom
        b, cs = shape[0], shape[1]
        lib, st = _lib.get(), _lib.stream_of(g)
        gl = gl.to(torch.float32).contiguous()
        dp = g.new_empty((b, cs, ldm))
        ws = g.new_empty((max(1, lib.skd_pairwise_backward_workspace_floats(b, cs, m)),))
        _lib.check(lib.skd_pairwise_backward(b, cs, m, ldm, fh_s.data_ptr(), g.data_ptr(),
                                             norm_s.data_ptr(), gl.data_ptr(), dp.data_ptr(), ws.data_ptr(), st),
                   "skd_pairwise_backward")
        return dp[:, :, :m].reshape(shape), None


def sim_dis(f_s, f_t):
    """sum((A_T - A_S)^2) / M^2 / B with A = normalised Gram over channels (utils.py:173-183);
    the channel L2 norm is a constant for autograd (utils.py:175) and f_T gets no gradient."""
    return _SimDis.apply(f_s, f_t.detach())


def _channels_last_quads(t):
    """A (B, C, H, W) tensor that is stored channels-last with whole channel quads: the layout the PSP features have inside
    NetModel -- the pooling kernels read it as it is (skd_maxpool_argmax_nhwc) instead of through an NCHW copy."""
    return (t.dim() == 4 and t.shape[1] % 4 == 0 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()
            and t.data_ptr() % 16 == 0)


def _pool_feature(lib, t, kh, kw, pooled, index, st):
    """pooled (B, C, M) / int32 argmax (flat h * W + w, or None) of MaxPool2d((kh, kw), ceil_mode=True) for either layout.
    Returns the tensor that was read (the caller keeps its layout for the backward)."""
    b, c, h, w = t.shape
    if _channels_last_quads(t):
        _lib.check(lib.skd_maxpool_argmax_nhwc(b, c, h, w, kh, kw, t.data_ptr(), pooled.data_ptr(), _lib.ptr(index), st),
                   "skd_maxpool_argmax_nhwc")
        return t
    t = t if t.is_contiguous() else t.contiguous()
    _lib.check(lib.skd_maxpool_argmax(b * c, h, w, kh, kw, t.data_ptr(), pooled.data_ptr(), _lib.ptr(index), st), "skd_maxpool_argmax")
    return t


def _unpool_feature(lib, nhwc, geom, dp, ldp, index, st):
    """d loss / d feature from the pooled gradient dp (B, C, ldp) through the argmax, in the layout the feature had."""
    b, cs, h, w, kh, kw = geom
    if nhwc:
        dx = torch.empty_strided((b, cs, h, w), (h * w * cs, 1, w * cs, cs), dtype=dp.dtype, device=dp.device)   # channels-last
        _lib.check(lib.skd_maxunpool_scatter_nhwc(b, cs, h, w, kh, kw, dp.data_ptr(), ldp, index.data_ptr(), dx.data_ptr(), st),
                   "skd_maxunpool_scatter_nhwc")
        return dx
    dx = dp.new_empty((b, cs, h, w))
    _lib.check(lib.skd_maxunpool_scatter(b * cs, h, w, kh, kw, dp.data_ptr(), ldp, index.data_ptr(), dx.data_ptr(), st),
               "skd_maxunpool_scatter")
    return dx


class _PairWise(Function):
    """Whole Pa loss in one node: pool(+argmax) -> normalise -> Gram/loss; backward scatters the
    pooled gradient straight from the padded GEMM output (no slice/copy in between).  Features may arrive NCHW-contiguous or
    channels-last (round 5: NetModel hands the PSP features over as they are); the student's gradient comes back in the
    layout its feature had."""

    @staticmethod
    def forward(ctx, feat_s, feat_t, kh, kw):
        _lib.require_device(feat_s, feat_t)
        for t in (feat_s, feat_t):
            if t.dtype != torch.float32:
                raise TypeError("pair_wise_loss: fp32 tensors only (got %s)" % t.dtype)
        b, cs, h, w = feat_s.shape
        ct = feat_t.shape[1]
        assert feat_t.shape[0] == b and tuple(feat_t.shape[2:]) == (h, w)
        oh, ow = pool_out_size(h, kh), pool_out_size(w, kw)
        m = oh * ow
        lib, st = _lib.get(), _lib.stream_of(feat_s)
        need = ctx.needs_input_grad[0]
        ctx.nhwc = _channels_last_quads(feat_s)
        if m <= 64:
            # small graphs (the reference default pools to 3 x 3 = 9 nodes): pool, then ONE fused launch for
            # normalise + both Grams + loss + gradient w.r.t. the pooled student features
            p_s, p_t = feat_s.new_empty((b, cs, m)), feat_s.new_empty((b, ct, m))
            index = torch.empty((b, cs, m), dtype=torch.int32, device=feat_s.device) if need else None
            _pool_feature(lib, feat_s, kh, kw, p_s, index, st)
            _pool_feature(lib, feat_t, kh, kw, p_t, None, st)
            loss = feat_s.new_empty(())
            dp = feat_s.new_empty((b, cs, m)) if need else None
            ws = feat_s.new_empty((b,))
            _lib.check(lib.skd_pairwise_small(b, cs, ct, m, p_s.data_ptr(), p_t.data_ptr(), loss.data_ptr(), _lib.ptr(dp),
                                              ws.data_ptr(), st), "skd_pairwise_small")
            ctx.geom = (b, cs, h, w, kh, kw, m, 0, 0)
            ctx.small = True
            ctx.save_for_backward(dp, index)
            return loss
        ctx.small = False
        ldm = lib.skd_pairwise_ldm(m)
        p_s = feat_s.new_empty((b, cs, m))
        p_t = feat_s.new_empty((b, ct, m))
        index = torch.empty((b, cs, m), dtype=torch.int32, device=feat_s.device) if need else None
        _pool_feature(lib, feat_s, kh, kw, p_s, index, st)
        _pool_feature(lib, feat_t, kh, kw, p_t, None, st)
        fh_s = feat_s.new_empty((b, cs, ldm))
        fh_t = feat_s.new_empty((b, ct, ldm))
        norm_s = feat_s.new_empty((b, m)) if need else None
        _lib.check(lib.skd_channel_l2_normalise(b, cs, m, p_s.data_ptr(), fh_s.data_ptr(), ldm,
                                                None, 0, _lib.ptr(norm_s), st),
                   "skd_channel_l2_normalise")
        _lib.check(lib.skd_channel_l2_normalise(b, ct, m, p_t.data_ptr(), fh_t.data_ptr(), ldm,
                                                None, 0, None, st), "skd_channel_l2_normalise")
        g = feat_s.new_empty((b, ldm, ldm)) if need else None
        loss = feat_s.new_empty(())
        ws = feat_s.new_empty((max(1, lib.skd_pairwise_workspace_floats(b, m)),))
        _lib.check(lib.skd_pairwise_gram_loss(b, cs, ct, m, ldm, fh_s.data_ptr(), fh_t.data_ptr(),
                                              _lib.ptr(g), loss.data_ptr(), ws.data_ptr(), st),
                   "skd_pairwise_gram_loss")
        ctx.geom = (b, cs, h, w, kh, kw, m, ldm, 0)
        ctx.save_for_backward(fh_s if need else None, g, norm_s, index)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        if ctx.small:
            dp, index = ctx.saved_tensors
            if dp is None:
                return None, None, None, None
            b, cs, h, w, kh, kw, m, _, _ = ctx.geom
            lib, st = _lib.get(), _lib.stream_of(dp)
            dpg = dp * gl.to(torch.float32)
            return _unpool_feature(lib, ctx.nhwc, (b, cs, h, w, kh, kw), dpg, m, index, st), None, None, None
        fh_s, g, norm_s, index = ctx.saved_tensors
        if g is None:
            return None, None, None, None
        b, cs, h, w, kh, kw, m, ldm, _ = ctx.geom
        lib, st = _lib.get(), _lib.stream_of(g)
        gl = gl.to(torch.float32).contiguous()
        dp = g.new_empty((b, cs, ldm))
        ws = g.new_empty((max(1, lib.skd_pairwise_backward_workspace_floats(b, cs, m)),))
        _lib.check(lib.skd_pairwise_backward(b, cs, m, ldm, fh_s.data_ptr(), g.data_ptr(),
                                             norm_s.data_ptr(), gl.data_ptr(), dp.data_ptr(), ws.data_ptr(), st),
                   "skd_pairwise_backward")
        return _unpool_feature(lib, ctx.nhwc, (b, cs, h, w, kh, kw), dp, ldm, index, st), None, None, None


def pair_wise_loss(feat_s, feat_t, kh, kw):
    """CriterionPairWiseforWholeFeatAfterPool.forward on raw (B,C,H,W) features, criterion.py:236-245."""
    return _PairWise.apply(feat_s, feat_t.detach(), int(kh), int(kw))


class _SpectralNormalize(Function):
    """w = w_bar / sigma after ONE power iteration that updates u and v in place (spectral.py:23-35).

    u and v are held by reference, not snapshotted: the reference rebinds ``u.data`` / ``v.data``
    without an autograd version bump, so a graph recorded by an earlier forward is back-propagated
    with whatever u, v hold at backward time (the D step runs three forwards before its single
    backward, kd_model.py:156-164).  Reading them at backward time reproduces that exactly.
    """

    @staticmethod
    def forward(ctx, w_bar, u, v):
        _lib.require_device(w_bar, u, v)
        wb = _f32c(w_bar, "spectral_normalize")
        h = wb.shape[0]
        wd = wb.numel() // h
        if not (u.is_contiguous() and v.is_contiguous()) or u.numel() != h or v.numel() != wd:
            raise ValueError("spectral_normalize: u/v must be contiguous vectors of length (out, in*kh*kw)")
        lib, st = _lib.get(), _lib.stream_of(wb)
        sigma = wb.new_empty(())
        w = torch.empty_like(wb)
        ws = wb.new_empty((max(1, lib.skd_spectral_workspace_floats(h, wd)),))
        _lib.check(lib.skd_spectral_norm_forward(h, wd, wb.data_ptr(), u.data_ptr(), v.data_ptr(),
                                                 sigma.data_ptr(), w.data_ptr(), ws.data_ptr(), st),
                   "skd_spectral_norm_forward")
        ctx.u, ctx.v = u, v
        ctx.save_for_backward(wb, sigma)
        return w

    @staticmethod
    @once_differentiable
    def backward(ctx, gw):
        wb, sigma = ctx.saved_tensors
        h = wb.shape[0]
        wd = wb.numel() // h
        gw = _f32c(gw, "spectral_normalize backward")
        lib, st = _lib.get(), _lib.stream_of(wb)
        gwb = torch.empty_like(wb)
        ws = wb.new_empty((max(1, lib.skd_spectral_workspace_floats(h, wd)),))
        _lib.check(lib.skd_spectral_norm_backward(h, wd, wb.data_ptr(), ctx.u.data_ptr(), ctx.v.data_ptr(),
                                                  sigma.data_ptr(), gw.data_ptr(), gwb.data_ptr(),
                                                  ws.data_ptr(), st), "skd_spectral_norm_backward")
        return gwb, None, None


def spectral_normalize(w_bar, u, v):
    return _SpectralNormalize.apply(w_bar, u, v)


class _SpectralNormalizeMulti(Function):
    """``_SpectralNormalize`` for ALL spectrally normalised layers of a network at once (skd_spectral_norm_*_multi): three
    launches forward and two backward for the whole set instead of per layer.  ``us`` / ``vs``: lists of the layers' u / v
    tensors (held by reference and read again at backward time, like the single-layer op); returns one normalised weight per
    layer."""

    @staticmethod
    def forward(ctx, us, vs, *w_bars):
        _lib.require_device(*w_bars, *us, *vs)
        wbs = [_f32c(w, "spectral_normalize_multi") for w in w_bars]
        hs = [w.shape[0] for w in wbs]
        ws_ = [w.numel() // h for w, h in zip(wbs, hs)]
        for u, v, h, wd in zip(us, vs, hs, ws_):
            if not (u.is_contiguous() and v.is_contiguous()) or u.numel() != h or v.numel() != wd:
                raise ValueError("spectral_normalize: u/v must be contiguous vectors of length (out, in*kh*kw)")
        lib, st = _lib.get(), _lib.stream_of(wbs[0])
        sigmas = wbs[0].new_empty((len(wbs),))
        outs = [torch.empty_like(w) for w in wbs]
        work = wbs[0].new_empty((sum(max(1, lib.skd_spectral_workspace_floats(h, wd)) for h, wd in zip(hs, ws_)),))
        sig = [sigmas[k:k + 1] for k in range(len(wbs))]
        _lib.check(lib.skd_spectral_norm_forward_multi(len(wbs), _lib.int_array(hs), _lib.int_array(ws_), _lib.ptr_array(wbs),
                                                       _lib.ptr_array(us), _lib.ptr_array(vs), _lib.ptr_array(sig),
                                                       _lib.ptr_array(outs), work.data_ptr(), st), "skd_spectral_norm_forward_multi")
        ctx.us, ctx.vs, ctx.hs, ctx.ws_ = list(us), list(vs), hs, ws_
        ctx.save_for_backward(sigmas, *wbs)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gws):
        sigmas, *wbs = ctx.saved_tensors
        lib, st = _lib.get(), _lib.stream_of(wbs[0])
        gws = [torch.zeros_like(w) if g is None else _f32c(g, "spectral_normalize backward") for g, w in zip(gws, wbs)]
        gwbs = [torch.empty_like(w) for w in wbs]
        work = wbs[0].new_empty((sum(max(1, lib.skd_spectral_workspace_floats(h, wd)) for h, wd in zip(ctx.hs, ctx.ws_)),))
        sig = [sigmas[k:k + 1] for k in range(len(wbs))]
        _lib.check(lib.skd_spectral_norm_backward_multi(len(wbs), _lib.int_array(ctx.hs), _lib.int_array(ctx.ws_), _lib.ptr_array(wbs),
                                                        _lib.ptr_array(ctx.us), _lib.ptr_array(ctx.vs), _lib.ptr_array(sig),
                                                        _lib.ptr_array(gws), _lib.ptr_array(gwbs), work.data_ptr(), st),
                   "skd_spectral_norm_backward_multi")
        return (None, None) + tuple(gwbs)


def spectral_normalize_multi(w_bars, us, vs):
    """[w_bar_k / sigma_k] after one power iteration per layer (u_k, v_k updated in place): all layers in 3 launches."""
    return _SpectralNormalizeMulti.apply(list(us), list(vs), *w_bars)


def spectral_power_iteration(w_bar, u, v):
    """u, v update only (extra iterations when power_iterations > 1); returns sigma (0-dim)."""
    _lib.require_device(w_bar, u, v)
    wb = _f32c(w_bar.detach(), "spectral_power_iteration")
    h = wb.shape[0]
    wd = wb.numel() // h
    lib, st = _lib.get(), _lib.stream_of(wb)
    sigma = wb.new_empty(())
    ws = wb.new_empty((max(1, lib.skd_spectral_workspace_floats(h, wd)),))
    _lib.check(lib.skd_spectral_norm_forward(h, wd, wb.data_ptr(), u.data_ptr(), v.data_ptr(),
                                             sigma.data_ptr(), None, ws.data_ptr(), st),
               "skd_spectral_norm_forward")
    return sigma
