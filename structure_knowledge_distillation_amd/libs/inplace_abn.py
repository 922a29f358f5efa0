"""Autograd functions for InPlace-ABN on MI355X.

Mirrors the *contract* of the reference's libs/functions.py:70-309 (in-place on the conv output,
forward output saved for backward, not double-differentiable, ValueError on non-contiguous input,
RuntimeError when a native call reports failure) on top of the fused kernels of csrc/abn.hip:

    reference (per layer, training)                  here
    ---------------------------------------------    -------------------------------------------
    K1 mean_var (2 passes over x)                    stats partials (1 pass) + finalize
    2 torch ops for the running stats                  (fused into finalize)
    K2 forward  + K5 leaky-relu pass                 one normalise+affine+activation pass
    K6 + K5(1/slope) rewrite dz, z ; K3 ; K4         reduce pass + dx pass, activation undone in
                                                     registers (z and dz are never rewritten)

Cross-replica synchronisation (libs/functions.py:185-205, 263-280: master/worker Queues +
comm.gather / broadcast_coalesced inside one process) is replaced by one-process-per-GPU
collectives on a torch.distributed group (RCCL over xGMI): all_gather of [mean, var] followed by
the reference's combine rule, and an averaged all_reduce of [edz, eydz].
"""
import os

import torch
import torch.autograd as autograd
import torch.distributed as dist
from torch.autograd.function import once_differentiable

from .. import _lib

ACT_LEAKY_RELU = "leaky_relu"
ACT_ELU = "elu"
ACT_NONE = "none"
ACT_RELU = "relu"          # forward-only (inference) fusion of the nn.ReLU that follows a BatchNorm2d
_ACT_CODE = {ACT_NONE: 0, ACT_LEAKY_RELU: 1, ACT_ELU: 2}
_EVAL_ACT_CODE = dict(_ACT_CODE, **{ACT_RELU: 3})


def _act_code(name):
    try:
        return _ACT_CODE[name]
    except KeyError:
        raise ValueError("unknown activation %r (expected leaky_relu, elu or none)" % (name,))


def _check_contiguous(*tensors):
    # libs/functions.py:65-67
    if not all(t is None or t.is_contiguous() for t in tensors):
        raise ValueError("Non-contiguous input")


def _dims(x):
    n, c = x.shape[0], x.shape[1]
    s = 1
    for d in x.shape[2:]:
        s *= d
    return n, c, s


def _same_phase(a, b):
    return ((a.data_ptr() ^ b.data_ptr()) & 15) == 0


def _group_size(group):
    if group is None:
        return 1
    return dist.get_world_size(group)


def _replicated(group):
    """The synchronised form applies: a group of more than one rank (or of one rank under utils.parallel.solo_rehearsal)."""
    if group is None:
        return False
    from ..utils import parallel
    return parallel.replicated(group)


def _default_group():
    """The default process group when the layer has to synchronise over it, else None."""
    from ..utils import parallel
    return dist.group.WORLD if parallel.replicated() else None


def combine_replica_stats(gathered):
    """gathered: (G, 2, C) per-rank [mean, var] -> combined (mean, var).

    The reference rule, libs/functions.py:196-197 (equal per-rank sample counts assumed):
        mean = means.mean(0);  var = (vars + (mean - means)**2).mean(0)
    """
    means, vars_ = gathered[:, 0], gathered[:, 1]
    mean = means.mean(0)
    var = (vars_ + (mean - means) ** 2).mean(0)
    return mean, var


def _is_nhwc(x):
    return x.dim() == 4 and (not x.is_contiguous()) and x.is_contiguous(memory_format=torch.channels_last)


class _Geom:
    """Layout of one activation tensor and the C-ABI entry set that goes with it.

    NCHW (the reference's layout, lib_cffi.cpp:24-34): (N, C, S) + the skd_abn_* entries.
    Channels-last: (rows = N*H*W, C) + the skd_abn_*_nhwc entries (C a power of two in [4, 1024]); lets MIOpen
    run its NHWC-native fp32 kernels without transposes."""

    def __init__(self, x):
        self.nhwc = _is_nhwc(x)
        if self.nhwc:
            c = x.shape[1]
            if c < 4 or c > 1024 or (c & (c - 1)):
                raise ValueError("channels-last InPlaceABN needs a power-of-two channel count in [4, 1024] (got %d)" % c)
            self.rows, self.c = x.shape[0] * x.shape[2] * x.shape[3], c
            self.count = self.rows
        else:
            _check_contiguous(x)
            self.n, self.c, self.s = _dims(x)
            self.count = self.n * self.s

    def like(self, t):
        """t in this tensor's memory format (gradients arrive in whatever format autograd produced)."""
        if self.nhwc:
            return t if _is_nhwc(t) else t.contiguous(memory_format=torch.channels_last)
        return t.contiguous()

    def workspace(self, lib, ref):
        n = (lib.skd_abn_nhwc_workspace_floats(self.rows, self.c) if self.nhwc
             else lib.skd_abn_workspace_floats(self.n, self.c, self.s))
        return ref.new_empty((max(1, n),))

    def _d(self):
        return (self.rows, self.c) if self.nhwc else (self.n, self.c, self.s)

    def stats(self, lib, x, mean, var, ws, st):
        fn = lib.skd_abn_stats_nhwc if self.nhwc else lib.skd_abn_stats
        _lib.check(fn(*self._d(), x.data_ptr(), mean.data_ptr(), var.data_ptr(), ws.data_ptr(), st), "skd_abn_stats")

    def forward_train(self, lib, x, res, out, weight, bias, rm, rv, mean, var, momentum, eps, act, slope, ws, st):
        if self.nhwc:
            _lib.check(lib.skd_abn_forward_train_nhwc(self.rows, self.c, x.data_ptr(), _lib.ptr(res), out.data_ptr(),
                                                      _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(rm), _lib.ptr(rv),
                                                      mean.data_ptr(), var.data_ptr(), momentum, eps, act, slope,
                                                      ws.data_ptr(), st), "skd_abn_forward_train_nhwc")
        elif res is None and out.data_ptr() == x.data_ptr():
            _lib.check(lib.skd_abn_forward_train(self.n, self.c, self.s, x.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                                                 _lib.ptr(rm), _lib.ptr(rv), mean.data_ptr(), var.data_ptr(), momentum,
                                                 eps, act, slope, ws.data_ptr(), st), "skd_abn_forward_train")
        else:
            _lib.check(lib.skd_abn_forward_train_to(self.n, self.c, self.s, x.data_ptr(), _lib.ptr(res), out.data_ptr(),
                                                    _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(rm), _lib.ptr(rv),
                                                    mean.data_ptr(), var.data_ptr(), momentum, eps, act, slope,
                                                    ws.data_ptr(), st), "skd_abn_forward_train_to")

    def apply_to(self, lib, x, res, out, mean, var, weight, bias, eps, act, slope, st):
        if self.nhwc:
            _lib.check(lib.skd_abn_apply_nhwc_to(self.rows, self.c, x.data_ptr(), _lib.ptr(res), out.data_ptr(),
                                                 mean.data_ptr(), var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), eps,
                                                 act, slope, st), "skd_abn_apply_nhwc_to")
        elif res is None and out.data_ptr() == x.data_ptr():
            _lib.check(lib.skd_abn_apply(self.n, self.c, self.s, x.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                         _lib.ptr(weight), _lib.ptr(bias), eps, act, slope, st), "skd_abn_apply")
        else:
            _lib.check(lib.skd_abn_apply_to(self.n, self.c, self.s, x.data_ptr(), _lib.ptr(res), out.data_ptr(),
                                            mean.data_ptr(), var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), eps, act,
                                            slope, st), "skd_abn_apply_to")

    def backward_reduce(self, lib, z, dz, weight, bias, edz, eydz, eps, act, slope, ws, st):
        fn = lib.skd_abn_backward_reduce_nhwc if self.nhwc else lib.skd_abn_backward_reduce
        _lib.check(fn(*self._d(), z.data_ptr(), dz.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), edz.data_ptr(),
                      eydz.data_ptr(), eps, act, slope, ws.data_ptr(), st), "skd_abn_backward_reduce")

    def new_param_grad(self, ref):
        """Buffer for dweight / dbias: the channels-last dx entries WRITE them (accumulate = 0), the NCHW entries keep
        the reference's accumulate-into convention (bn.cu:217-229) and need zeros."""
        return torch.empty_like(ref) if self.nhwc else torch.zeros_like(ref)

    def backward_dx(self, lib, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, act, slope, st):
        if self.nhwc:
            _lib.check(lib.skd_abn_backward_dx_nhwc(self.rows, self.c, z.data_ptr(), dz.data_ptr(), var.data_ptr(),
                                                    _lib.ptr(weight), _lib.ptr(bias), edz.data_ptr(), eydz.data_ptr(),
                                                    _lib.ptr(dx), _lib.ptr(dweight), _lib.ptr(dbias), eps, act, slope, 0, st),
                       "skd_abn_backward_dx_nhwc")
            return
        fn = lib.skd_abn_backward_dx
        _lib.check(fn(*self._d(), z.data_ptr(), dz.data_ptr(), var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                      edz.data_ptr(), eydz.data_ptr(), _lib.ptr(dx), _lib.ptr(dweight), _lib.ptr(dbias), eps, act, slope,
                      st), "skd_abn_backward_dx")

    def relu_backward_reduce(self, lib, x, out, dout, mean, var, edz, eydz, eps, ws, st, weight=None, bias=None):
        if out is None:      # channels-last forward without residual: the mask is recomputed from x (4 B/element less)
            _lib.check(lib.skd_abn_relu_backward_reduce_nhwc_x(self.rows, self.c, x.data_ptr(), dout.data_ptr(), mean.data_ptr(),
                                                               var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), edz.data_ptr(),
                                                               eydz.data_ptr(), eps, ws.data_ptr(), st),
                       "skd_abn_relu_backward_reduce_nhwc_x")
            return
        fn = lib.skd_abn_relu_backward_reduce_nhwc if self.nhwc else lib.skd_abn_relu_backward_reduce
        _lib.check(fn(*self._d(), x.data_ptr(), out.data_ptr(), dout.data_ptr(), mean.data_ptr(), var.data_ptr(),
                      edz.data_ptr(), eydz.data_ptr(), eps, ws.data_ptr(), st), "skd_abn_relu_backward_reduce")

    def relu_backward_dx(self, lib, x, out, dout, mean, var, weight, edz, eydz, dx, dres, dweight, dbias, eps, st, bias=None):
        if out is None:
            _lib.check(lib.skd_abn_relu_backward_dx_nhwc_x(self.rows, self.c, x.data_ptr(), dout.data_ptr(), mean.data_ptr(),
                                                           var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), edz.data_ptr(),
                                                           eydz.data_ptr(), dx.data_ptr(), _lib.ptr(dweight), _lib.ptr(dbias),
                                                           eps, 0, st), "skd_abn_relu_backward_dx_nhwc_x")
            return
        if self.nhwc:
            _lib.check(lib.skd_abn_relu_backward_dx_nhwc(self.rows, self.c, x.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                                         mean.data_ptr(), var.data_ptr(), _lib.ptr(weight), edz.data_ptr(),
                                                         eydz.data_ptr(), dx.data_ptr(), _lib.ptr(dres), _lib.ptr(dweight),
                                                         _lib.ptr(dbias), eps, 0, st), "skd_abn_relu_backward_dx_nhwc")
            return
        fn = lib.skd_abn_relu_backward_dx
        _lib.check(fn(*self._d(), x.data_ptr(), out.data_ptr(), dout.data_ptr(), mean.data_ptr(), var.data_ptr(),
                      _lib.ptr(weight), edz.data_ptr(), eydz.data_ptr(), dx.data_ptr(), _lib.ptr(dres),
                      _lib.ptr(dweight), _lib.ptr(dbias), eps, st), "skd_abn_relu_backward_dx")


def _replica_weights(group):
    from ..utils import parallel
    w = parallel.replica_weights()
    if w is not None and w.numel() == _group_size(group):
        return w
    return None


def _sync_stats(stat, c, count, group, running_mean, running_var, momentum, lib, st):
    """Cross-replica statistics (libs/functions.py:185-209): all_gather [mean, var], the reference combine rule
    (pooled with the per-rank sample weights when utils.parallel.set_replica_batch announced them), running-stat
    update with the pooled n.  Returns contiguous (mean, var)."""
    from ..utils.parallel import comm_timer, SyncMailbox
    g = _group_size(group)
    mb = SyncMailbox.get(group, stat.device)
    if mb is not None and c <= mb.max_channels:
        # one launch: store into every replica's mailbox (one xGMI hop), flags, wait, combine + running update (csrc/sync.hip)
        w = _replica_weights(group)
        out = stat.new_empty((2, c))
        tok = comm_timer.begin("syncabn", stat)
        _lib.check(lib.skd_abn_sync_stats(mb.ctx, c, stat.data_ptr(), _lib.ptr(w), out[0].data_ptr(), out[1].data_ptr(),
                                          _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum),
                                          float(count) if w is not None else float(count * g), st), "skd_abn_sync_stats")
        comm_timer.end(tok)
        return out[0], out[1]
    gathered = stat.new_empty((g, 2, c))
    tok = comm_timer.begin("syncabn", stat)
    dist.all_gather_into_tensor(gathered.view(-1), stat.view(-1), group=group)
    comm_timer.end(tok)
    out = stat.new_empty((2, c))
    w = _replica_weights(group)
    _lib.check(lib.skd_abn_combine_stats(g, c, gathered.data_ptr(), _lib.ptr(w), dist.get_rank(group) if w is not None else 0,
                                         out[0].data_ptr(), out[1].data_ptr(), _lib.ptr(running_mean),
                                         _lib.ptr(running_var), float(momentum),
                                         float(count) if w is not None else float(count * g), st), "skd_abn_combine_stats")
    return out[0], out[1]


def _mailbox_for(geo, group, ref):
    """The group's IPC mailbox context when this (channels-last) call can use the one-call synchronised entries of
    include/skd.h section 12 (statistics exchange inside the library: ONE register-resident launch when the tensor fits)."""
    if not geo.nhwc:
        return None
    from ..utils.parallel import SyncMailbox
    mb = SyncMailbox.get(group, ref.device)
    return mb if (mb is not None and geo.c <= mb.max_channels) else None


def _pooled_count(count, group):
    """(replica weights or None, n) as skd_abn_combine_stats wants them: n = this replica's count with weights, pooled without."""
    w = _replica_weights(group)
    return w, (float(count) if w is not None else float(count * _group_size(group)))


def _sync_grad_stats(stat, group):
    """libs/functions.py:271-272: [edz, eydz] are averaged over the replicas -- weighted by the per-rank sample
    counts when they are known (the pooled expectation), plain mean otherwise."""
    from ..utils.parallel import comm_timer, SyncMailbox
    w = _replica_weights(group)
    mb = SyncMailbox.get(group, stat.device)
    if mb is not None and stat.shape[1] <= mb.max_channels:
        tok = comm_timer.begin("syncabn", stat)
        _lib.check(mb.lib.skd_abn_sync_grad_stats(mb.ctx, stat.shape[1], stat.data_ptr(), _lib.ptr(w), _lib.stream_of(stat)),
                   "skd_abn_sync_grad_stats")
        comm_timer.end(tok)
        return
    if w is not None:
        stat.mul_(w[dist.get_rank(group)])
    tok = comm_timer.begin("syncabn", stat)
    dist.all_reduce(stat, op=dist.ReduceOp.SUM, group=group)
    comm_timer.end(tok)
    if w is None:
        stat.div_(_group_size(group))


class _InPlaceABN(autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps,
                activation, slope, group):
        _lib.require_device(x, weight, bias, running_mean, running_var)
        if x.dtype != torch.float32:
            raise TypeError("InPlaceABN kernels are fp32 only (got %s)" % x.dtype)
        ctx.training = bool(training)
        ctx.eps = float(eps)
        ctx.act = _act_code(activation)
        ctx.slope = float(slope)
        ctx.group = group if _replicated(group) else None
        if x.dim() < 2:
            raise ValueError("InPlaceABN expects (N, C, ...) input")
        _check_contiguous(weight, bias, running_mean, running_var)
        if x.numel() == 0:
            _check_contiguous(x)
            ctx.var = running_var
            ctx.save_for_backward(x, weight, bias)
            ctx.mark_dirty(x)
            return x
        geo = _Geom(x)          # raises ValueError("Non-contiguous input") like libs/functions.py:65-67
        c = geo.c
        lib = _lib.get()
        st = _lib.stream_of(x)

        if ctx.training:
            stat = x.new_empty((2, c))
            mean, var = stat[0], stat[1]
            ws = geo.workspace(lib, x)
            if ctx.group is None:
                geo.forward_train(lib, x, None, x, weight, bias, running_mean, running_var, mean, var, float(momentum),
                                  ctx.eps, ctx.act, ctx.slope, ws, st)
            else:
                mb = _mailbox_for(geo, ctx.group, x)
                if mb is not None:
                    from ..utils.parallel import comm_timer
                    w, n = _pooled_count(geo.count, ctx.group)
                    tok = comm_timer.begin("syncabn_fused", x)
                    _lib.check(lib.skd_abn_forward_train_nhwc_sync(
                        mb.ctx, geo.rows, c, x.data_ptr(), None, x.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                        _lib.ptr(running_mean), _lib.ptr(running_var), mean.data_ptr(), var.data_ptr(), _lib.ptr(w),
                        float(momentum), ctx.eps, ctx.act, ctx.slope, n, ws.data_ptr(), st), "skd_abn_forward_train_nhwc_sync")
                    comm_timer.end(tok)
                else:
                    geo.stats(lib, x, mean, var, ws, st)
                    mean, var = _sync_stats(stat, c, geo.count, ctx.group, running_mean, running_var, momentum, lib, st)
                    geo.apply_to(lib, x, None, x, mean, var, weight, bias, ctx.eps, ctx.act, ctx.slope, st)
        else:
            var = running_var
            geo.apply_to(lib, x, None, x, running_mean, running_var, weight, bias, ctx.eps, ctx.act, ctx.slope, st)

        ctx.var = var
        ctx.save_for_backward(x, weight, bias)
        ctx.mark_dirty(x)
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        z, weight, bias = ctx.saved_tensors
        var = ctx.var
        del ctx.var
        need_dx, need_dw, need_db = ctx.needs_input_grad[0:3]
        if z.numel() == 0:
            return (torch.zeros_like(z) if need_dx else None,
                    torch.zeros_like(weight) if need_dw and weight is not None else None,
                    torch.zeros_like(bias) if need_db and bias is not None else None,
                    None, None, None, None, None, None, None, None)
        geo = _Geom(z)
        c = geo.c
        dz = geo.like(dz)
        if not geo.nhwc and not _same_phase(z, dz):
            dz = dz.clone(memory_format=torch.contiguous_format)
        lib = _lib.get()
        st = _lib.stream_of(z)

        dx = torch.empty_like(z) if need_dx else None
        dweight = geo.new_param_grad(weight) if (need_dw and weight is not None) else None
        dbias = geo.new_param_grad(bias) if (need_db and bias is not None) else None
        # functions.py:146-147: inference-mode backward uses edz = eydz = 0 (training overwrites both)
        stat = z.new_empty((2, c)) if ctx.training else z.new_zeros((2, c))
        edz, eydz = stat[0], stat[1]
        if dx is None and geo.nhwc:
            dx = torch.empty_like(z)        # the channels-last dx entry always writes dx
        if ctx.training and ctx.group is None and geo.nhwc:
            # reduce + dx in one call: one register-resident launch when the tensor fits (csrc/abn.hip), else two
            ws = geo.workspace(lib, z)
            _lib.check(lib.skd_abn_backward_nhwc(geo.rows, c, z.data_ptr(), dz.data_ptr(), var.data_ptr(), _lib.ptr(weight),
                                                 _lib.ptr(bias), edz.data_ptr(), eydz.data_ptr(), dx.data_ptr(),
                                                 _lib.ptr(dweight), _lib.ptr(dbias), ctx.eps, ctx.act, ctx.slope, 0,
                                                 ws.data_ptr(), st), "skd_abn_backward_nhwc")
            return (dx if need_dx else None), dweight, dbias, None, None, None, None, None, None, None, None
        mb = _mailbox_for(geo, ctx.group, z) if (ctx.training and ctx.group is not None) else None
        if mb is not None:
            from ..utils.parallel import comm_timer
            ws = geo.workspace(lib, z)
            tok = comm_timer.begin("syncabn_fused", z)
            _lib.check(lib.skd_abn_backward_nhwc_sync(mb.ctx, geo.rows, c, z.data_ptr(), dz.data_ptr(), var.data_ptr(),
                                                      _lib.ptr(weight), _lib.ptr(bias), edz.data_ptr(), eydz.data_ptr(),
                                                      dx.data_ptr(), _lib.ptr(dweight), _lib.ptr(dbias),
                                                      _lib.ptr(_replica_weights(ctx.group)), ctx.eps, ctx.act, ctx.slope, 0,
                                                      ws.data_ptr(), st), "skd_abn_backward_nhwc_sync")
            comm_timer.end(tok)
            return (dx if need_dx else None), dweight, dbias, None, None, None, None, None, None, None, None
        if ctx.training:
            ws = geo.workspace(lib, z)
            geo.backward_reduce(lib, z, dz, weight, bias, edz, eydz, ctx.eps, ctx.act, ctx.slope, ws, st)
            if ctx.group is not None:
                _sync_grad_stats(stat, ctx.group)
        geo.backward_dx(lib, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, ctx.eps, ctx.act, ctx.slope, st)
        return (dx if need_dx else None), dweight, dbias, None, None, None, None, None, None, None, None


class _ABNRelu(autograd.Function):
    """Training-time ``relu(bn(x) [+ residual])`` as one op (csrc/abn.hip, "out of place" sections).

    The reference runs InPlace-ABN(activation='none') and then nn.ReLU -- at the tail of a residual block
    ``out + residual`` in between (networks/pspnet_combine.py:36-43, 68-82) -- keeping z and relu(z) alive.  Here the
    convolution output x is kept instead of z (y is recomputed from x, mean, var in backward) and ``out`` is the
    ReLU output, so the same two tensors per layer live on while the separate ReLU / add passes disappear.
    Works on NCHW and on channels-last tensors; cross-replica statistics exactly as in _InPlaceABN."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, momentum, eps, group):
        _lib.require_device(x, weight, bias, running_mean, running_var, residual)
        if x.dtype != torch.float32:
            raise TypeError("InPlaceABN kernels are fp32 only (got %s)" % x.dtype)
        _check_contiguous(weight, bias, running_mean, running_var)
        ctx.eps = float(eps)
        ctx.group = group if _replicated(group) else None
        geo = _Geom(x)
        c = geo.c
        lib, st = _lib.get(), _lib.stream_of(x)
        out = torch.empty_like(x)
        if residual is not None:
            residual = geo.like(residual)
            if not geo.nhwc and not _same_phase(x, residual):
                residual = residual.clone(memory_format=torch.contiguous_format)
        if not geo.nhwc and not _same_phase(x, out):   # cannot happen with the caching allocator's 512-byte granularity
            raise RuntimeError("abn_relu: misaligned output allocation")
        stat = x.new_empty((2, c))
        mean, var = stat[0], stat[1]
        ws = geo.workspace(lib, x)
        relu = _EVAL_ACT_CODE[ACT_RELU]
        if ctx.group is None:
            geo.forward_train(lib, x, residual, out, weight, bias, running_mean, running_var, mean, var, float(momentum),
                              ctx.eps, relu, 0.0, ws, st)
        else:
            mb = _mailbox_for(geo, ctx.group, x)
            if mb is not None:
                from ..utils.parallel import comm_timer
                w, n = _pooled_count(geo.count, ctx.group)
                tok = comm_timer.begin("syncabn_fused", x)
                _lib.check(lib.skd_abn_forward_train_nhwc_sync(
                    mb.ctx, geo.rows, c, x.data_ptr(), _lib.ptr(residual), out.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                    _lib.ptr(running_mean), _lib.ptr(running_var), mean.data_ptr(), var.data_ptr(), _lib.ptr(w),
                    float(momentum), ctx.eps, relu, 0.0, n, ws.data_ptr(), st), "skd_abn_forward_train_nhwc_sync")
                comm_timer.end(tok)
            else:
                geo.stats(lib, x, mean, var, ws, st)
                mean, var = _sync_stats(stat, c, geo.count, ctx.group, running_mean, running_var, momentum, lib, st)
                geo.apply_to(lib, x, residual, out, mean, var, weight, bias, ctx.eps, relu, 0.0, st)
        ctx.has_residual = residual is not None
        # without a residual the ReLU mask is a function of x alone: the channels-last backward recomputes it instead of
        # reading `out` (which stays alive anyway as the next layer's input, but is not touched again here)
        ctx.mask_from_x = geo.nhwc and residual is None        # (A/B of round 2: profiles/r02h, 67.8 -> 66.2 ms per step)
        ctx.save_for_backward(x, None if ctx.mask_from_x else out, weight, bias, mean, var)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, out, weight, bias, mean, var = ctx.saved_tensors
        need_dx, need_dw, need_db = ctx.needs_input_grad[0:3]
        need_res = ctx.has_residual and ctx.needs_input_grad[5]
        geo = _Geom(x)
        c = geo.c
        dout = geo.like(dout)
        if not geo.nhwc and not _same_phase(x, dout):
            dout = dout.clone(memory_format=torch.contiguous_format)
        lib, st = _lib.get(), _lib.stream_of(x)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if need_res else None
        dweight = geo.new_param_grad(weight) if (need_dw and weight is not None) else None
        dbias = geo.new_param_grad(weight) if (need_db and weight is not None) else None
        stat = x.new_empty((2, c))
        edz, eydz = stat[0], stat[1]
        ws = geo.workspace(lib, x)
        if ctx.group is None and geo.nhwc:
            _lib.check(lib.skd_abn_relu_backward_nhwc(geo.rows, c, x.data_ptr(), _lib.ptr(out), dout.data_ptr(), mean.data_ptr(),
                                                      var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), edz.data_ptr(),
                                                      eydz.data_ptr(), dx.data_ptr(), _lib.ptr(dres), _lib.ptr(dweight),
                                                      _lib.ptr(dbias), ctx.eps, 0, ws.data_ptr(), st),
                       "skd_abn_relu_backward_nhwc")
            return (dx if need_dx else None), dweight, dbias, None, None, dres, None, None, None
        mb = _mailbox_for(geo, ctx.group, x) if ctx.group is not None else None
        if mb is not None:
            from ..utils.parallel import comm_timer
            tok = comm_timer.begin("syncabn_fused", x)
            _lib.check(lib.skd_abn_relu_backward_nhwc_sync(
                mb.ctx, geo.rows, c, x.data_ptr(), _lib.ptr(out), dout.data_ptr(), mean.data_ptr(), var.data_ptr(), _lib.ptr(weight),
                _lib.ptr(bias), edz.data_ptr(), eydz.data_ptr(), dx.data_ptr(), _lib.ptr(dres), _lib.ptr(dweight), _lib.ptr(dbias),
                _lib.ptr(_replica_weights(ctx.group)), ctx.eps, 0, ws.data_ptr(), st), "skd_abn_relu_backward_nhwc_sync")
            comm_timer.end(tok)
            return (dx if need_dx else None), dweight, dbias, None, None, dres, None, None, None
        geo.relu_backward_reduce(lib, x, out, dout, mean, var, edz, eydz, ctx.eps, ws, st, weight=weight, bias=bias)
        if ctx.group is not None:
            _sync_grad_stats(stat, ctx.group)
        geo.relu_backward_dx(lib, x, out, dout, mean, var, weight, edz, eydz, dx, dres, dweight, dbias, ctx.eps, st, bias=bias)
        return (dx if need_dx else None), dweight, dbias, None, None, dres, None, None, None


def _nhwc_unsupported(x):
    """Channels-last tensor whose channel count the channels-last kernels do not take (power of two in [4, 1024])."""
    if not _is_nhwc(x):
        return False
    c = x.shape[1]
    return c < 4 or c > 1024 or bool(c & (c - 1))


_warned_via_nchw = False


def _via_nchw(fn, x, *rest, residual=None, inplace=False):
    """Run ``fn`` on an NCHW copy and hand the result back channels-last.  The reference accepts any channel count
    (libs/functions.py:70-162); the channels-last kernels do not, so odd widths (and the teacher's 2048-channel layers
    when somebody trains or differentiates it) take the NCHW kernels through two layout copies instead of failing
    (a RuntimeWarning says so once).  ``inplace``: the caller's contract is "mutates and returns x" (inplace_abn /
    inplace_abn_sync, libs/functions.py:100-108): the result is copied back INTO x (a differentiable copy_, so autograd
    still reaches ``fn``'s backward) and x is returned, for callers that ignore the return value."""
    global _warned_via_nchw
    if not _warned_via_nchw:
        _warned_via_nchw = True
        import warnings
        warnings.warn("InPlace-ABN on a channels-last tensor with %d channels (not a power of two in [4, 1024]): NCHW kernels "
                      "through two layout copies" % x.shape[1], RuntimeWarning)
    if residual is not None:
        residual = residual.contiguous()
    out = fn(x.contiguous(), *rest) if residual is None else fn(x.contiguous(), *rest, residual)
    if inplace:
        x.copy_(out)
        return x
    return out.contiguous(memory_format=torch.channels_last)


def abn_relu_train(x, weight, bias, running_mean, running_var, residual=None, momentum=0.1, eps=1e-05, group=None,
                   sync=True):
    """relu(bn_batch(x) [+ residual]) with running-statistics update: the training-time fusion of
    InPlace-ABN(activation='none') -> [+ residual] -> ReLU.  Differentiable (first order) w.r.t. x, weight,
    bias and residual.  ``sync``: synchronise the statistics over ``group`` (default group when None and
    torch.distributed is initialised) like InPlaceABNSync; False = this replica only (InPlaceABN)."""
    if not sync:
        group = None
    elif group is None:
        group = _default_group()
    if _nhwc_unsupported(x):
        return _via_nchw(lambda xc, r=None: _ABNRelu.apply(xc, weight, bias, running_mean, running_var, r, momentum, eps, group),
                         x, residual=residual)
    return _ABNRelu.apply(x, weight, bias, running_mean, running_var, residual, momentum, eps, group)


class _ABNReluMaxPool(autograd.Function):
    """Training-time ``maxpool3x3s2(relu(bn_batch(x)))`` of the student's stem (networks/pspnet_combine.py:176-180: bn3 -> relu3 ->
    maxpool on the (B, 128, 256, 256) conv3 output) WITHOUT the normalised tensor (csrc/abn.hip, "student stem"): forward =
    statistics + ONE kernel that normalises, rectifies and pools on the fly (68 MB written instead of 268 + 68, argmax as one byte
    per element); backward = the edz / eydz reduction and the dx pass, both gathering the pooled gradient through the argmax bytes
    and recomputing the ReLU mask from x.  Values, indices and gradients are those of abn_relu_train followed by the stem pool
    (bit for bit in the forward; the backward's reductions add in another order).  Channels-last fp32 only; cross-replica
    statistics exactly as in _ABNRelu's three-launch form (statistics -> exchange -> apply / reduce -> exchange -> dx)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, oh, ow, momentum, eps, group):
        _lib.require_device(x, weight, bias, running_mean, running_var)
        if x.dtype != torch.float32 or not _is_nhwc(x):
            raise TypeError("abn_relu_maxpool_train: channels-last fp32 tensors only")
        _check_contiguous(weight, bias, running_mean, running_var)
        ctx.eps = float(eps)
        ctx.group = group if _replicated(group) else None
        geo = _Geom(x)
        c = geo.c
        b, _, h, w = x.shape
        lib, st = _lib.get(), _lib.stream_of(x)
        stat = x.new_empty((2, c))
        mean, var = stat[0], stat[1]
        ws = geo.workspace(lib, x)
        geo.stats(lib, x, mean, var, ws, st)
        if ctx.group is None:
            if running_mean is not None and running_var is not None:
                _lib.check(lib.skd_abn_update_running(c, running_mean.data_ptr(), running_var.data_ptr(), mean.data_ptr(),
                                                      var.data_ptr(), float(momentum), float(geo.count), st), "skd_abn_update_running")
        else:
            mean, var = _sync_stats(stat, c, geo.count, ctx.group, running_mean, running_var, momentum, lib, st)
        pooled = x.new_empty((b, oh, ow, c)).permute(0, 3, 1, 2)          # channels-last memory
        arg = torch.empty((b, oh, ow, c), dtype=torch.uint8, device=x.device)
        _lib.check(lib.skd_abn_relu_maxpool3x3s2_nhwc(b, c, h, w, oh, ow, x.data_ptr(), mean.data_ptr(), var.data_ptr(), _lib.ptr(weight),
                                                      _lib.ptr(bias), ctx.eps, pooled.data_ptr(), arg.data_ptr(), st),
                   "skd_abn_relu_maxpool3x3s2_nhwc")
        ctx.geom = (b, c, h, w, oh, ow)
        ctx.save_for_backward(x, arg, weight, bias, mean, var)
        return pooled

    @staticmethod
    @once_differentiable
    def backward(ctx, dpooled):
        x, arg, weight, bias, mean, var = ctx.saved_tensors
        b, c, h, w, oh, ow = ctx.geom
        need_dx, need_dw, need_db = ctx.needs_input_grad[0:3]
        dpooled = dpooled if _is_nhwc(dpooled) else dpooled.contiguous(memory_format=torch.channels_last)
        lib, st = _lib.get(), _lib.stream_of(x)
        geo = _Geom(x)
        stat = x.new_empty((2, c))
        edz, eydz = stat[0], stat[1]
        ws = geo.workspace(lib, x)
        _lib.check(lib.skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc(
            b, c, h, w, oh, ow, x.data_ptr(), dpooled.data_ptr(), arg.data_ptr(), mean.data_ptr(), var.data_ptr(), _lib.ptr(weight),
            _lib.ptr(bias), edz.data_ptr(), eydz.data_ptr(), ctx.eps, ws.data_ptr(), st), "skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc")
        if ctx.group is not None:
            _sync_grad_stats(stat, ctx.group)
        dx = torch.empty_like(x)
        dweight = torch.empty_like(weight) if (need_dw and weight is not None) else None
        dbias = torch.empty_like(weight) if (need_db and weight is not None) else None
        _lib.check(lib.skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(
            b, c, h, w, oh, ow, x.data_ptr(), dpooled.data_ptr(), arg.data_ptr(), mean.data_ptr(), var.data_ptr(), _lib.ptr(weight),
            _lib.ptr(bias), edz.data_ptr(), eydz.data_ptr(), dx.data_ptr(), _lib.ptr(dweight), _lib.ptr(dbias), ctx.eps, 0, st),
            "skd_abn_relu_maxpool3x3s2_backward_dx_nhwc")
        return (dx if need_dx else None), dweight, dbias, None, None, None, None, None, None, None


def abn_relu_maxpool_supported(x):
    """channels-last fp32 tensor with a channel count the channels-last kernels take (a power of two in [4, 1024])."""
    return x.dtype == torch.float32 and _is_nhwc(x) and not _nhwc_unsupported(x) and (x.is_cuda or _lib.test_backend_active())


def abn_relu_maxpool_train(x, weight, bias, running_mean, running_var, oh, ow, momentum=0.1, eps=1e-05, group=None, sync=True):
    """maxpool3x3s2(relu(bn_batch(x))), (oh, ow) = the pooled size (torch's rule for kernel 3, stride 2, padding 1, ceil_mode as the
    caller's pool has it), with the running-statistics update: the training stem in two fused passes per direction."""
    if not sync:
        group = None
    elif group is None:
        group = _default_group()
    return _ABNReluMaxPool.apply(x, weight, bias, running_mean, running_var, int(oh), int(ow), momentum, eps, group)


def abn_eval_fused(x, weight, bias, running_mean, running_var, eps=1e-05, activation=ACT_RELU, slope=0.01,
                   residual=None):
    """Inference-only InPlace-ABN: ``x <- act(bn_running(x) [+ residual])`` in ONE in-place pass.

    Fuses what networks/pspnet_combine.py runs as separate ops around an eval-mode BatchNorm2d
    (= InPlaceABNSync(activation='none'), :12): the following nn.ReLU (:36, :68, :72) and, at the
    tail of a residual block, ``out + residual`` then ReLU (:41-43, :80-82).  No autograd graph is
    recorded: callers use it under ``torch.no_grad()`` with the module in eval mode (the frozen
    teacher, kd_model.py:121-122)."""
    if torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad)):
        raise RuntimeError("abn_eval_fused is inference-only; use inplace_abn for differentiable calls")
    _lib.require_device(x, weight, bias, running_mean, running_var, residual)
    if x.dtype != torch.float32:
        raise TypeError("InPlaceABN kernels are fp32 only (got %s)" % x.dtype)
    try:
        act = _EVAL_ACT_CODE[activation]
    except KeyError:
        raise ValueError("unknown activation %r" % (activation,))
    if x.numel() == 0:
        return x
    if _is_nhwc(x) and x.shape[1] % 4 != 0:       # channels-last with an odd channel count: NCHW kernels on a copy
        xc = abn_eval_fused(x.contiguous(), weight, bias, running_mean, running_var, eps, activation, slope,
                            None if residual is None else residual.contiguous())
        return x.copy_(xc)
    lib, st = _lib.get(), _lib.stream_of(x)
    if (x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
            and x.shape[1] % 4 == 0):
        # channels-last (NHWC) activations: the frozen teacher's MIOpen convolutions run NHWC-native
        _check_contiguous(weight, bias, running_mean, running_var)
        if residual is not None:
            if residual.shape != x.shape:
                raise ValueError("residual shape %s != input shape %s" % (tuple(residual.shape), tuple(x.shape)))
            if not residual.is_contiguous(memory_format=torch.channels_last) or residual.is_contiguous():
                residual = residual.contiguous(memory_format=torch.channels_last)
        rows = x.shape[0] * x.shape[2] * x.shape[3]
        _lib.check(lib.skd_abn_apply_nhwc(rows, x.shape[1], x.data_ptr(), _lib.ptr(residual), running_mean.data_ptr(),
                                          running_var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), float(eps), act,
                                          float(slope), st), "skd_abn_apply_nhwc")
        return x
    _check_contiguous(x, weight, bias, running_mean, running_var)
    n, c, s = _dims(x)
    if residual is None:
        _lib.check(lib.skd_abn_apply(n, c, s, x.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                                     _lib.ptr(weight), _lib.ptr(bias), float(eps), act, float(slope), st),
                   "skd_abn_apply")
    else:
        if residual.shape != x.shape:
            raise ValueError("residual shape %s != input shape %s" % (tuple(residual.shape), tuple(x.shape)))
        if not residual.is_contiguous() or not _same_phase(x, residual):
            residual = residual.clone(memory_format=torch.contiguous_format)
        _lib.check(lib.skd_abn_apply_residual(n, c, s, x.data_ptr(), residual.data_ptr(), running_mean.data_ptr(),
                                              running_var.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), float(eps),
                                              act, float(slope), st), "skd_abn_apply_residual")
    return x


def inplace_abn(x, weight, bias, running_mean, running_var, training=True, momentum=0.1, eps=1e-05,
                activation=ACT_LEAKY_RELU, slope=0.01):
    """Signature of libs/functions.py:70-73 (InPlaceABN.apply)."""
    if _nhwc_unsupported(x):
        return _via_nchw(lambda xc: _InPlaceABN.apply(xc, weight, bias, running_mean, running_var, training, momentum, eps,
                                                      activation, slope, None), x, inplace=True)
    return _InPlaceABN.apply(x, weight, bias, running_mean, running_var, training, momentum, eps,
                             activation, slope, None)


def inplace_abn_sync(x, weight, bias, running_mean, running_var, extra=None, training=True,
                     momentum=0.1, eps=1e-05, activation=ACT_LEAKY_RELU, slope=0.01):
    """Signature of libs/functions.py:165-168 (InPlaceABNSync.apply).

    ``extra`` carried the master/worker queues in the reference; here it is either None
    (default process group when torch.distributed is initialised with world_size > 1, otherwise
    no synchronisation) or a dict {"group": ProcessGroup}.
    """
    group = None
    if isinstance(extra, dict):
        group = extra.get("group")
    elif extra is not None:
        group = extra
    if group is None:
        group = _default_group()
    if _nhwc_unsupported(x):
        return _via_nchw(lambda xc: _InPlaceABN.apply(xc, weight, bias, running_mean, running_var, training, momentum, eps,
                                                      activation, slope, group), x, inplace=True)
    return _InPlaceABN.apply(x, weight, bias, running_mean, running_var, training, momentum, eps,
                             activation, slope, group)
