"""nn.Module front-ends of InPlace-ABN -- same constructor signatures, parameter / buffer names
and state-dict keys (weight, bias, running_mean, running_var; no num_batches_tracked) as the
reference's libs/bn.py:24-215, so networks/pspnet_combine.py builds unchanged on top of them."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .inplace_abn import (abn_eval_fused, abn_relu_maxpool_supported, abn_relu_maxpool_train, abn_relu_train, inplace_abn,
                          inplace_abn_sync)

_sync_group = {"group": None, "explicit": False}


def set_sync_group(group):
    """Process group used by every InPlaceABNSync (None = default group when initialised)."""
    _sync_group["group"] = group
    _sync_group["explicit"] = True


def get_sync_group():
    return _sync_group["group"]


class ABN(nn.Sequential):
    """BatchNorm2d + activation as two stock modules (libs/bn.py:24-45)."""

    def __init__(self, num_features, activation=None, **kwargs):
        act = activation if activation is not None else nn.ReLU(inplace=True)
        super().__init__(OrderedDict([("bn", nn.BatchNorm2d(num_features, **kwargs)), ("act", act)]))


class _ABNBase(nn.Module):
    def __init__(self, num_features, eps, momentum, affine, activation, slope):
        super().__init__()
        self.num_features = num_features
        self.affine = affine
        self.eps = eps
        self.momentum = momentum
        self.activation = activation
        self.slope = slope
        if affine:
            self.weight = nn.Parameter(torch.empty(num_features))
            self.bias = nn.Parameter(torch.empty(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.reset_parameters()

    def reset_parameters(self):
        # libs/bn.py:86-91
        with torch.no_grad():
            self.running_mean.zero_()
            self.running_var.fill_(1)
            if self.affine:
                self.weight.fill_(1)
                self.bias.zero_()

    def fused_eval(self, x, activation="relu", residual=None):
        """Inference-only: ``x <- activation(bn(x) [+ residual])`` with the running statistics, one pass."""
        return abn_eval_fused(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                              activation, self.slope, residual)

    def forward_relu(self, x, residual=None):
        """``relu(self(x) [+ residual])`` for an ``activation='none'`` module -- BatchNorm2d followed by nn.ReLU in
        networks/pspnet_combine.py -- as one fused op: training -> abn_relu_train, inference (no grad) ->
        abn_eval_fused, anything else -> the plain op sequence."""
        if self.activation != "none":
            raise ValueError("forward_relu fuses BN(activation='none') + ReLU; this module has activation=%r" % self.activation)
        if self.training:
            sync = isinstance(self, InPlaceABNSync)
            group = _sync_group["group"] if (sync and _sync_group["explicit"]) else None
            return abn_relu_train(x, self.weight, self.bias, self.running_mean, self.running_var, residual,
                                  self.momentum, self.eps, group, sync=sync)
        if not torch.is_grad_enabled():
            return self.fused_eval(x, "relu", residual)
        out = self(x)
        return torch.relu(out if residual is None else out + residual)

    def forward_relu_maxpool(self, x, pool):
        """``pool(relu(self(x)))`` for an ``activation='none'`` module followed by nn.ReLU and the stem's
        nn.MaxPool2d(3, 2, 1) (networks/pspnet_combine.py:176-180).  Training on a channels-last fp32 tensor: two fused passes
        per direction that never write the normalised tensor (abn_relu_maxpool_train); anything else: forward_relu, then the pool."""
        from .. import functional as SF
        if (self.activation == "none" and self.training and abn_relu_maxpool_supported(x) and SF.is_stem_pool(pool)):
            sync = isinstance(self, InPlaceABNSync)
            group = _sync_group["group"] if (sync and _sync_group["explicit"]) else None
            oh, ow = SF.stem_pool_out(x.shape[2], pool.ceil_mode), SF.stem_pool_out(x.shape[3], pool.ceil_mode)
            return abn_relu_maxpool_train(x, self.weight, self.bias, self.running_mean, self.running_var, oh, ow, self.momentum,
                                          self.eps, group, sync=sync)
        return SF.max_pool_stem(self.forward_relu(x), pool)

    def extra_repr(self):
        rep = "{num_features}, eps={eps}, momentum={momentum}, affine={affine}, activation={activation}"
        if self.activation == "leaky_relu":
            rep += ", slope={slope}"
        return rep.format(**self.__dict__)


class InPlaceABN(_ABNBase):
    """libs/bn.py:48-105."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                 slope=0.01):
        super().__init__(num_features, eps, momentum, affine, activation, slope)

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            return self.fused_eval(x, self.activation)      # inference: no autograd node, NCHW or NHWC
        return inplace_abn(x, self.weight, self.bias, self.running_mean, self.running_var,
                           self.training, self.momentum, self.eps, self.activation, self.slope)


class InPlaceABNSync(_ABNBase):
    """libs/bn.py:108-193.  ``devices`` is accepted for signature compatibility; replicas are
    separate processes here (one per GPU), synchronised through torch.distributed (RCCL)."""

    def __init__(self, num_features, devices=None, eps=1e-5, momentum=0.1, affine=True,
                 activation="leaky_relu", slope=0.01):
        super().__init__(num_features, eps, momentum, affine, activation, slope)
        self.devices = list(devices) if devices else []

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            return self.fused_eval(x, self.activation)      # inference: no autograd node, NCHW or NHWC
        extra = {"group": _sync_group["group"]} if _sync_group["explicit"] else None
        return inplace_abn_sync(x, self.weight, self.bias, self.running_mean, self.running_var,
                                extra, self.training, self.momentum, self.eps, self.activation,
                                self.slope)


class InPlaceABNWrapper(nn.Module):
    """libs/bn.py:196-204."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.bn = InPlaceABN(*args, **kwargs)

    def forward(self, input):
        return self.bn(input)


class InPlaceABNSyncWrapper(nn.Module):
    """libs/bn.py:207-215."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.bn = InPlaceABNSync(*args, **kwargs)

    def forward(self, input):
        return self.bn(input)
