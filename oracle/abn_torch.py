"""CPU restatement of InPlace-ABN (fused BN + activation) in plain torch ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, formula by formula:
  * libs/src/bn.cu:125-138   mean / biased variance over (N, spatial)
  * libs/src/bn.cu:140-165   y = (x-mean)*invStd ; z = y*(|w|+eps) + b
  * libs/src/bn.cu:302-315   leaky-relu (x<0 -> x*slope)
  * libs/functions.py:81-91  n = N*S ; running_mean/var update with var*n/(n-1)
  * libs/functions.py:177,209  sync variant: n is multiplied by the number of replicas
  * libs/functions.py:196-197  sync combine: mean = means.mean(0); var = (vars+(mean-means)^2).mean(0)
  * libs/bn.py:69-91         parameters (weight=1, bias=0) and buffers (running_mean=0, running_var=1)

Two flavours are provided:
  * ``abn_autograd`` -- differentiable closed form; autograd supplies the backward.
  * ``abn_backward_formula`` -- the reference's hand-written backward
    (libs/functions.py:112-162 + bn.cu:167-232) restated op by op, so the two can be
    cross-checked against each other (tests/test_oracle_abn.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

ACT_LEAKY_RELU = "leaky_relu"
ACT_ELU = "elu"
ACT_NONE = "none"


def _bshape(x):
    return [1, x.shape[1]] + [1] * (x.dim() - 2)


def batch_stats(x):
    """bn.cu:125-138 -- two-pass mean and *biased* variance per channel."""
    dims = [d for d in range(x.dim()) if d != 1]
    mean = x.mean(dim=dims)
    var = ((x - mean.view(_bshape(x))) ** 2).mean(dim=dims)
    return mean, var


def count_samples(x):
    """functions.py:37-42."""
    n = 1
    for i, s in enumerate(x.shape):
        if i != 1:
            n *= s
    return n


def update_running(running_mean, running_var, mean, var, n, momentum):
    """functions.py:90-91 (n already includes the replica count in the sync variant)."""
    with torch.no_grad():
        running_mean.mul_(1 - momentum).add_(momentum * mean)
        running_var.mul_(1 - momentum).add_(momentum * var * n / (n - 1))


def combine_replica_stats(means, vars_):
    """functions.py:196-197. means, vars_: (G, C) per-replica statistics (equal counts assumed)."""
    mean = means.mean(0)
    var = (vars_ + (mean - means) ** 2).mean(0)
    return mean, var


def normalise_affine(x, mean, var, weight, bias, eps):
    """bn.cu:146-159."""
    shp = _bshape(x)
    inv_std = torch.where((var != 0) | torch.tensor(eps != 0), 1.0 / torch.sqrt(var + eps),
                          torch.zeros_like(var))
    y = (x - mean.view(shp)) * inv_std.view(shp)
    gamma = (weight.abs() + eps).view(shp) if weight is not None else 1.0
    beta = bias.view(shp) if bias is not None else 0.0
    return y * gamma + beta


def activate(z, activation, slope):
    if activation == ACT_LEAKY_RELU:
        return torch.where(z < 0, z * slope, z)          # bn.cu:302-315
    if activation == ACT_ELU:
        return torch.where(z < 0, torch.exp(z) - 1.0, z)  # bn.cu:333-346
    if activation == ACT_NONE:
        return z
    raise ValueError("unknown activation %r" % (activation,))


def abn_autograd(x, weight, bias, running_mean, running_var, training=True, momentum=0.1,
                 eps=1e-5, activation=ACT_LEAKY_RELU, slope=0.01, replicas=1, stats=None):
    """Differentiable restatement of libs/functions.py:70-110 (out of place).

    ``stats`` optionally supplies externally combined (mean, var) -- the sync path.
    """
    if training:
        if stats is None:
            mean, var = batch_stats(x)
        else:
            mean, var = stats
        n = count_samples(x) * replicas
        update_running(running_mean, running_var, mean.detach(), var.detach(), n, momentum)
    else:
        mean, var = running_mean, running_var
    z = normalise_affine(x, mean, var, weight, bias, eps)
    return activate(z, activation, slope)


def abn_backward_formula(z_out, dz, var, weight, bias, training, eps, activation, slope,
                         edz_eydz=None):
    """The reference's hand-written backward, out of place.

    z_out: forward output (post activation).  Returns (dx, dweight, dbias, edz, eydz).
    functions.py:112-162; kernels bn.cu:167-232, 317-331 (+ 302-315 with 1/slope as inverse).
    ``edz_eydz`` optionally supplies externally averaged (edz, eydz) -- the sync path
    (functions.py:263-280).
    """
    z = z_out.clone()
    dz = dz.clone()
    # _act_backward, functions.py:54-62
    if activation == ACT_LEAKY_RELU:
        dz = torch.where(z < 0, dz * slope, dz)
        z = torch.where(z < 0, z * (1.0 / slope), z)
    elif activation == ACT_ELU:
        dz = torch.where(z < 0, dz * (z + 1.0), dz)
        z = torch.where(z < 0, torch.log1p(z), z)
    shp = _bshape(z)
    C = z.shape[1]
    gamma = (weight.abs() + eps) if weight is not None else torch.ones(C, dtype=z.dtype)
    beta = bias if bias is not None else torch.zeros(C, dtype=z.dtype)
    y = (z - beta.view(shp)) / gamma.view(shp)
    dims = [d for d in range(z.dim()) if d != 1]
    if training:
        if edz_eydz is None:
            edz = dz.mean(dim=dims)               # bn.cu:175-177
            eydz = (y * dz).mean(dim=dims)
        else:
            edz, eydz = edz_eydz
    else:
        edz = torch.zeros(C, dtype=z.dtype)       # functions.py:146-147
        eydz = torch.zeros(C, dtype=z.dtype)
    inv_std = torch.where((var != 0) | torch.tensor(eps != 0), 1.0 / torch.sqrt(var + eps),
                          torch.zeros_like(var))
    mul = gamma * inv_std
    dx = (dz - edz.view(shp) - y * eydz.view(shp)) * mul.view(shp)   # bn.cu:203-210
    norm = float(count_samples(z))
    dweight = dbias = None
    if weight is not None:
        dweight = torch.sign(weight) * eydz * norm                   # bn.cu:217-223
    if bias is not None:
        dbias = edz * norm                                           # bn.cu:226-229
    return dx, dweight, dbias, edz, eydz


class InPlaceABN(nn.Module):
    """Stand-in for libs/bn.py:48-105 with the same ctor signature, params and buffers."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                 slope=0.01):
        super().__init__()
        self.num_features = num_features
        self.affine = affine
        self.eps = eps
        self.momentum = momentum
        self.activation = activation
        self.slope = slope
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x):
        return abn_autograd(x, self.weight, self.bias, self.running_mean, self.running_var,
                            self.training, self.momentum, self.eps, self.activation, self.slope)


class InPlaceABNSync(InPlaceABN):
    """Stand-in for libs/bn.py:108-193 on ONE replica (devices list of length 1)."""

    def __init__(self, num_features, devices=None, eps=1e-5, momentum=0.1, affine=True,
                 activation="leaky_relu", slope=0.01):
        super().__init__(num_features, eps, momentum, affine, activation, slope)
        self.devices = devices if devices else [0]
