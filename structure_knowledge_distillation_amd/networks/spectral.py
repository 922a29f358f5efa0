"""Spectral normalisation wrapper backed by csrc/spectral.hip.

Same wrapper contract as the reference's networks/spectral.py:14-68: wraps a module, replaces its
``weight`` Parameter by ``weight_bar`` (learnable) plus ``weight_u`` / ``weight_v``
(``requires_grad=False`` Parameters, so they appear in the state dict under the same keys), and on
EVERY forward runs ``power_iterations`` power steps that overwrite u and v, then sets
``module.weight = weight_bar / sigma`` with a differentiable sigma = u^T W v.
The ~12 stock launches per layer of the reference become 3 kernel launches (functional.py).
"""
import torch
from torch import nn
from torch.nn import Parameter

from .. import functional as SF


def l2normalize(v, eps=1e-12):
    return v / (v.norm() + eps)


class SpectralNorm(nn.Module):
    def __init__(self, module, name="weight", power_iterations=1):
        super().__init__()
        self.module = module
        self.name = name
        self.power_iterations = power_iterations
        if not self._made_params():
            self._make_params()

    def _update_u_v(self):
        u = getattr(self.module, self.name + "_u")
        v = getattr(self.module, self.name + "_v")
        w = getattr(self.module, self.name + "_bar")
        for _ in range(self.power_iterations - 1):
            SF.spectral_power_iteration(w, u.data, v.data)
        setattr(self.module, self.name, SF.spectral_normalize(w, u.data, v.data))

    def _made_params(self):
        return all(hasattr(self.module, self.name + s) for s in ("_u", "_v", "_bar"))

    def _make_params(self):
        w = getattr(self.module, self.name)
        height = w.data.shape[0]
        width = w.view(height, -1).data.shape[1]
        u = Parameter(w.data.new(height).normal_(0, 1), requires_grad=False)
        v = Parameter(w.data.new(width).normal_(0, 1), requires_grad=False)
        u.data = l2normalize(u.data)
        v.data = l2normalize(v.data)
        w_bar = Parameter(w.data)
        del self.module._parameters[self.name]
        self.module.register_parameter(self.name + "_u", u)
        self.module.register_parameter(self.name + "_v", v)
        self.module.register_parameter(self.name + "_bar", w_bar)

    def forward(self, *args):
        self._update_u_v()
        return self.module.forward(*args)
