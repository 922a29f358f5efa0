"""Spectral normalisation of a wrapped layer's weight, evaluated by csrc/spectral.hip.

Drop-in for the wrapper the reference's discriminator is built from (networks/spectral.py:14-68, used at
networks/sagan_models.py:117-136).  What a caller can observe is kept:

  * ``SpectralNorm(layer, name="weight", power_iterations=1)``; the wrapped layer is reachable as ``.module``;
  * the layer loses its ``weight`` Parameter and gains three: ``weight_bar`` (trainable, the raw weight) and
    ``weight_u`` / ``weight_v`` (``requires_grad=False``), so state dicts carry ``module.weight_bar/_u/_v`` exactly
    like checkpoints written by the reference; u is drawn before v, both N(0, 1) then scaled to unit length, so a
    seeded construction yields the same vectors;
  * EVERY forward -- training or not, graph or no graph -- advances u and v by ``power_iterations`` power steps
    (written through ``.data``: no autograd version bump, which is what lets the D step of kd_model.py:156-164 run three
    forwards before one backward) and then installs ``weight = weight_bar / sigma`` on the layer as a plain tensor,
    sigma = u . (W v) differentiable with respect to ``weight_bar`` only.

How it is computed differs: the reference issues ~12 small torch ops per layer and forward; here the last power step,
sigma and the division are three launches of one fused kernel family with a hand-written backward
(``functional.spectral_normalize``), earlier power steps (``power_iterations > 1``) use its u / v-only entry.
"""
import torch
from torch import nn

from .. import functional as SF

_SUFFIXES = ("_u", "_v", "_bar")


def l2normalize(v, eps=1e-12):
    """v / (|v| + eps) (spectral.py:10-11); exported because callers of the reference module import it."""
    return v / (v.norm() + eps)


def _unit_gaussian(like, n):
    """A fresh N(0, 1) vector of length n on ``like``'s device / dtype, scaled to unit length, as a frozen Parameter."""
    p = nn.Parameter(like.new_empty(n).normal_(0, 1), requires_grad=False)
    p.data = l2normalize(p.data)
    return p


class SpectralNorm(nn.Module):
    def __init__(self, module, name="weight", power_iterations=1):
        super().__init__()
        self.module = module
        self.name = name
        self.power_iterations = power_iterations
        if not all(hasattr(module, name + s) for s in _SUFFIXES):     # wrapping an already-wrapped layer is a no-op
            self._split_weight()

    def _split_weight(self):
        """weight -> (weight_u, weight_v, weight_bar), registered in that order (the reference's state-dict order)."""
        layer, name = self.module, self.name
        raw = layer._parameters.pop(name)
        rows = raw.shape[0]
        cols = raw.numel() // rows
        layer.register_parameter(name + "_u", _unit_gaussian(raw.data, rows))
        layer.register_parameter(name + "_v", _unit_gaussian(raw.data, cols))
        layer.register_parameter(name + "_bar", nn.Parameter(raw.data))

    def _parts(self):
        layer, name = self.module, self.name
        return getattr(layer, name + "_bar"), getattr(layer, name + "_u"), getattr(layer, name + "_v")

    def forward(self, *args):
        if getattr(self, "_prepared", False):
            # normalize_together() has advanced u, v and installed the normalised weight for THIS forward already
            self._prepared = False
            return self.module.forward(*args)
        w_bar, u, v = self._parts()
        for _ in range(self.power_iterations - 1):
            SF.spectral_power_iteration(w_bar, u.data, v.data)
        # the final power step, sigma and w_bar / sigma in one fused op; u.data / v.data are overwritten in place
        setattr(self.module, self.name, SF.spectral_normalize(w_bar, u.data, v.data))
        return self.module.forward(*args)


def normalize_together(wrappers):
    """One power step + ``weight = weight_bar / sigma`` for SEVERAL ``SpectralNorm`` wrappers in three kernel launches
    (functional.spectral_normalize_multi) instead of three per wrapper; each wrapper's next ``forward`` then only runs its
    layer.  What a caller can observe is what ``SpectralNorm.forward`` does one wrapper at a time (spectral.py:23-35, 63-68):
    every wrapper's u / v advance exactly once per call, through ``.data``; the layers are independent, so doing their power
    steps side by side instead of interleaved with the convolutions changes no number.  Wrappers with more than one power
    iteration, CPU tensors without the test double, or a single wrapper keep the per-wrapper path (returns False)."""
    wrappers = list(wrappers)
    for w in wrappers:
        # a flag left over from a forward that raised between this function and the layer (OOM, interrupt) must not make the next
        # forward skip its power step and reuse a weight whose graph may be gone: every call starts clean and recomputes
        w._prepared = False
    if len(wrappers) < 2 or len(wrappers) > 8 or any(w.power_iterations != 1 for w in wrappers):
        return False
    parts = [w._parts() for w in wrappers]
    if len({(p[0].device, p[0].dtype) for p in parts}) != 1:
        return False
    weights = SF.spectral_normalize_multi([p[0] for p in parts], [p[1].data for p in parts], [p[2].data for p in parts])
    for w, t in zip(wrappers, weights):
        setattr(w.module, w.name, t)
        w._prepared = True
    return True
