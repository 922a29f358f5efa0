"""Kink-aware comparison of the critic (test infrastructure, shared by tests/test_step_gpu.py and tests/test_distributed_gpu.py).

The holistic loss differentiates through the discriminator's LeakyReLU slopes (networks/sagan_models.py:116-134 of the reference;
the gradient penalty of utils/criterion.py:98-120 even differentiates the derivative), i.e. through step functions of the
pre-activations.  An fp32 evaluation (the product, on the GPU) and the fp64 oracle of the SAME logits disagree about the sign of a
pre-activation that lies within rounding of zero; one such unit of the first layer moves a gradient by ~1 / sqrt(units) = 4e-3.
Round 4 answered that by widening bounds (2e-4 -> 3e-2, 1e-3 -> 5e-3).  This module restores the tight bounds instead:

  * ``LeakyRecorder`` hooks every ``nn.LeakyReLU`` of the product's Discriminator and records the decisions (output > 0) it
    actually took, per call, in call order;
  * the oracle (oracle/step_torch.py: ``LeakyMasks``) is then evaluated ON THE SAME LINEAR PIECE -- the recorded decisions replace
    its own signs -- and reports how many units it had to override and how far from the kink they were;
  * ``assert_only_rounding_flips`` holds those to "a handful, all within rounding distance of zero": a product that took a wrong
    branch anywhere else fails here, a product whose arithmetic is wrong fails the (tight again) gradient bound.
"""
import torch

KINK_TAU = 1e-5          # an overridden unit's |pre-activation| must be below this fraction of its layer's rms pre-activation
                         # (fp32 convolutions over K = 304 ... 4096 terms land within ~1e-6 of the fp64 value; six hardware runs of
                         # rounds 5-6 saw overrides at <= 3e-7 of the rms: 30 x margin.  Round 5 shipped 1e-4)
KINK_FLIPS_ABS, KINK_FLIPS_REL = 4, 2e-6      # at most 4 + 2e-6 x (units evaluated) overrides = 11 of 3.9 M (measured: 0-2)


class LeakyRecorder:
    """with LeakyRecorder(D) as rec: ...   ->   rec.masks: [bool tensor per LeakyReLU call, in call order] (on the CPU)."""

    def __init__(self, module):
        self.module, self._dev, self._handles = module, [], []

    def __enter__(self):
        for m in self.module.modules():
            if isinstance(m, torch.nn.LeakyReLU):
                self._handles.append(m.register_forward_hook(lambda mod, inp, out: self._dev.append(out.detach() > 0)))
        return self

    def __exit__(self, *exc):
        for h in self._handles:
            h.remove()
        self._handles = []
        return False

    @property
    def masks(self):
        return [m.cpu() for m in self._dev]


def assert_only_rounding_flips(lm, what):
    """``lm``: an oracle LeakyMasks after its evaluation."""
    limit = KINK_FLIPS_ABS + KINK_FLIPS_REL * lm.units
    print("%s: %d of %d LeakyReLU decisions taken from the evaluation under test (limit %d), farthest from the kink: %.1e of the "
          "layer's rms pre-activation (limit %.0e)" % (what, lm.flipped, lm.units, limit, lm.worst, KINK_TAU))
    assert lm.flipped <= limit, (what, lm.flipped, lm.units)
    assert lm.worst <= KINK_TAU, (what, lm.worst)
