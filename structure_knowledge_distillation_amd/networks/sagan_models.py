"""Holistic (SAGAN) discriminator, networks/sagan_models.py:9-41 (Self_Attn) and :105-168
(Discriminator) of the reference; same constructor, module names and state-dict keys (37 tensors).
Spectral-normalised convolutions use the gfx950 kernel through networks/spectral.py; the
convolutions and the two tiny attention bmm's run on MIOpen / rocBLAS (stock ops, so WGAN-GP's
double backward through them keeps working).  The unused ``Generator`` is out of scope."""
import torch
import torch.nn as nn

from .spectral import SpectralNorm, normalize_together


class Self_Attn(nn.Module):
    """out = gamma * (V softmax(Q^T K)^T) + x, attention returned as (B, N, N)."""

    def __init__(self, in_dim, activation):
        super().__init__()
        self.chanel_in = in_dim
        self.activation = activation
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, 1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        b, c, width, height = x.size()
        n = width * height
        q = self.query_conv(x).view(b, -1, n).permute(0, 2, 1)
        k = self.key_conv(x).view(b, -1, n)
        attention = self.softmax(torch.bmm(q, k))
        v = self.value_conv(x).view(b, -1, n)
        out = torch.bmm(v, attention.permute(0, 2, 1)).view(b, c, width, height)
        return self.gamma * out + x, attention


class NativeBatchNorm2d(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` evaluated by PyTorch's own kernels instead of MIOpen's batch-norm.

    Same parameters, buffers and state-dict keys.  Measured on MI355X (tests/diagnostics/diag_dfwd.py): MIOpen's
    training-mode batch-norm is off by 4.6e-4 relative on the discriminator's (B, 19, 65, 65) input -- enough
    to break the 1e-4 loss tolerance and to put 1-3 % of error on every D gradient -- while the native kernel
    agrees with the fp64 reference to 5e-8.  It also keeps WGAN-GP's double backward on PyTorch's formulas."""

    def forward(self, x):
        with torch.backends.cudnn.flags(enabled=False):
            return super().forward(x)


class Discriminator(nn.Module):
    def __init__(self, preprocess_GAN_mode, input_channel, batch_size=64, image_size=64, conv_dim=64):
        super().__init__()
        self.imsize = image_size

        def sn_block(cin, cout):
            return nn.Sequential(SpectralNorm(nn.Conv2d(cin, cout, 4, 2, 1)), nn.LeakyReLU(0.1))

        curr = conv_dim
        self.l1 = sn_block(input_channel, curr)
        self.l2 = sn_block(curr, curr * 2)
        curr *= 2
        self.l3 = sn_block(curr, curr * 2)
        curr *= 2
        if self.imsize == 65:                        # sagan_models.py:131-136
            self.l4 = sn_block(curr, curr * 2)
            curr *= 2
        self.last = nn.Sequential(nn.Conv2d(curr, 1, 4))
        self.attn1 = Self_Attn(256, "relu")
        self.attn2 = Self_Attn(512, "relu")
        if preprocess_GAN_mode == 1:
            self.preprocess_additional = NativeBatchNorm2d(input_channel)     # nn.BatchNorm2d, sagan_models.py:148
        elif preprocess_GAN_mode == 2:
            self.preprocess_additional = nn.Tanh()
        elif preprocess_GAN_mode == 3:
            self.preprocess_additional = lambda x: 2 * (x / 255 - 0.5)
        else:
            raise ValueError("preprocess_GAN_mode should be 1:bn or 2:tanh or 3:-1 - 1")

    def forward(self, x):
        # the four spectrally normalised weights of this forward in 3 launches instead of 12 (csrc/spectral.hip, "several layers
        # per launch"; bit-identical to one wrapper at a time: tests/test_kernels_gpu.py::test_spectral_norm_multi_*, step-neutral A/B
        # in profiles/r04g_bench_ab_SN_TOGETHER_0.json)
        if not torch.are_deterministic_algorithms_enabled():      # SKD_DETERMINISTIC=1 keeps round 3's one-wrapper-at-a-time order
            normalize_together([blk[0] for blk in (self.l1, self.l2, self.l3, getattr(self, "l4", None))
                                if blk is not None and isinstance(blk[0], SpectralNorm)])
        x = self.preprocess_additional(x)
        out = self.l3(self.l2(self.l1(x)))
        out, p1 = self.attn1(out)
        out = self.l4(out)
        out, p2 = self.attn2(out)
        return [self.last(out), p1, p2]
