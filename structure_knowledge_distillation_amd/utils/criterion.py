"""The distillation criteria of the reference's utils/criterion.py, same class names, constructor
arguments and list-indexing convention (``preds[0]`` logits, ``preds[1]`` DSN logits, ``preds[-5]``
post-PSP feature), each returning a 0-dim tensor that participates in autograd.

    CriterionDSN                              :168-188   fused HIP kernels (csrc/ce_dsn.hip): upsample + CE, main + 0.4*aux
    CriterionPixelWise                        :211-226   fused HIP kernel (csrc/pixelwise.hip)
    CriterionPairWiseforWholeFeatAfterPool    :228-245   fused HIP kernels (csrc/pairwise.hip)
    CriterionAdvForG / CriterionAdv           :122-166   wgan-gp / hinge on D's (B,1,1,1) output
    CriterionAdditionalGP                     :92-120    WGAN-GP gradient penalty (double backward in D)
The OHEM variants (:11-90, :190-209) are never constructed on this path (kd_model.py:79) -- out of scope.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as SF
from .utils import sim_dis_compute


class CriterionDSN(nn.Module):
    def __init__(self, ignore_index=255, use_weight=True, reduce=True):
        super().__init__()
        self.ignore_index = ignore_index
        self.reduction = "mean" if reduce else "none"
        if not reduce:
            print("disabled the reduce.")

    def forward(self, preds, target):
        if self.reduction == "mean" and preds[0].shape[1] <= 64 and preds[0].dtype == torch.float32:
            # fused upsample + CE for both heads (csrc/ce_dsn.hip): nothing of size (B, C, H, W) is written
            return SF.cross_entropy_dsn(preds[0], preds[1], target, self.ignore_index, 0.4)
        h, w = target.size(1), target.size(2)     # reduce=False / very wide class counts: stock ops
        up = F.interpolate(preds[0], size=(h, w), mode="bilinear", align_corners=True)
        loss1 = F.cross_entropy(up, target, ignore_index=self.ignore_index, reduction=self.reduction)
        up = F.interpolate(preds[1], size=(h, w), mode="bilinear", align_corners=True)
        loss2 = F.cross_entropy(up, target, ignore_index=self.ignore_index, reduction=self.reduction)
        return loss1 + loss2 * 0.4


class CriterionPixelWise(nn.Module):
    def __init__(self, ignore_index=255, use_weight=True, reduce=True):
        super().__init__()
        self.ignore_index = ignore_index
        if not reduce:
            print("disabled the reduce.")

    def forward(self, preds_S, preds_T):
        assert preds_S[0].shape == preds_T[0].shape, "the output dim of teacher and student differ"
        return SF.pixel_wise_loss(preds_S[0], preds_T[0])


class CriterionPairWiseforWholeFeatAfterPool(nn.Module):
    def __init__(self, scale, feat_ind):
        """inter pair-wise loss from inter feature maps"""
        super().__init__()
        self.criterion = sim_dis_compute
        self.feat_ind = feat_ind
        self.scale = scale

    def forward(self, preds_S, preds_T):
        feat_S = preds_S[self.feat_ind]
        feat_T = preds_T[self.feat_ind]
        total_w, total_h = feat_T.shape[2], feat_T.shape[3]
        patch_w, patch_h = int(total_w * self.scale), int(total_h * self.scale)
        return SF.pair_wise_loss(feat_S.float(), feat_T.float(), patch_w, patch_h)


def _check_adv_type(adv_type):
    if adv_type != "wgan-gp" and adv_type != "hinge":
        raise ValueError("adv_type should be wgan-gp or hinge")


class CriterionAdvForG(nn.Module):
    def __init__(self, adv_type):
        super().__init__()
        _check_adv_type(adv_type)
        self.adv_loss = adv_type

    def forward(self, d_out_S, d_out_S_no_use=None):
        return -d_out_S[0].mean()          # identical for wgan-gp and hinge (criterion.py:131-134)


class CriterionAdv(nn.Module):
    def __init__(self, adv_type):
        super().__init__()
        _check_adv_type(adv_type)
        self.adv_loss = adv_type

    def forward(self, d_out_S, d_out_T):
        assert d_out_S[0].shape == d_out_T[0].shape, "the output dim of D with teacher and student as input differ"
        real, fake = d_out_T[0], d_out_S[0]
        if self.adv_loss == "wgan-gp":
            return -torch.mean(real) + fake.mean()
        return F.relu(1.0 - real).mean() + F.relu(1.0 + fake).mean()


class CriterionAdditionalGP(nn.Module):
    def __init__(self, D_net, lambda_gp):
        super().__init__()
        self.D = D_net
        self.lambda_gp = lambda_gp

    def forward(self, d_in_S, d_in_T, alpha=None):
        assert d_in_S[0].shape == d_in_T[0].shape, "the output dim of D with teacher and student as input differ"
        real, fake = d_in_T[0].detach(), d_in_S[0].detach()
        if alpha is None:   # criterion.py:104: one uniform sample per image
            alpha = torch.rand(real.size(0), 1, 1, 1, device=real.device, dtype=real.dtype)
        interpolated = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
        out = self.D(interpolated)
        grad = torch.autograd.grad(outputs=out[0], inputs=interpolated, grad_outputs=torch.ones_like(out[0]),
                                   retain_graph=True, create_graph=True, only_inputs=True)[0]
        grad = grad.view(grad.size(0), -1)
        grad_l2norm = torch.sqrt(torch.sum(grad ** 2, dim=1))
        return self.lambda_gp * torch.mean((grad_l2norm - 1) ** 2)
