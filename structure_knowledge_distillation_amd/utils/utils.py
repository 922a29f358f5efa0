"""Hot-path helpers of the reference's utils/utils.py: the similarity math (:170-183) on the
gfx950 pair-wise kernels, and the tuple-string builders NetModel evaluates (:12-38).  Logging /
checkpoint-loading helpers of that file are host-side I/O outside the hot path; the small ones
NetModel calls are provided in plain Python."""
import logging

import torch

from .. import functional as SF


def to_tuple_str(str_first, gpu_num, str_ind):
    """utils.py:12-21 (kept verbatim in behaviour: NetModel eval()s the result)."""
    if gpu_num > 1:
        parts = ["(%s[%d]%s,)" % (str_first, i, str_ind) for i in range(gpu_num)]
        return "(" + ", ".join(parts) + ")"
    return str_first + str_ind


def to_cat_str(str_first, gpu_num, str_ind, dim_):
    if gpu_num > 1:
        parts = ["%s[%d]%s" % (str_first, i, str_ind) for i in range(gpu_num)]
        return "torch.cat((" + ", ".join(parts) + "), dim=" + str(dim_) + ")"
    return str_first + str_ind


def to_tuple(list_data, gpu_num, sec_ind):
    return tuple(list_data[i][sec_ind] for i in range(gpu_num))


def L2(f_):
    """Channel L2 norm + 1e-8, shape (B,1,H,W) (utils.py:170-171).  Stock ops; the fused path
    (sim_dis_compute) never materialises it."""
    return (((f_ ** 2).sum(dim=1)) ** 0.5).reshape(f_.shape[0], 1, f_.shape[2], f_.shape[3]) + 1e-8


def similarity(feat):
    """(B, M, M) normalised Gram matrix (utils.py:173-178), for callers that want the matrix itself.
    The loss path below does not call this: it fuses normalise + both Grams + the squared error."""
    feat = feat.float()
    feat = feat / L2(feat).detach()
    feat = feat.reshape(feat.shape[0], feat.shape[1], -1)
    return torch.bmm(feat.transpose(1, 2), feat)


def sim_dis_compute(f_S, f_T):
    """sum((sim(f_T) - sim(f_S))^2) / (h*w)^2 / B (utils.py:180-183) in one fused kernel chain."""
    return SF.sim_dis(f_S.float(), f_T.float())


def print_model_parm_nums(model, string):
    n = sum(p.numel() for p in model.parameters())
    logging.info(string + ": Number of params: %.2fM", n / 1e6)
    return n
