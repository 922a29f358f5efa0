"""Drop-in for the reference's ``libs`` package (libs/__init__.py:1): the InPlace-ABN modules
used by networks/pspnet_combine.py:11, backed by the gfx950 kernels in csrc/abn.hip."""
from .inplace_abn import abn_eval_fused, abn_relu_train, inplace_abn, inplace_abn_sync, ACT_LEAKY_RELU, ACT_ELU, ACT_NONE, ACT_RELU
from .modules import ABN, InPlaceABN, InPlaceABNSync, InPlaceABNWrapper, InPlaceABNSyncWrapper
from .modules import set_sync_group, get_sync_group

__all__ = ["ABN", "InPlaceABN", "InPlaceABNSync", "InPlaceABNWrapper", "InPlaceABNSyncWrapper",
           "inplace_abn", "inplace_abn_sync", "abn_eval_fused", "abn_relu_train", "set_sync_group", "get_sync_group"]
