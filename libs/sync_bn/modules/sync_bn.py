"""Import-path shim for `from libs.sync_bn.modules.sync_bn import BatchNorm2d, BatchNorm3d`
(models/GANet_deep.py:8).  The reference's thread/Queue SyncBN only makes sense under
nn.DataParallel and is out of scope; under one-process-per-GPU DDP use
torch.nn.SyncBatchNorm.convert_sync_batchnorm(model) -- RCCL all-gather of the statistics.

The classes below ARE torch's BatchNorm (same parameters, buffers and state_dict keys); the only
addition is a one-time warning when they are asked for training statistics in a setting where the
reference would have synchronised them across GPUs and these do not (nn.DataParallel replicas, or
torch.distributed with world size > 1): statistics are then per replica."""
import warnings

import torch
from torch.nn import SyncBatchNorm  # noqa: F401

__all__ = ["SyncBatchNorm", "BatchNorm1d", "BatchNorm2d", "BatchNorm3d"]

_warned = False


def _warn_if_unsynchronised(mod):
    global _warned
    if _warned or not mod.training:
        return
    replica = getattr(mod, "_is_replica", False)                       # set by nn.DataParallel's replicate()
    multi = torch.distributed.is_available() and torch.distributed.is_initialized() and \
        torch.distributed.get_world_size() > 1
    if replica or multi:
        _warned = True
        warnings.warn("libs.sync_bn shim: BatchNorm statistics are NOT synchronised across GPUs here (the reference's "
                      "SyncBN was); call torch.nn.SyncBatchNorm.convert_sync_batchnorm(model) before wrapping the model "
                      "in DistributedDataParallel to get cross-GPU statistics over RCCL.", RuntimeWarning, stacklevel=3)


class BatchNorm1d(torch.nn.BatchNorm1d):
    def forward(self, x):
        _warn_if_unsynchronised(self)
        return super().forward(x)


class BatchNorm2d(torch.nn.BatchNorm2d):
    def forward(self, x):
        _warn_if_unsynchronised(self)
        return super().forward(x)


class BatchNorm3d(torch.nn.BatchNorm3d):
    def forward(self, x):
        _warn_if_unsynchronised(self)
        return super().forward(x)
