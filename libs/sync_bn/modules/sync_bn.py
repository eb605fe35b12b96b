"""Import-path shim for `from libs.sync_bn.modules.sync_bn import BatchNorm2d, BatchNorm3d`
(models/GANet_deep.py:8).  The reference's thread/Queue SyncBN only makes sense under
nn.DataParallel and is out of scope; under one-process-per-GPU DDP use
torch.nn.SyncBatchNorm.convert_sync_batchnorm(model) -- RCCL all-gather of the statistics."""
from torch.nn import BatchNorm1d, BatchNorm2d, BatchNorm3d, SyncBatchNorm  # noqa: F401

__all__ = ["SyncBatchNorm", "BatchNorm1d", "BatchNorm2d", "BatchNorm3d"]
