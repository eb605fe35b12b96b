"""The training / inference step of the reference around its model, for one process per GPU.

Mirrors (reference file:line):
  loss mix            train.py:100-118   GANet_deep: 0.2*L1s(disp0) + 0.6*L1s(disp1) + L(disp2), L = MyLoss2(thresh=3, alpha=2)
                                         on KITTI else smooth-L1; GANet11: 0.4*L1s(disp1) + 1.2*L(disp2)
  valid-pixel mask    train.py:93-96     target < max_disp
  optimizer           train.py:74        Adam(lr=1e-3, betas=(0.9, 0.999))
  data parallelism    train.py:73        nn.DataParallel + thread SyncBN  ->  here DistributedDataParallel over RCCL
                                         ("nccl" backend on ROCm) + torch.nn.SyncBatchNorm, one process per GPU
  checkpoints         train.py:75-82, 193-197, predict.py:57-63   {'epoch','state_dict','optimizer'}, keys carry the
                                         `module.` prefix of the DataParallel wrapper; loaded with strict=False
  inference           predict.py:100-114 model.eval(); with torch.no_grad(): model(left, right)
"""
import os

import torch
import torch.nn.functional as F

from . import refmodel


def build_model(name="GANet_deep", max_disp=192, device="cuda", sync_bn=False, ddp=False, local_rank=0, hook=None):
    """GANet(max_disp) of the reference on `device`.  sync_bn: torch.nn.SyncBatchNorm (statistics all-gathered over
    RCCL); ddp: wrap in DistributedDataParallel (gradient all-reduce bucketed and overlapped with backward by torch).
    hook(model) runs before wrapping (tests use it to route CPU tensors through the CPU oracle)."""
    model = refmodel.model_class(name)(max_disp)
    if hook is not None:
        hook(model)
    model = model.to(device)
    if sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    if ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP
        # GANet_deep constructs a block it never calls (cost_agg.deconv0b, models/GANet_deep.py:306): its parameters get
        # no gradient, so the reducer must be told to look for unused parameters
        unused = any(k.startswith("cost_agg.deconv0b.") for k, _ in model.named_parameters())
        if torch.device(device).type == "cuda":
            model = DDP(model, device_ids=[local_rank], output_device=local_rank, bucket_cap_mb=32,
                        find_unused_parameters=unused)
        else:
            model = DDP(model, find_unused_parameters=unused)
    return model


def criterion(kitti=True):
    from libs.GANet.modules.GANet import MyLoss2
    return MyLoss2(thresh=3, alpha=2) if kitti else (lambda a, b: F.smooth_l1_loss(a, b, reduction="mean"))


def loss_mix(name, outputs, target, mask, crit):
    """train.py:100-118 for the two shipped models."""
    l1 = lambda d: F.smooth_l1_loss(d[mask], target[mask], reduction="mean")   # noqa: E731
    if name == "GANet11":
        disp1, disp2 = outputs
        return 0.4 * l1(disp1) + 1.2 * crit(disp2[mask], target[mask])
    disp0, disp1, disp2 = outputs
    return 0.2 * l1(disp0) + 0.6 * l1(disp1) + crit(disp2[mask], target[mask])


def train_step(model, optimizer, name, left, right, target, max_disp, crit):
    """One optimisation step (train.py:85-120).  Returns (loss, mean abs error of the last disparity)."""
    model.train()
    mask = (target < max_disp).detach()
    # train.py:97-99 skips a batch without valid pixels.  With one process per GPU the skip has to be COLLECTIVE: a rank that
    # returned early would leave the others waiting in the gradient all-reduce.  The step is skipped only if no
    # rank has a valid pixel; a rank whose own shard is empty runs forward + backward with a zero-weighted loss.
    nvalid = mask.sum()
    multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    stat = torch.stack([nvalid, (nvalid == 0).to(nvalid.dtype)])       # valid pixels of all ranks, ranks with an empty shard
    if multi:
        torch.distributed.all_reduce(stat)
    total, n_empty = int(stat[0]), int(stat[1])
    if total == 0:
        return None, None
    optimizer.zero_grad()
    outputs = model(left, right)
    if int(nvalid) == 0:
        # EVERY rank runs the same backward graph: DDP's reducer marks unused parameters ready on its own (a loss that
        # touched them directly made it mark them twice), and SyncBatchNorm's backward all-reduces on the default group -- a rank
        # that bypassed the data path would issue a different collective sequence than its peers.  So the zero-weighted loss
        # stays ON the forward graph; what that can do to the gradients (0 * inf = NaN) is dealt with after the backward.
        loss = sum(torch.nan_to_num(o).sum() for o in outputs) * 0.0
        err = loss.detach()
    else:
        loss = loss_mix(name, outputs, target, mask, crit)
        err = torch.mean(torch.abs(outputs[-1][mask] - target[mask])).detach()
    loss.backward()
    if multi and n_empty:
        # a non-finite activation on a zero-weighted rank has reached everybody's averaged gradients by now: the step is skipped
        # by a COLLECTIVE decision (the averaged gradients are the same on every rank; the flag is reduced all the same, so that
        # no rank can step alone)
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        bad = (~torch.isfinite(torch.stack(torch._foreach_norm(grads)))).any().to(torch.int32) if grads else torch.zeros((), dtype=torch.int32)
        bad = bad.to(nvalid.device)
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
        if int(bad):
            optimizer.zero_grad()
            return loss.detach(), err
    optimizer.step()
    return loss.detach(), err


@torch.no_grad()
def predict(model, left, right):
    """predict.py:107-114."""
    model.eval()
    return model(left, right)


def synthetic_batch(batch, height, width, max_disp, device, seed=123):
    """Standardised random images and a plausible disparity map (no dataset offline): same shapes and value ranges as
    dataloader/dataset.py produces (per-channel zero-mean unit-variance images, disparities in [0, max_disp))."""
    g = torch.Generator().manual_seed(seed)
    left = torch.randn(batch, 3, height, width, generator=g)
    right = torch.randn(batch, 3, height, width, generator=g)
    target = torch.rand(batch, height, width, generator=g) * (max_disp * 0.9)
    return left.to(device), right.to(device), target.to(device)


# ---- checkpoints -------------------------------------------------------------------------------------------------

def unwrap(model):
    return model.module if hasattr(model, "module") else model


def checkpoint_state(model, optimizer, epoch):
    """The dict train.py:193-197 saves.  Keys carry `module.` (the reference saves the DataParallel wrapper's state)."""
    sd = {"module." + k: v for k, v in unwrap(model).state_dict().items()}
    return {"epoch": epoch, "state_dict": sd, "optimizer": optimizer.state_dict()}


def load_state_dict_compat(model, state_dict, strict=False):
    """Loads a reference checkpoint's `state_dict` whatever wrapper it was saved from: strips or adds the `module.` prefix
    to match `model` (train.py:79 / predict.py:60 load into a DataParallel wrapper with strict=False)."""
    want_prefix = hasattr(model, "module")
    fixed = {}
    for k, v in state_dict.items():
        has = k.startswith("module.")
        if has and not want_prefix:
            k = k[len("module."):]
        elif not has and want_prefix:
            k = "module." + k
        fixed[k] = v
    return model.load_state_dict(fixed, strict=strict)


def save_checkpoint(path, model, optimizer, epoch):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(checkpoint_state(model, optimizer, epoch), path)


def load_checkpoint(path, model, optimizer=None, map_location="cpu"):
    ck = torch.load(path, map_location=map_location, weights_only=False)
    res = load_state_dict_compat(model, ck["state_dict"])
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    return ck.get("epoch", 0), res
