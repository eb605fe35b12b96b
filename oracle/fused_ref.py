"""TEST INFRASTRUCTURE -- CPU oracle for the callers' op chains of SURVEY.md 8f (never imported by ganet_amd).

The reference implements these steps with stock PyTorch ops, so the oracle IS the reference's own statement
sequence executed on the CPU (pinned by construction; torch CPU ships in this image and on the GPU box):
  sgablock_guidance   models/GANet_deep.py:263-267   torch.split + view + F.normalize(p=1, dim=2)
  lga_filters         models/GANet_deep.py:235       F.normalize(g, p=1, dim=1)
  norm_regression     models/GANet_deep.py:246-247 + libs/GANet/modules/GANet.py:142-147
                      F.normalize(x, p=1, dim=1), then sum(x * arange(maxdisp+1), 1)
  sgablock_tail       models/GANet_deep.py:270-277   bn (of conv_refine) -> `x += rem` -> relu
  dispagg_tail        models/GANet_deep.py:243-247   lga -> Softmin(dim=1) -> lga -> normalize -> regression,
                      the two LGA2 calls through the C oracle (oracle/ganet_oracle.c)
Gradients come from torch.autograd on the same CPU graph.
"""
import numpy as np
import torch
import torch.nn.functional as F


def sgablock_guidance(g, channels):
    """g [N,20C,H,W] -> [k1..k4], each [N,C,5,H,W]   (models/GANet_deep.py:263-267)"""
    N, _, H, W = g.shape
    ks = torch.split(g, (channels * 5,) * 4, 1)
    return [F.normalize(k.reshape(N, channels, 5, H, W), p=1, dim=2) for k in ks]


def lga_filters(g):
    return F.normalize(g, p=1, dim=1)          # models/GANet_deep.py:235


def disparity_regression(x, maxdisp):
    """libs/GANet/modules/GANet.py:142-147 without the hard-coded .cuda()"""
    disp = torch.arange(0, maxdisp + 1, dtype=x.dtype, device=x.device).reshape(1, maxdisp + 1, 1, 1)
    disp = disp.repeat(x.size(0), 1, x.size(2), x.size(3))
    return torch.sum(x * disp, 1)


def norm_regression(x, maxdisp):
    return disparity_regression(F.normalize(x, p=1, dim=1), maxdisp)     # models/GANet_deep.py:246-247


class _OracleLga2(torch.autograd.Function):
    """Lga2Function (functions/GANet.py:174-203) on the C oracle, so the tail can be differentiated on the CPU."""

    @staticmethod
    def forward(ctx, x, f, ora, radius):
        y, ins = ora.lga_chain_forward(x.detach().numpy(), f.detach().numpy(), radius, 2)
        ctx.ora, ctx.radius, ctx.ins = ora, radius, ins
        ctx.save_for_backward(f)
        return torch.from_numpy(np.ascontiguousarray(y))

    @staticmethod
    def backward(ctx, gy):
        f, = ctx.saved_tensors
        gx, gf = ctx.ora.lga_chain_backward(ctx.ins, f.detach().numpy(), np.ascontiguousarray(gy.numpy()), ctx.radius)
        return torch.from_numpy(np.ascontiguousarray(gx)), torch.from_numpy(np.ascontiguousarray(gf)), None, None


def dispagg_tail(x, lg1, lg2, maxdisp, ora, radius=2, parts=None):
    """models/GANet_deep.py:243-247 (x already upsampled and squeezed to [N,maxdisp+1,H,W]).  `parts` (a dict) also receives
    the volume in front of the final normalise + regression (the second LGA2's output), for tests of the fused tail."""
    x = _OracleLga2.apply(x, lga_filters(lg1), ora, radius)
    x = F.softmin(x, dim=1)
    x = _OracleLga2.apply(x, lga_filters(lg2), ora, radius)
    if parts is not None:
        parts["y2"] = x.detach()
    return norm_regression(x, maxdisp)


def sgablock_tail(t, rem, bn):
    """models/GANet_deep.py:270-277 behind the convolution: `x = bn(t)` (conv_refine's BatchNorm3d, BasicConv.forward :36-38;
    refine=False blocks: the block's own bn, :273), `x += rem`, `relu(x)` -- the reference's statements on the CPU; `bn` is a
    torch.nn.BatchNorm3d in whichever mode the caller put it."""
    x = bn(t)
    x += rem
    return F.relu(x, inplace=True)
