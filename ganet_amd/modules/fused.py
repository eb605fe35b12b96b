"""Fused forms of the op chains around SGA / LGA in models/GANet_deep.py.  Opt-in: the
reference-compatible modules in ganet_amd.modules.GANet are unchanged."""
import torch
from torch.nn.modules.module import Module

from ..functions.fused import (LgaRegressFunction, NormDisparityRegressionFunction, SoftminDisparityRegressionFunction,
                               SoftminFunction, TrilinearUpsampleFunction, normalize_filters, normalize_guidance,
                               sga_forward_infer)
from ..functions.GANet import Lga2Function, LgaFunction, SgaFunction

__all__ = ["GuidedSGA", "GuidedSGABnRelu", "NormalizedLGA2", "NormDisparityRegression", "SoftminDisparityRegression",
           "DispAggTail", "TrilinearUpsample"]


class GuidedSGA(Module):
    """SGA on the RAW guidance: forward(x [N,C,D,H,W], g [N,20C,H,W]) == SGABlock.forward lines
    models/GANet_deep.py:263-269 (split, view, 4 x F.normalize(p=1, dim=2), SGA)."""

    def forward(self, x, g):
        k1, k2, k3, k4 = normalize_guidance(g, x.shape[1])
        return SgaFunction.apply(x, k1, k2, k3, k4)


class GuidedSGABnRelu(Module):
    """SGABlock.forward lines models/GANet_deep.py:263-271 for refine=True blocks: normalise the guidance, SGA, then
    `bn_relu` = BatchNorm3d + ReLU.  In eval mode without autograd the BatchNorm affine and the ReLU are applied inside
    the direction-merge kernel (no mask / arg-max / saved volumes); otherwise the ops run one after the other."""

    def __init__(self, bn):
        super().__init__()
        self.bn = bn                      # the block's torch.nn.BatchNorm3d (shared, not copied)

    def forward(self, x, g):
        ks = normalize_guidance(g, x.shape[1])
        if self.training or torch.is_grad_enabled() or not self.bn.track_running_stats:
            return torch.relu(self.bn(SgaFunction.apply(x, *ks)))
        bn = self.bn
        scale = torch.rsqrt(bn.running_var + bn.eps)
        if bn.affine:
            scale = scale * bn.weight
            shift = bn.bias - bn.running_mean * scale
        else:
            shift = -bn.running_mean * scale
        return sga_forward_infer(x, *ks, scale.float().contiguous(), shift.float().contiguous())


class NormalizedLGA2(Module):
    """DispAgg.lga (models/GANet_deep.py:234-237): LGA2(x, F.normalize(g, p=1, dim=1))."""

    def __init__(self, radius=2):
        super().__init__()
        self.radius = radius

    def forward(self, x, g):
        return Lga2Function.apply(x, normalize_filters(g), self.radius)


class NormDisparityRegression(Module):
    """F.normalize(x, p=1, dim=1) + DisparityRegression(maxdisp) in one pass (models/GANet_deep.py:246-247)."""

    def __init__(self, maxdisp):
        super().__init__()
        self.maxdisp = maxdisp + 1

    def forward(self, x):
        return NormDisparityRegressionFunction.apply(x.contiguous(), self.maxdisp)


class SoftminDisparityRegression(Module):
    """Disp.forward after the upsampling (models/GANet_deep.py:217-219): Softmin(dim=1) + DisparityRegression(maxdisp)
    in one pass over the volume (used for the two auxiliary disparities in training)."""

    def __init__(self, maxdisp):
        super().__init__()
        self.maxdisp = maxdisp + 1

    def forward(self, x):
        return SoftminDisparityRegressionFunction.apply(x.contiguous(), self.maxdisp)


class DispAggTail(Module):
    """DispAgg.forward after the trilinear upsampling (models/GANet_deep.py:243-247):
    lga(x, lg1) -> Softmin(dim=1) -> lga(x, lg2) -> F.normalize(p=1, dim=1) -> DisparityRegression.
    The last LGA pass carries the normalise + regression reductions in its epilogue (LgaRegressFunction): three LGA passes,
    one Softmin, one fused pass and a per-pixel division -- under no_grad the final volume is never written."""

    def __init__(self, maxdisp=192, radius=2):
        super().__init__()
        self.radius = radius
        self.ndisp = maxdisp + 1
        self.lga = NormalizedLGA2(radius)

    def forward(self, x, lg1, lg2):
        assert lg1.size() == lg2.size()
        x = self.lga(x, lg1)
        x = SoftminFunction.apply(x.contiguous())
        f2 = normalize_filters(lg2)
        x = LgaFunction.apply(x, f2, self.radius)                       # first pass of the second LGA2
        return LgaRegressFunction.apply(x, f2, self.radius, self.ndisp)  # second pass + normalise + regression


class TrilinearUpsample(Module):
    """F.interpolate(x, size, mode='trilinear', align_corners=False) for [N,C,D,H,W] volumes: the up-sampling in front of
    the Softmin / LGA tails (models/GANet_deep.py:212, 240), with a gather backward (see TrilinearUpsampleFunction)."""

    def forward(self, x, size):
        return TrilinearUpsampleFunction.apply(x.contiguous(), tuple(int(v) for v in size))
