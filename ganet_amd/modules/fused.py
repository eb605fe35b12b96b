"""Fused forms of the op chains around SGA / LGA in models/GANet_deep.py.  Opt-in: the
reference-compatible modules in ganet_amd.modules.GANet are unchanged."""
import torch
from torch.nn.modules.module import Module

from ..functions.fused import (LgaRegressFunction, NormDisparityRegressionFunction, ResidualReluFunction,
                               SoftminDisparityRegressionFunction, SoftminFunction, TrilinearUpsampleFunction,
                               normalize_filters, normalize_guidance, sga_forward_infer)
from ..functions.GANet import Lga2Function, LgaFunction, SgaFunction

__all__ = ["GuidedSGA", "GuidedSGABnRelu", "NormalizedLGA2", "NormDisparityRegression", "SoftminDisparityRegression",
           "DispAggTail", "TrilinearUpsample", "ResidualBnRelu", "folded_bn"]


def folded_bn(bn):
    """(scale, shift) with bn(x) == scale[c] * x + shift[c] for a BatchNorm in eval mode (running statistics).  The pair is
    kept on the module and recomputed when any of its four tensors has been written since (their autograd version counters):
    six tiny launches per call otherwise, which is what an SGABlock tail on a 26 MB volume costs altogether."""
    src = (bn.running_mean, bn.running_var) + ((bn.weight, bn.bias) if bn.affine else ())
    key = tuple((t.data_ptr(), t._version) for t in src) + (bn.eps,)
    hit = bn.__dict__.get("_ganet_folded")
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        scale = torch.rsqrt(bn.running_var + bn.eps)
        if bn.affine:
            scale = scale * bn.weight
            shift = bn.bias - bn.running_mean * scale
        else:
            shift = -bn.running_mean * scale
        pair = (scale.float().contiguous(), shift.float().contiguous())
    bn.__dict__["_ganet_folded"] = (key, pair)       # not a buffer, not a parameter: state_dict keys stay the reference's
    return pair


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
        return sga_forward_infer(x, *ks, *folded_bn(self.bn))


class ResidualBnRelu(Module):
    """The end of SGABlock.forward (models/GANet_deep.py:270-277): forward(t, rem) == relu(bn(t) + rem) for
    `x = conv_refine(x); x += rem; return relu(x)`, where conv_refine = Conv3d + `bn` (BasicConv(relu=False), :238) and the
    caller passes t = conv_refine.conv(x).  (refine=False blocks, :273: t = the SGA output, bn = the block's own `bn`.)

    Eval mode with frozen statistics: the BatchNorm is folded into a per-channel affine and the whole tail is ONE pass over the
    volumes (stock PyTorch: batch_norm + add_ + relu_ = 7 volume passes, here 3); its backward hands g = grad * [y > 0] to
    `rem` and bn_scale[c] * g to t.  Training mode, or a BatchNorm whose parameters want gradients: `bn` runs in the framework
    (batch statistics, running-stat updates, SyncBatchNorm's collectives all stay what they were) and add + ReLU are fused."""

    def __init__(self, bn, inplace=True):
        super().__init__()
        self.bn = bn                      # the block's BatchNorm3d (shared, not copied)
        self.inplace = inplace            # y overwrites t, as the reference's `x += rem` does (t: the convolution's output)

    def forward(self, t, rem):
        bn = self.bn
        frozen = (not bn.training and bn.track_running_stats and
                  not (torch.is_grad_enabled() and any(p.requires_grad for p in bn.parameters())))
        t, rem = t.contiguous(), rem.contiguous()
        if frozen:
            # autograd refuses an in-place write to a leaf that wants a gradient; anything else (a convolution's output: its
            # backward does not read it) may be overwritten
            inplace = self.inplace and not (torch.is_grad_enabled() and t.requires_grad and t.is_leaf)
            return ResidualReluFunction.apply(t, rem, *folded_bn(bn), inplace)
        return ResidualReluFunction.apply(bn(t), rem, None, None, True)     # bn(t) is a temporary of this call: always in place


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
