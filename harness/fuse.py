"""Swaps the opt-in fused op chains (ganet_amd.modules.fused) into an instantiated reference
model, without touching its parameters or state_dict keys: the three call-site edits of INTEGRATION.md done by
rebinding `forward` on the SGABlock / DispAgg / Disp instances.

  SGABlock.forward  models/GANet_deep.py:262-277   split + view + 4x normalize + SGA (+ bn_relu) -> GuidedSGABnRelu;
                                                   conv_refine's BatchNorm3d + `x += rem` + relu -> ResidualBnRelu (the convolution
                                                   itself stays MIOpen's)
  DispAgg.forward   models/GANet_deep.py:239-247   after the upsampling: lga, Softmin, lga, normalize, regression -> DispAggTail
  Disp.forward      models/GANet_deep.py:213-219   after the upsampling: Softmin + regression -> SoftminDisparityRegression
  both tails        models/GANet_deep.py:212, 240  F.interpolate(trilinear) -> TrilinearUpsample (gather backward instead of
                                                   ATen's atomics: 22.5 ms of a 114 ms training step, profiles/r2_model_train_stock.json)
"""
import types

import torch

from ganet_amd.modules.fused import (DispAggTail, GuidedSGA, GuidedSGABnRelu, ResidualBnRelu, SoftminDisparityRegression,
                                     TrilinearUpsample)

_UP = TrilinearUpsample()


def _upsampled(self, x):
    # (the reference: F.interpolate(..., mode='trilinear', align_corners=False), models/GANet_deep.py:212, 240)
    x = _UP(self.conv32x1(x), [self.maxdisp + 1, x.size()[3] * 3, x.size()[4] * 3])
    return torch.squeeze(x, 1)


def _sgablock_forward(self, x, g):
    rem = x
    self._fused_sga.train(self.training)      # the helpers are not submodules: they follow the block's mode by hand
    x = self._fused_sga(x, g)                 # normalise guidance + SGA (+ bn_relu)
    if self.refine:
        x = self.conv_refine.conv(x)          # BasicConv(relu=False) = Conv3d + BatchNorm3d (models/GANet_deep.py:238): the
    return self._fused_tail(x, rem)           # BatchNorm, `x += rem` and the ReLU (:270-277) are ResidualBnRelu's one pass


def _dispagg_forward(self, x, lg1, lg2):
    return self._fused_tail(_upsampled(self, x), lg1, lg2)


def _disp_forward(self, x):
    return self._fused_tail(_upsampled(self, x))


def use_fused_ops(model):
    """Returns the number of call sites rebound."""
    n = 0
    for m in model.modules():
        kind = type(m).__name__
        if kind == "SGABlock":
            # registered through object.__setattr__-free assignment would add a submodule (and state_dict keys):
            # keep the helper out of the module tree
            helper = GuidedSGABnRelu(m.bn_relu[0]) if m.refine else GuidedSGA()
            object.__setattr__(m, "_fused_sga", helper)
            object.__setattr__(m, "_fused_tail", ResidualBnRelu(m.conv_refine.bn if m.refine else m.bn))
            m.forward = types.MethodType(_sgablock_forward, m)
            n += 1
        elif kind == "DispAgg":
            object.__setattr__(m, "_fused_tail", DispAggTail(m.maxdisp))
            m.forward = types.MethodType(_dispagg_forward, m)
            n += 1
        elif kind == "Disp":
            object.__setattr__(m, "_fused_tail", SoftminDisparityRegression(m.maxdisp))
            m.forward = types.MethodType(_disp_forward, m)
            n += 1
    return n
