"""nn.Module wrappers with the names and call forms of the reference's
libs/GANet/modules/GANet.py (SGA()(x,g0,g1,g2,g3), LGA2(r)(x,f), GetCostVolume(m)(x,y),
DisparityRegression(m)(x), ...), so models/GANet_deep.py and models/GANet11.py run unchanged."""
import torch
from torch.nn.modules.module import Module

from ..functions.GANet import (DisparityRegressionFunction, GetCostVolumeFunction, Lga2Function,
                               Lga3d2Function, Lga3d3Function, Lga3dFunction, Lga3Function,
                               LgaFunction, MyLoss2Function, MyLossFunction, SgaFunction)

__all__ = ["MyNormalize", "MyLoss2", "MyLoss", "SGA", "LGA3D3", "LGA3D2", "LGA3D", "LGA3", "LGA2", "LGA",
           "GetCostVolume", "DisparityRegression"]


class MyNormalize(Module):
    """x / (sum_dim |x| +- 1e-6)  (modules/GANet.py:18-33; unused by the shipped models)."""

    def __init__(self, dim):
        self.dim = dim
        super(MyNormalize, self).__init__()

    def forward(self, x):
        norm = torch.sum(torch.abs(x), self.dim, keepdim=True)
        # the reference first subtracts 1e-6 where norm <= 0, then adds 1e-6 where the result
        # is >= 0; with norm >= 0 this nets to: 0 -> -1e-6 (stays), > 0 -> +1e-6
        norm = torch.where(norm <= 0, norm - 1e-6, norm)
        norm = torch.where(norm >= 0, norm + 1e-6, norm)
        return torch.div(x, norm)


class MyLoss2(Module):
    def __init__(self, thresh=1, alpha=2):
        super(MyLoss2, self).__init__()
        self.thresh = thresh
        self.alpha = alpha

    def forward(self, input1, input2):
        return MyLoss2Function.apply(input1, input2, self.thresh, self.alpha)


class MyLoss(Module):
    def __init__(self, upper_thresh=5, lower_thresh=1):
        super(MyLoss, self).__init__()
        self.upper_thresh = 5      # the reference ignores its arguments (modules/GANet.py:45-46)
        self.lower_thresh = 1

    def forward(self, input1, input2):
        return MyLossFunction.apply(input1, input2, self.upper_thresh, self.lower_thresh)


class SGA(Module):
    def __init__(self):
        super(SGA, self).__init__()

    def forward(self, input, g0, g1, g2, g3):
        return SgaFunction.apply(input, g0, g1, g2, g3)


def _lga_module(name, fn):
    def __init__(self, radius=2):
        Module.__init__(self)
        self.radius = radius

    def forward(self, input1, input2):
        return fn.apply(input1, input2, self.radius)

    return type(name, (Module,), {"__init__": __init__, "forward": forward})


LGA3D3 = _lga_module("LGA3D3", Lga3d3Function)
LGA3D2 = _lga_module("LGA3D2", Lga3d2Function)
LGA3D = _lga_module("LGA3D", Lga3dFunction)
LGA3 = _lga_module("LGA3", Lga3Function)
LGA2 = _lga_module("LGA2", Lga2Function)
LGA = _lga_module("LGA", LgaFunction)


class GetCostVolume(Module):
    def __init__(self, maxdisp):
        super(GetCostVolume, self).__init__()
        self.maxdisp = maxdisp + 1

    def forward(self, x, y):
        assert x.is_contiguous()
        return GetCostVolumeFunction.apply(x, y.contiguous(), self.maxdisp)


class DisparityRegression(Module):
    def __init__(self, maxdisp):
        super(DisparityRegression, self).__init__()
        self.maxdisp = maxdisp + 1

    def forward(self, x):
        assert x.is_contiguous()
        return DisparityRegressionFunction.apply(x, self.maxdisp)
