"""TEST INFRASTRUCTURE -- the guided-aggregation operators as CPU torch.autograd.Functions on top of the C oracle
(oracle/ganet_oracle.c / oracle/_ref), so that a whole reference MODEL can be run and differentiated on the CPU as the
parity twin of the same model on the HIP ops.  Never imported by ganet_amd or harness/ (tests pass
`route_cpu_through_oracle` to harness.steps.build_model as a hook).

Buffer roles and pass chaining follow the reference's libs/GANet/functions/GANet.py (SgaFunction :8-48, Lga*Function
:51-263) through oracle.Oracle; GetCostVolume / DisparityRegression are the reference's own torch statements
(libs/GANet/modules/GANet.py:119-147) without the hard-coded .cuda()."""
import types

import numpy as np
import torch

from .fused_ref import disparity_regression


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)


def _t(a, like=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if like is None else t.to(like.device)        # device tensors make a host round trip (hybrid test arm)


class OracleSga(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ora, x, g0, g1, g2, g3):
        out, tmp, mask = ora.sga_forward(_np(x), _np(g0), _np(g1), _np(g2), _np(g3))
        ctx.ora, ctx.tmp, ctx.mask = ora, tmp, mask
        ctx.save_for_backward(x, g0, g1, g2, g3)
        return _t(out, x)

    @staticmethod
    def backward(ctx, go):
        x, g0, g1, g2, g3 = ctx.saved_tensors
        grads = ctx.ora.sga_backward(_np(x), _np(g0), _np(g1), _np(g2), _np(g3), ctx.tmp, ctx.mask, _np(go))
        return (None, *[_t(g, go) for g in grads])


class OracleLgaChain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ora, x, f, radius, passes):
        y, ins = ora.lga_chain_forward(_np(x), _np(f), radius, passes)
        ctx.ora, ctx.radius, ctx.ins = ora, radius, ins
        ctx.save_for_backward(f)
        return _t(y, x)

    @staticmethod
    def backward(ctx, gy):
        f, = ctx.saved_tensors
        gx, gf = ctx.ora.lga_chain_backward(ctx.ins, _np(f), _np(gy), ctx.radius)
        return None, _t(gx, gy), _t(gf, gy), None, None


def cost_volume(x, y, ndisp):
    """libs/GANet/modules/GANet.py:119-134 (ndisp = maxdisp + 1), on whatever device x lives."""
    N, C, H, W = x.shape
    cost = x.new_zeros(N, 2 * C, ndisp, H, W)
    for i in range(ndisp):
        if i > 0:
            cost[:, :C, i, :, i:] = x[:, :, :, i:]
            cost[:, C:, i, :, i:] = y[:, :, :, :-i]
        else:
            cost[:, :C, i] = x
            cost[:, C:, i] = y
    return cost.contiguous()


_LGA_PASSES = {"LGA": 1, "LGA2": 2, "LGA3": 3, "LGA3D": 1, "LGA3D2": 2, "LGA3D3": 3}


def route_cpu_through_oracle(model, ora):
    """Rebinds forward of every GA-op module instance in `model` (matched by the reference's class names, so it works
    for this repo's modules and for the reference's own) to the oracle-backed CPU forms above.  Returns the count.
    Works on a model that lives on the GPU too (each op then copies its tensors to the host and back): the hybrid arm of
    tests/test_gpu_model.py, which isolates the GA ops from PyTorch's own CPU-vs-GPU arithmetic differences."""
    n = 0
    for m in model.modules():
        kind = type(m).__name__
        if kind == "SGA":
            m.forward = types.MethodType(lambda self, x, g0, g1, g2, g3: OracleSga.apply(ora, x, g0, g1, g2, g3), m)
        elif kind in _LGA_PASSES:
            p = _LGA_PASSES[kind]
            m.forward = types.MethodType(lambda self, x, f, p=p: OracleLgaChain.apply(ora, x, f, self.radius, p), m)
        elif kind == "GetCostVolume":
            m.forward = types.MethodType(lambda self, x, y: cost_volume(x, y, self.maxdisp), m)
        elif kind == "DisparityRegression":
            m.forward = types.MethodType(lambda self, x: disparity_regression(x, self.maxdisp - 1), m)
        else:
            continue
        n += 1
    return n
