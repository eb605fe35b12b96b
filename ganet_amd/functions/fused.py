"""Callers' normalisations folded into single kernels.

In the reference these are chains of stock PyTorch ops AROUND the guided-aggregation operators:
  * SGABlock.forward   (models/GANet_deep.py:263-268): torch.split + view + 4 x F.normalize(p=1, dim=2)
  * DispAgg.lga        (models/GANet_deep.py:235):     F.normalize(g, p=1, dim=1)
  * DispAgg.forward    (models/GANet_deep.py:246-247): F.normalize(x, p=1, dim=1) + DisparityRegression
They sit BESIDE the drop-in API (libs/GANet/... keeps the reference's call forms unchanged): a model opts in by
calling GuidedSGA / NormalizedLGA2 / DispAggTail from ganet_amd.modules.fused instead of the op chains.
"""
import torch
from torch.autograd import Function

from .. import _native
from .GANet import _check, _p, _sga_infer, _stream

__all__ = ["L1NormalizeGroupsFunction", "NormDisparityRegressionFunction", "normalize_guidance", "normalize_filters",
           "sga_forward_infer", "SoftminFunction", "SoftminDisparityRegressionFunction", "TrilinearUpsampleFunction", "LgaRegressFunction",
           "ResidualReluFunction"]


def _lib():
    return _native.lib()


class L1NormalizeGroupsFunction(Function):
    """x [N, G*C*K, H, W] (= [N,G,C,K,H,W]) -> G tensors [N,C,K,H,W], each L1-normalised over K.

    G=4, K=5 is SGABlock's split/view/normalize of the guidance; G=C=1 is F.normalize(g, p=1, dim=1)."""

    @staticmethod
    def forward(ctx, x, G, C, K):
        _check(x)
        if x.dim() != 4 or x.shape[1] != G * C * K:
            raise ValueError(f"expected [N,{G * C * K},H,W], got {tuple(x.shape)}")
        if not 1 <= G <= 4:
            raise ValueError("1 <= G <= 4")
        N, _, H, W = x.shape
        ctx.dims = (N, G, C, K, H, W)
        with torch.cuda.device_of(x):
            y = torch.empty((G, N, C, K, H, W), dtype=x.dtype, device=x.device)
            ys = [y[g] for g in range(G)]
            ptrs = [_p(t) for t in ys] + [None] * (4 - G)
            _lib().call("ganet_l1_normalize_forward", _p(x), *ptrs, N, G, C, K, H, W, _stream())
        ctx.save_for_backward(x)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *grads):
        x, = ctx.saved_tensors
        N, G, C, K, H, W = ctx.dims
        gs = []
        for g in grads:
            g = torch.zeros((N, C, K, H, W), dtype=x.dtype, device=x.device) if g is None else g.contiguous()
            _check(g)
            gs.append(g)
        with torch.cuda.device_of(x):
            gx = torch.empty_like(x)
            ptrs = [_p(t) for t in gs] + [None] * (4 - G)
            _lib().call("ganet_l1_normalize_backward", _p(x), *ptrs, _p(gx), N, G, C, K, H, W, _stream())
        return gx, None, None, None


def normalize_guidance(g, channels):
    """[N, 20*channels, H, W] -> (k1, k2, k3, k4), each [N, channels, 5, H, W] and L1-normalised over dim 2
    (what SGABlock.forward computes with split + view + F.normalize, models/GANet_deep.py:263-268)."""
    return L1NormalizeGroupsFunction.apply(g.contiguous(), 4, channels, 5)


def normalize_filters(g):
    """F.normalize(g, p=1, dim=1) for LGA filters [N, K, H, W] (models/GANet_deep.py:235)."""
    N, K, H, W = g.shape
    (y,) = L1NormalizeGroupsFunction.apply(g.contiguous(), 1, 1, K)
    return y.view(N, K, H, W)


class NormDisparityRegressionFunction(Function):
    """out[N,H,W] = sum_d d * x[N,D,H,W] / max(sum_d |x|, 1e-12): F.normalize(x, p=1, dim=1) followed by
    DisparityRegression (models/GANet_deep.py:246-247, modules/GANet.py:136-148) in one pass."""

    @staticmethod
    def forward(ctx, x, ndisp):
        _check(x)
        if x.dim() != 4 or x.shape[1] != ndisp:
            raise ValueError(f"expected [N,{ndisp},H,W], got {tuple(x.shape)}")
        N, D, H, W = x.shape
        ctx.dims = (N, D, H, W)
        with torch.cuda.device_of(x):
            out = torch.empty((N, H, W), dtype=x.dtype, device=x.device)
            snorm = torch.empty_like(out)
            _lib().call("ganet_norm_disparity_regression_forward", _p(x), _p(out), _p(snorm), N, D, H, W, _stream())
        ctx.save_for_backward(x, out, snorm)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, out, snorm = ctx.saved_tensors
        g = grad_out.contiguous()
        _check(g)
        N, D, H, W = ctx.dims
        with torch.cuda.device_of(g):
            gx = torch.empty_like(x)
            _lib().call("ganet_norm_disparity_regression_backward", _p(x), _p(out), _p(snorm), _p(g), _p(gx),
                        N, D, H, W, _stream())
        return gx, None


def sga_forward_infer(x, g0, g1, g2, g3, bn_scale=None, bn_shift=None):
    """Inference-only SGA: relu(bn_scale[c] * SGA(x, g0..g3) + bn_shift[c]) in the merge kernel (plain SGA output when
    bn_scale is None).  Nothing is saved for a backward: call it under torch.no_grad()
    (SGABlock.forward, models/GANet_deep.py:269-271, in eval mode)."""
    ts = [x, g0, g1, g2, g3] + ([bn_scale, bn_shift] if bn_scale is not None else [])
    _check(*ts)
    if any(t.requires_grad for t in ts) and torch.is_grad_enabled():
        raise RuntimeError("sga_forward_infer keeps nothing for backward: wrap the call in torch.no_grad()")
    N, C, D, H, W = x.shape
    if bn_scale is not None and (bn_scale.numel() != C or bn_shift.numel() != C):
        raise ValueError("bn_scale / bn_shift must have one entry per channel")
    with torch.cuda.device_of(x):
        out = _sga_infer(x, g0, g1, g2, g3, torch.empty_like(x), bn_scale, bn_shift)
    return out


class SoftminFunction(Function):
    """nn.Softmin(dim=1) on [N,D,H,W] (models/GANet_deep.py:244) in one kernel each way."""

    @staticmethod
    def forward(ctx, x):
        _check(x)
        if x.dim() != 4:
            raise ValueError("expected [N,D,H,W]")
        N, D, H, W = x.shape
        with torch.cuda.device_of(x):
            y = torch.empty_like(x)
            _lib().call("ganet_softmin_forward", _p(x), _p(y), N, D, H, W, _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        y, = ctx.saved_tensors
        g = grad_y.contiguous()
        _check(g)
        N, D, H, W = y.shape
        with torch.cuda.device_of(g):
            gx = torch.empty_like(y)
            _lib().call("ganet_softmin_backward", _p(y), _p(g), _p(gx), N, D, H, W, _stream())
        return gx


class SoftminDisparityRegressionFunction(Function):
    """out[N,H,W] = sum_d d * softmin_d(x): Disp.forward's Softmin(dim=1) + DisparityRegression
    (models/GANet_deep.py:217-219) in one pass; the probabilities are never materialised."""

    @staticmethod
    def forward(ctx, x, ndisp):
        _check(x)
        if x.dim() != 4 or x.shape[1] != ndisp:
            raise ValueError(f"expected [N,{ndisp},H,W], got {tuple(x.shape)}")
        N, D, H, W = x.shape
        with torch.cuda.device_of(x):
            out = torch.empty((N, H, W), dtype=x.dtype, device=x.device)
            mx, ssum = torch.empty_like(out), torch.empty_like(out)
            _lib().call("ganet_softmin_regression_forward", _p(x), _p(out), _p(mx), _p(ssum), N, D, H, W, _stream())
        ctx.save_for_backward(x, out, mx, ssum)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, out, mx, ssum = ctx.saved_tensors
        g = grad_out.contiguous()
        _check(g)
        N, D, H, W = x.shape
        with torch.cuda.device_of(g):
            gx = torch.empty_like(x)
            _lib().call("ganet_softmin_regression_backward", _p(x), _p(out), _p(mx), _p(ssum), _p(g), _p(gx),
                        N, D, H, W, _stream())
        return gx, None


class TrilinearUpsampleFunction(Function):
    """F.interpolate(x, size, mode='trilinear', align_corners=False) on [N,C,D,H,W] (the cost-volume up-sampling of
    Disp.forward / DispAgg.forward, models/GANet_deep.py:212, 240) with a GATHER backward: one lane per input voxel sums the
    output gradients that read it, instead of ATen's eight atomicAdds per output element."""

    @staticmethod
    def forward(ctx, x, size):
        _check(x)
        if x.dim() != 5 or len(size) != 3:
            raise ValueError("TrilinearUpsample expects [N,C,D,H,W] and a 3-element output size")
        N, C, Di, Hi, Wi = x.shape
        Do, Ho, Wo = (int(v) for v in size)
        ctx.dims = (N, C, Di, Hi, Wi, Do, Ho, Wo)
        with torch.cuda.device_of(x):
            y = torch.empty((N, C, Do, Ho, Wo), dtype=x.dtype, device=x.device)
            _lib().call("ganet_trilinear_upsample_forward", _p(x), _p(y), N * C, Di, Hi, Wi, Do, Ho, Wo, _stream())
        return y

    @staticmethod
    def backward(ctx, gy):
        g = gy.contiguous()
        _check(g)
        N, C, Di, Hi, Wi, Do, Ho, Wo = ctx.dims
        with torch.cuda.device_of(g):
            gx = torch.empty((N, C, Di, Hi, Wi), dtype=g.dtype, device=g.device)
            _lib().call("ganet_trilinear_upsample_backward", _p(g), _p(gx), N * C, Di, Hi, Wi, Do, Ho, Wo, _stream())
        return gx, None


class LgaRegressFunction(Function):
    """One LGA pass followed by F.normalize(p=1, dim=1) + DisparityRegression -- the last three statements of
    DispAgg.forward (models/GANet_deep.py:245-247 after the first pass of the second LGA2) -- with the two reductions over
    the disparity axis done in the LGA kernel's epilogue (ganet_lga_forward_regress): out = sum_d d*y / max(sum_d |y|, 1e-12)
    is a per-pixel finish, and without autograd the volume y is never written."""

    @staticmethod
    def forward(ctx, x, filters, radius, ndisp):
        _check(x, filters)
        if x.dim() != 4 or x.shape[1] != ndisp:
            raise ValueError(f"expected [N,{ndisp},H,W], got {tuple(x.shape)}")
        N, D, H, W = x.shape
        if tuple(filters.shape) != (N, 3 * (2 * radius + 1) ** 2, H, W):
            raise ValueError("LGA filters must be [N, 3(2r+1)^2, H, W]")
        ctx.radius, ctx.dims = radius, (N, D, H, W)
        keep = any(ctx.needs_input_grad[:2])
        with torch.cuda.device_of(x):
            y = torch.empty_like(x) if keep else None
            snorm = torch.empty((N, H, W), dtype=x.dtype, device=x.device)
            sdy = torch.empty_like(snorm)
            try:
                _lib().call("ganet_lga_forward_regress", _p(x), _p(filters), _p(y) if keep else None, _p(snorm), _p(sdy),
                            N, D, H, W, radius, _stream())
                snorm.clamp_(min=1e-12)
                out = sdy / snorm
            except _native.GanetError as e:                # no fused kernel for this shape: the two separate entries
                if e.code != _native.E_UNSUPPORTED:
                    raise
                y = torch.empty_like(x)
                out = sdy
                _lib().call("ganet_lga_forward", _p(x), _p(filters), _p(y), N, D, H, W, radius, _stream())
                _lib().call("ganet_norm_disparity_regression_forward", _p(y), _p(out), _p(snorm), N, D, H, W, _stream())
        if keep:
            ctx.save_for_backward(x, filters, y, out, snorm)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, filters, y, out, snorm = ctx.saved_tensors
        g = grad_out.contiguous()
        _check(g)
        N, D, H, W = ctx.dims
        with torch.cuda.device_of(g):
            gy = torch.empty_like(y)
            _lib().call("ganet_norm_disparity_regression_backward", _p(y), _p(out), _p(snorm), _p(g), _p(gy),
                        N, D, H, W, _stream())
            gx = torch.empty_like(x)
            gf = torch.empty_like(filters)
            _lib().call("ganet_lga_backward", _p(x), _p(filters), _p(gy), _p(gx), _p(gf), N, D, H, W, ctx.radius, 0, _stream())
        return gx, gf, None, None


class ResidualReluFunction(Function):
    """The end of SGABlock.forward (models/GANet_deep.py:270-277) in one pass each way:
        y = relu(bn_scale[c] * t + bn_shift[c] + rem)      bn_scale / bn_shift: conv_refine's BatchNorm3d folded from its running
                                                           statistics (eval mode; constants: they get no gradient here)
        y = relu(t + rem)                                  bn_scale is None: t = bn(conv(..)) from the framework (training mode)
    for `x = conv_refine(x); x += rem; return relu(x)`.  `inplace` writes y over t as the reference's `x += rem` does (t must
    be a temporary nobody else needs: the convolution's / BatchNorm's output -- BatchNorm's backward reads its input, not t).
    Backward: g = grad_y where y > 0; both inputs take g (bn_scale given: t takes bn_scale[c] * g)."""

    @staticmethod
    def forward(ctx, t, rem, bn_scale=None, bn_shift=None, inplace=False):
        ts = [t, rem] + ([bn_scale, bn_shift] if bn_scale is not None else [])
        _check(*ts)
        if t.dim() != 5 or t.shape != rem.shape:
            raise ValueError(f"expected two [N,C,D,H,W] volumes of one shape, got {tuple(t.shape)} and {tuple(rem.shape)}")
        N, C, D, H, W = t.shape
        if (bn_scale is None) != (bn_shift is None):
            raise ValueError("bn_scale and bn_shift come together")
        if bn_scale is not None and (bn_scale.numel() != C or bn_shift.numel() != C):
            raise ValueError("bn_scale / bn_shift must have one entry per channel")
        ctx.dims = (N, C, D, H, W)
        with torch.cuda.device_of(t):
            y = t if inplace else torch.empty_like(t)
            _lib().call("ganet_residual_relu_forward", _p(t), _p(rem), _p(bn_scale) if bn_scale is not None else None,
                        _p(bn_shift) if bn_shift is not None else None, _p(y), N, C, D, H, W, _stream())
        if inplace:
            ctx.mark_dirty(t)
        ctx.scaled = bn_scale is not None
        if ctx.scaled:
            ctx.save_for_backward(y, bn_scale)
        else:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        y = ctx.saved_tensors[0]
        g = grad_y.contiguous()
        _check(g)
        N, C, D, H, W = ctx.dims
        with torch.cuda.device_of(g):
            g_rem = torch.empty_like(y)
            if ctx.scaled:
                g_t = torch.empty_like(y)
                _lib().call("ganet_residual_relu_backward", _p(y), _p(g), _p(ctx.saved_tensors[1]), _p(g_t), _p(g_rem),
                            N, C, D, H, W, _stream())
            else:
                g_t = g_rem
                _lib().call("ganet_residual_relu_backward", _p(y), _p(g), None, None, _p(g_rem), N, C, D, H, W, _stream())
        return g_t, g_rem, None, None, None
