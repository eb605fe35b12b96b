"""Kernel logic of the fused caller-side normalisations on the CPU emulator build, through the
C ABI, against the reference's own torch statements executed on the CPU (oracle/fused_ref.py)."""
import numpy as np

import parity_cases as pc
import pytest
import torch

from oracle import fused_ref as fr

RTOL, ATOL = 1e-5, 1e-6      # fp32 elementwise ops: the only freedom is summation order / fma contraction


@pytest.fixture(scope="module")
def sim():
    from sim_util import sim_api
    return sim_api()


class _p:
    """An array argument of _call."""

    def __init__(self, a):
        assert a.flags["C_CONTIGUOUS"]
        self.a = a


_GUARD = "end"


@pytest.fixture(autouse=True, params=["end", "start"])
def guard_mode(request):
    """Every test of this file runs twice: buffers that END at an inaccessible page, buffers that BEGIN behind one."""
    global _GUARD
    _GUARD = request.param
    yield
    _GUARD = "end"


def _call(sim, name, *args):
    """sim.call with every array argument passed as a copy that ends at a guard page (parity_cases.guarded_empty): reads or
    writes past the end of a tensor crash here instead of depending on what the neighbouring memory is."""
    import parity_cases as pc
    bufs, conv = [], []
    for v in args:
        if isinstance(v, _p):
            g = pc.guarded_empty(v.a.shape, v.a.dtype, _GUARD)
            g[...] = v.a
            bufs.append((v.a, g))
            conv.append(g.ctypes.data)
        else:
            conv.append(v)
    try:
        return sim.call(name, *conv)
    finally:
        for a, g in bufs:
            a[...] = g


@pytest.mark.parametrize("N,C,H,W", [(1, 2, 3, 4), (2, 3, 5, 7), (1, 1, 1, 1)])
def test_guidance_normalize_forward_backward(sim, N, C, H, W):
    rng = np.random.default_rng(N * 100 + C)
    g = rng.standard_normal((N, 20 * C, H, W)).astype(np.float32)
    g[0, 0:5, 0, 0] = 0.0                                  # an all-zero tap group: the clamped-norm branch
    g[0, 7, 0, 0] = 0.0                                    # sgn(0) = 0
    gys = [rng.standard_normal((N, C, 5, H, W)).astype(np.float32) for _ in range(4)]
    ys = [np.full((N, C, 5, H, W), np.nan, np.float32) for _ in range(4)]
    _call(sim, "ganet_l1_normalize_forward", _p(g), *[_p(y) for y in ys], N, 4, C, 5, H, W, None)
    tg = torch.from_numpy(g).requires_grad_()
    want = fr.sgablock_guidance(tg, C)
    for y, w in zip(ys, want):
        np.testing.assert_allclose(y, w.detach().numpy(), rtol=RTOL, atol=ATOL)
    torch.autograd.backward(want, [torch.from_numpy(a) for a in gys])
    gx = np.full_like(g, np.nan)
    _call(sim, "ganet_l1_normalize_backward", _p(g), *[_p(a) for a in gys], _p(gx), N, 4, C, 5, H, W, None)
    wg = tg.grad.numpy()
    ok = np.abs(wg) < 1e6                                  # (clamped group: gradient = gy / 1e-12, compare relatively)
    np.testing.assert_allclose(gx[ok], wg[ok], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gx[~ok], wg[~ok], rtol=1e-4)


@pytest.mark.parametrize("K", [75, 27, 5, 9])
def test_filter_normalize_any_K(sim, K):
    rng = np.random.default_rng(K)
    N, H, W = 2, 4, 6
    g = rng.standard_normal((N, K, H, W)).astype(np.float32)
    gy = rng.standard_normal((N, K, H, W)).astype(np.float32)
    y = np.full_like(g, np.nan)
    _call(sim, "ganet_l1_normalize_forward", _p(g), _p(y), None, None, None, N, 1, 1, K, H, W, None)
    tg = torch.from_numpy(g).requires_grad_()
    want = fr.lga_filters(tg)
    np.testing.assert_allclose(y, want.detach().numpy(), rtol=RTOL, atol=ATOL)
    want.backward(torch.from_numpy(gy))
    gx = np.full_like(g, np.nan)
    _call(sim, "ganet_l1_normalize_backward", _p(g), _p(gy), None, None, None, _p(gx), N, 1, 1, K, H, W, None)
    np.testing.assert_allclose(gx, tg.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("N,maxdisp,H,W", [(1, 8, 3, 5), (2, 23, 4, 4), (1, 192, 2, 3)])
def test_norm_disparity_regression(sim, N, maxdisp, H, W):
    rng = np.random.default_rng(maxdisp)
    D = maxdisp + 1
    x = rng.random((N, D, H, W)).astype(np.float32)        # post-LGA probabilities are non-negative ...
    x[0, :, 0, 0] = rng.standard_normal(D)                 # ... but signs must work too
    x[0, 3, 0, 1] = 0.0
    go = rng.standard_normal((N, H, W)).astype(np.float32)
    out = np.full((N, H, W), np.nan, np.float32)
    sn = np.full((N, H, W), np.nan, np.float32)
    _call(sim, "ganet_norm_disparity_regression_forward", _p(x), _p(out), _p(sn), N, D, H, W, None)
    tx = torch.from_numpy(x).requires_grad_()
    want = fr.norm_regression(tx, maxdisp)
    np.testing.assert_allclose(out, want.detach().numpy(), rtol=1e-5, atol=1e-4)   # disparities up to 192: 1e-4 px
    np.testing.assert_allclose(sn, np.abs(x).sum(1), rtol=1e-5)
    want.backward(torch.from_numpy(go))
    gx = np.full_like(x, np.nan)
    _call(sim, "ganet_norm_disparity_regression_backward", _p(x), _p(out), _p(sn), _p(go), _p(gx), N, D, H, W, None)
    np.testing.assert_allclose(gx, tx.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_argument_errors(sim):
    from ganet_amd._native import GanetError
    a = np.zeros((1, 20, 2, 2), np.float32)
    with pytest.raises(GanetError, match="null"):
        _call(sim, "ganet_l1_normalize_forward", None, _p(a), None, None, None, 1, 1, 1, 5, 2, 2, None)
    with pytest.raises(GanetError, match="groups"):
        _call(sim, "ganet_l1_normalize_forward", _p(a), _p(a), _p(a), _p(a), _p(a), 1, 5, 1, 5, 2, 2, None)
    with pytest.raises(GanetError, match="null output 1"):
        _call(sim, "ganet_l1_normalize_forward", _p(a), _p(a), None, None, None, 1, 2, 1, 5, 2, 2, None)


@pytest.mark.parametrize("shape", [(1, 3, 9, 4, 8), (2, 2, 5, 3, 5)])     # slice % 4 == 0 (16-byte path) and not
@pytest.mark.parametrize("with_bn", [True, False])
def test_sga_forward_infer_bn_relu_epilogue(sim, port_oracle, shape, with_bn):
    """out = relu(scale[c] * SGA(x, g) + shift[c]) (models/GANet_deep.py:269-271 in eval mode) vs the C oracle + torch."""
    import parity_cases as pc
    x, gs, _ = pc.sga_inputs(shape, seed=sum(shape))
    N, C, D, H, W = shape
    rng = np.random.default_rng(3)
    scale = rng.uniform(0.5, 2.0, C).astype(np.float32)
    shift = rng.standard_normal(C).astype(np.float32)
    A = np.empty((4,) + shape, np.float32)
    out = np.full(shape, np.nan, np.float32)
    _call(sim, "ganet_sga_forward_infer", _p(x), *[_p(g) for g in gs], _p(A), _p(out),
             _p(scale) if with_bn else None, _p(shift) if with_bn else None, N, C, D, H, W, None)
    want, _, _ = port_oracle.sga_forward(x, *gs)
    if with_bn:
        want = np.maximum(want * scale.reshape(1, C, 1, 1, 1) + shift.reshape(1, C, 1, 1, 1), 0).astype(np.float32)
        np.testing.assert_allclose(out, want, rtol=1e-6, atol=1e-6)     # fma vs mul+add
    else:
        assert np.array_equal(out, want)


@pytest.mark.parametrize("N,D,H,W", [(1, 7, 3, 5), (2, 193, 2, 3), (1, 16, 4, 4), (2, 193, 2, 4), (1, 5, 3, 8)])
def test_softmin_forward_backward(sim, N, D, H, W):
    """nn.Softmin(dim=1) (models/GANet_deep.py:244) vs torch on the CPU, incl. large-magnitude inputs."""
    rng = np.random.default_rng(D)
    x = (rng.standard_normal((N, D, H, W)) * 5).astype(np.float32)
    x[0, :, 0, 0] *= 20.0                                   # spread of ~100: the running-max rescale matters
    gy = rng.standard_normal((N, D, H, W)).astype(np.float32)
    y = np.full_like(x, np.nan)
    _call(sim, "ganet_softmin_forward", _p(x), _p(y), N, D, H, W, None)
    tx = torch.from_numpy(x).requires_grad_()
    want = torch.nn.functional.softmin(tx, dim=1)
    np.testing.assert_allclose(y, want.detach().numpy(), rtol=2e-6, atol=1e-7)
    want.backward(torch.from_numpy(gy))
    gx = np.full_like(x, np.nan)
    y_ref = np.ascontiguousarray(want.detach().numpy())          # (kept alive across the call)
    _call(sim, "ganet_softmin_backward", _p(y_ref), _p(gy), _p(gx), N, D, H, W, None)
    # (gy_d - sum gy*y cancels where y ~ 1: the order of the 193-term dot product shows up at the 1e-6 level)
    np.testing.assert_allclose(gx, tx.grad.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,maxdisp,H,W", [(1, 6, 3, 5), (2, 192, 2, 3)])
def test_softmin_disparity_regression(sim, N, maxdisp, H, W):
    """Disp.forward lines models/GANet_deep.py:217-219 (Softmin(dim=1) + DisparityRegression) in one pass, vs torch."""
    rng = np.random.default_rng(maxdisp + 1)
    D = maxdisp + 1
    x = (rng.standard_normal((N, D, H, W)) * 3).astype(np.float32)
    x[0, :, 0, 0] *= 15.0
    go = rng.standard_normal((N, H, W)).astype(np.float32)
    out, mx, ss = (np.full((N, H, W), np.nan, np.float32) for _ in range(3))
    _call(sim, "ganet_softmin_regression_forward", _p(x), _p(out), _p(mx), _p(ss), N, D, H, W, None)
    tx = torch.from_numpy(x).requires_grad_()
    want = fr.disparity_regression(torch.nn.functional.softmin(tx, dim=1), maxdisp)
    np.testing.assert_allclose(out, want.detach().numpy(), rtol=1e-5, atol=1e-4)
    want.backward(torch.from_numpy(go))
    gx = np.full_like(x, np.nan)
    _call(sim, "ganet_softmin_regression_backward", _p(x), _p(out), _p(mx), _p(ss), _p(go), _p(gx), N, D, H, W, None)
    np.testing.assert_allclose(gx, tx.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("isz,osz", [((3, 4, 5), (7, 12, 15)), ((5, 6, 7), (13, 18, 21)), ((4, 3, 6), (4, 3, 6)),
                                     ((2, 5, 4), (9, 7, 22)), ((6, 8, 9), (3, 5, 4)), ((1, 1, 1), (4, 3, 2))])
def test_trilinear_upsample_matches_torch(sim, isz, osz):
    """ganet_trilinear_upsample_forward / _backward against F.interpolate(mode='trilinear', align_corners=False) and its
    autograd adjoint on the CPU: the 3x zoom of the models (65 -> 193 is 3x - 2: a non-integer ratio on the depth axis),
    identity, a fractional zoom, down-sampling and a single voxel."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(sum(isz) + sum(osz))
    S = 3
    x = rng.standard_normal((1, S) + isz).astype(np.float32)
    gy = rng.standard_normal((1, S) + osz).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_()
    yt = F.interpolate(xt, size=list(osz), mode="trilinear", align_corners=False)
    yt.backward(torch.from_numpy(gy))
    y = np.full((1, S) + osz, np.nan, np.float32)
    gx = np.full_like(x, np.nan)
    _call(sim, "ganet_trilinear_upsample_forward", _p(x), _p(y), S, *isz, *osz, None)
    _call(sim, "ganet_trilinear_upsample_backward", _p(gy), _p(gx), S, *isz, *osz, None)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gx, xt.grad.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape,r", [((1, 12, 5, 34), 2), ((2, 7, 3, 36), 2), ((1, 31, 2, 8), 2), ((1, 6, 4, 40), 1)])
@pytest.mark.parametrize("with_y", [True, False])
def test_lga_pass_with_regression_epilogue(sim, port_oracle, shape, r, with_y):
    """ganet_lga_forward_regress: y (when asked for) equals the plain pass, snorm = sum_d |y| and sdy = sum_d d*y of the
    oracle's output; y = NULL must not be touched."""
    rng = np.random.default_rng(sum(shape) + r)
    B, D, H, W = shape
    x = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal((B, 3 * (2 * r + 1) ** 2, H, W)), 1)
    want = port_oracle.lga_forward(x, f, r)
    y = np.full(shape, np.nan, np.float32)
    snorm = np.full((B, H, W), np.nan, np.float32)
    sdy = np.full((B, H, W), np.nan, np.float32)
    _call(sim, "ganet_lga_forward_regress", _p(x), _p(f), _p(y) if with_y else None, _p(snorm),
             _p(sdy), B, D, H, W, r, None)
    if with_y:
        assert np.abs(y - want).max() < 2e-5
    d = np.arange(D, dtype=np.float64)[None, :, None, None]
    np.testing.assert_allclose(snorm, np.abs(want.astype(np.float64)).sum(1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(sdy, (want.astype(np.float64) * d).sum(1), rtol=1e-5, atol=2e-4)


def _bn3d(C, seed, affine=True):
    """a BatchNorm3d with non-trivial statistics and affine parameters"""
    bn = torch.nn.BatchNorm3d(C, affine=affine)
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(C, generator=gen))
        bn.running_var.copy_(torch.rand(C, generator=gen) + 0.25)
        if affine:
            bn.weight.copy_(torch.randn(C, generator=gen))
            bn.bias.copy_(torch.randn(C, generator=gen))
    return bn


def _folded(bn):
    scale = (bn.running_var + bn.eps).rsqrt() * (bn.weight if bn.affine else 1.0)
    shift = (bn.bias if bn.affine else 0.0) - bn.running_mean * scale
    return (np.ascontiguousarray(scale.detach().numpy().astype(np.float32)),
            np.ascontiguousarray(shift.detach().numpy().astype(np.float32)))


@pytest.mark.parametrize("shape", [(1, 2, 3, 4, 8), (2, 3, 5, 3, 7), (2, 1, 1, 1, 1), (1, 5, 2, 2, 6)])
@pytest.mark.parametrize("folded", [True, False])
def test_sgablock_residual_tail(sim, shape, folded):
    """ganet_residual_relu_forward / _backward == models/GANet_deep.py:270-277 behind the convolution (bn -> `x += rem` -> relu):
    eval mode with the BatchNorm folded into (scale, shift), and the training form relu(u + rem) on u = bn(t) from torch;
    16-byte and scalar kernels (slice sizes 96 / 105 / 1 / 24), out of place and in place."""
    N, C, D, H, W = shape
    rng = np.random.default_rng(sum(shape))
    t = rng.standard_normal(shape).astype(np.float32)
    rem = rng.standard_normal(shape).astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    bn = _bn3d(C, 11).eval() if folded else _bn3d(C, 11).train()
    tt, tr = torch.from_numpy(t.copy()).requires_grad_(), torch.from_numpy(rem.copy()).requires_grad_()
    want = fr.sgablock_tail(tt, tr, bn)
    want.backward(torch.from_numpy(gy))
    if folded:
        scale, shift = _folded(bn)
        src, ps, ph = t, _p(scale), _p(shift)
    else:
        with torch.no_grad():
            src = np.ascontiguousarray(_bn3d(C, 11).train()(torch.from_numpy(t)).numpy())      # u = bn(t): the framework's part
        ps = ph = None
    y = np.full(shape, np.nan, np.float32)
    _call(sim, "ganet_residual_relu_forward", _p(src), _p(rem), ps, ph, _p(y), N, C, D, H, W, None)
    np.testing.assert_allclose(y, want.detach().numpy(), rtol=1e-5, atol=1e-6)
    # relu's zero set is exactly the reference's wherever the pre-activation is not within rounding of zero
    pre = (bn.eval() if folded else bn)(torch.from_numpy(t)).detach().numpy() + rem if folded else src + rem
    clear = np.abs(pre) > 1e-5
    assert np.array_equal((y > 0)[clear], (want.detach().numpy() > 0)[clear])
    # in place over t
    y2 = src.copy()
    api_y2 = pc.guarded_empty(shape, np.float32, _GUARD)
    api_y2[...] = y2
    sim.call("ganet_residual_relu_forward", api_y2.ctypes.data, np.ascontiguousarray(rem).ctypes.data,
             ps.a.ctypes.data if ps else None, ph.a.ctypes.data if ph else None, api_y2.ctypes.data, N, C, D, H, W, None)
    assert np.array_equal(api_y2, y)
    # backward
    g_rem = np.full(shape, np.nan, np.float32)
    if folded:
        g_t = np.full(shape, np.nan, np.float32)
        _call(sim, "ganet_residual_relu_backward", _p(y), _p(gy), ps, _p(g_t), _p(g_rem), N, C, D, H, W, None)
        np.testing.assert_allclose(g_t[clear], tt.grad.numpy()[clear], rtol=1e-5, atol=1e-6)
    else:
        _call(sim, "ganet_residual_relu_backward", _p(y), _p(gy), None, None, _p(g_rem), N, C, D, H, W, None)
    assert np.array_equal(g_rem[clear], tr.grad.numpy()[clear])          # a masked copy: exact


def test_sgablock_residual_tail_argument_checks(sim):
    from ganet_amd import _native
    a = np.zeros((1, 2, 2, 2, 4), np.float32)
    sc = np.ones(2, np.float32)
    with pytest.raises(_native.GanetError):        # scale without shift
        sim.call("ganet_residual_relu_forward", a.ctypes.data, a.ctypes.data, sc.ctypes.data, None, a.ctypes.data, 1, 2, 2, 2, 4, None)
    with pytest.raises(_native.GanetError):        # a scaled gradient needs grad_t
        sim.call("ganet_residual_relu_backward", a.ctypes.data, a.ctypes.data, sc.ctypes.data, None, a.ctypes.data, 1, 2, 2, 2, 4, None)
    with pytest.raises(_native.GanetError):
        sim.call("ganet_residual_relu_forward", a.ctypes.data, a.ctypes.data, None, None, a.ctypes.data, 1, 0, 2, 2, 4, None)
    # NaN passes through relu as in ATen
    t = np.array([np.nan, -1.0, 2.0, 0.0], np.float32).reshape(1, 1, 1, 1, 4)
    y = np.empty_like(t)
    sim.call("ganet_residual_relu_forward", t.ctypes.data, np.zeros_like(t).ctypes.data, None, None, y.ctypes.data, 1, 1, 1, 1, 4, None)
    assert np.isnan(y.ravel()[0]) and list(y.ravel()[1:]) == [0.0, 2.0, 0.0]
