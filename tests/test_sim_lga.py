"""LGA / GetCostVolume / DisparityRegression kernel logic on the CPU emulator
(tests/hipsim), through the C ABI, against golden fixtures and the oracle."""
import numpy as np
import pytest

import parity_cases as pc
from golden_util import lga_case_names, load


@pytest.fixture(scope="module")
def sim():
    from sim_util import sim_api
    return sim_api()


DEV = pc.NumpyDev()


@pytest.mark.parametrize("name", lga_case_names())
def test_lga_chain_matches_golden(sim, name):
    z = load("lga_golden.npz")
    r, passes = (int(v) for v in z[f"{name}.meta"])
    want = {"y": z[f"{name}.y"], "gx": z[f"{name}.gx"], "gf": z[f"{name}.gf"]}
    err = pc.check_lga_chain(sim, DEV, z[f"{name}.x"], z[f"{name}.f"], z[f"{name}.gy"], r, passes, want)
    assert max(err.values()) < 2e-5, err


@pytest.mark.parametrize("shape,r", [((1, 9, 10, 34), 2), ((2, 5, 17, 33), 2), ((1, 4, 9, 40), 1),
                                     ((1, 6, 3, 70), 3), ((1, 1, 8, 32), 2), ((1, 13, 16, 64), 2),
                                     ((1, 50, 5, 66), 2), ((1, 2, 2, 2), 2),
                                     # W % 4 == 0: the input is staged planar by 16-byte copies -- images narrower than a tile and than its staged row
                                     ((1, 3, 2, 4), 2), ((1, 5, 1, 8), 2), ((2, 7, 9, 12), 2), ((1, 1, 1, 4), 2), ((1, 2, 3, 16), 2), ((1, 9, 7, 36), 2)])
def test_lga_single_pass_vs_oracle(sim, port_oracle, shape, r):
    """Shapes that cross tile borders and LDS stage boundaries, odd and even widths (both window parities)."""
    rng = np.random.default_rng(sum(shape) + r)
    fs = list(shape)
    fs[1] = 3 * (2 * r + 1) ** 2
    x = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal(fs), 1)
    gy = rng.standard_normal(shape).astype(np.float32)
    y = port_oracle.lga_forward(x, f, r)
    gx, gf = port_oracle.lga_backward(x, f, gy, r)
    err = pc.check_lga_chain(sim, DEV, x, f, gy, r, 1, {"y": y, "gx": gx, "gf": gf})
    assert max(err.values()) < 2e-5, err


@pytest.mark.parametrize("wave,segs", [(0, 0), (1, 1), (1, 2), (1, 3), (1, 5), (1, 64)])
@pytest.mark.parametrize("shape,r", [((1, 11, 5, 34), 2), ((2, 6, 3, 68), 2), ((1, 7, 4, 40), 1), ((1, 5, 2, 36), 3),
                                     ((1, 1, 3, 32), 2), ((1, 9, 1, 2), 2), ((1, 31, 3, 8), 2), ((1, 14, 2, 4), 1),
                                     ((1, 12, 3, 33), 2), ((1, 2, 5, 7), 2), ((1, 26, 2, 35), 2)])
def test_lga_kernel_families_and_depth_segments(sim, port_oracle, shape, r, wave, segs):
    """256-thread tile kernels (GANET_LGA_WAVE=0, the fallback) vs the wave-autonomous plane-pair kernels (1, the default:
    forward and data-backward; any width; segments start on even planes; radius 3 falls back to the tile kernels), the latter
    with the disparity range cut into 1..D segments (every seam, single-pair segments, more segments than planes) and more
    planes than ring slots."""
    rng = np.random.default_rng(sum(shape) + r)
    fs = list(shape)
    fs[1] = 3 * (2 * r + 1) ** 2
    x = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal(fs), 1)
    gy = rng.standard_normal(shape).astype(np.float32)
    y = port_oracle.lga_forward(x, f, r)
    gx, gf = port_oracle.lga_backward(x, f, gy, r)
    sim.set_option("GANET_LGA_WAVE", wave)
    sim.set_option("GANET_LGA_SEGS", segs)
    try:
        err = pc.check_lga_chain(sim, DEV, x, f, gy, r, 1, {"y": y, "gx": gx, "gf": gf})
        assert max(err.values()) < 2e-5, err
    finally:
        sim.set_option("GANET_LGA_WAVE", 2)
        sim.set_option("GANET_LGA_SEGS", 0)


@pytest.mark.parametrize("guard", ["end", "start"])
@pytest.mark.parametrize("W", [11, 12, 8])      # scalar kernels (W % 4 != 0) and the four-columns-per-lane ones
def test_cost_volume_and_regression(sim, port_oracle, W, guard):
    DEV = pc.NumpyDev(guard)
    rng = np.random.default_rng(5)
    N, C, H, maxdisp = 2, 3, 4, 6
    Dn = maxdisp + 1
    x = DEV.to(rng.standard_normal((N, C, H, W)).astype(np.float32))
    y = DEV.to(rng.standard_normal((N, C, H, W)).astype(np.float32))
    cost = DEV.empty((N, 2 * C, Dn, H, W))
    sim.call("ganet_cost_volume_forward", x.ctypes.data, y.ctypes.data, cost.ctypes.data, N, C, Dn, H, W, None)
    assert np.array_equal(cost, port_oracle.cost_volume(x, y, maxdisp))
    # adjoint identity <cost(x,y), g> == <x, gx> + <y, gy>
    g = DEV.to(rng.standard_normal(cost.shape).astype(np.float32))
    gx, gy = DEV.empty(x.shape), DEV.empty(y.shape)
    sim.call("ganet_cost_volume_backward", g.ctypes.data, gx.ctypes.data, gy.ctypes.data, N, C, Dn, H, W, None)
    lhs = float((cost.astype(np.float64) * g).sum())
    rhs = float((x.astype(np.float64) * gx).sum() + (y.astype(np.float64) * gy).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
    p = DEV.to(rng.random((N, Dn, H, W)).astype(np.float32))
    out = DEV.empty((N, H, W))
    sim.call("ganet_disparity_regression_forward", p.ctypes.data, out.ctypes.data, N, Dn, H, W, None)
    np.testing.assert_allclose(out, port_oracle.disparity_regression(p, maxdisp), atol=1e-5)
    go = DEV.to(rng.standard_normal((N, H, W)).astype(np.float32))
    gp = DEV.empty(p.shape)
    sim.call("ganet_disparity_regression_backward", go.ctypes.data, gp.ctypes.data, N, Dn, H, W, None)
    want = go[:, None] * np.arange(Dn, dtype=np.float32)[None, :, None, None]
    assert np.array_equal(gp, want)


def test_lga_unsupported_radius(sim):
    from ganet_amd._native import GanetError
    x = np.zeros((1, 2, 2, 2), np.float32)
    with pytest.raises(GanetError, match="radius"):
        sim.call("ganet_lga_forward", x.ctypes.data, x.ctypes.data, x.ctypes.data + 4, 1, 2, 2, 2, 4, None)


@pytest.mark.parametrize("shape,r", [((1, 11, 5, 34), 2), ((2, 6, 3, 68), 2), ((1, 7, 4, 40), 1), ((1, 1, 3, 32), 2),
                                     ((1, 2, 5, 7), 2), ((1, 26, 2, 35), 2),
                                     ((1, 3, 2, 4), 2), ((1, 5, 1, 8), 2), ((2, 7, 9, 12), 2), ((1, 1, 1, 4), 2), ((1, 2, 3, 16), 2)])    # (planar staging, narrow images)
def test_plane_pair_filter_gradient(sim, port_oracle, shape, r):
    """lga_filter_grad_pp: odd and even D (a last pair with one real plane), more pairs than ring slots, accumulate mode
    through a two-pass chain."""
    rng = np.random.default_rng(sum(shape) + r)
    fs = list(shape)
    fs[1] = 3 * (2 * r + 1) ** 2
    x = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal(fs), 1)
    gy = rng.standard_normal(shape).astype(np.float32)
    y, ins = port_oracle.lga_chain_forward(x, f, r, 2)
    gx, gf = port_oracle.lga_chain_backward(ins, f, gy, r)
    err = pc.check_lga_chain(sim, DEV, x, f, gy, r, 2, {"y": y, "gx": gx, "gf": gf})
    assert max(err.values()) < 2e-5, err


@pytest.mark.parametrize("segs", [1, 2, 3])
@pytest.mark.parametrize("shape", [(1, 65, 3, 34), (1, 64, 2, 36), (2, 47, 2, 40)])
def test_plane_pair_apply_long_marches(sim, port_oracle, shape, segs):
    """lga_apply_pp on depth ranges long enough to reach its predicate-free steady body (several groups of pair steps), with
    the body changes at different phases: one, two and three segments; odd and even D."""
    rng = np.random.default_rng(sum(shape) + segs)
    x = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal((shape[0], 75) + shape[2:]), 1)
    gy = rng.standard_normal(shape).astype(np.float32)
    y = port_oracle.lga_forward(x, f, 2)
    gx, gf = port_oracle.lga_backward(x, f, gy, 2)
    sim.set_option("GANET_LGA_WAVE", 2)
    sim.set_option("GANET_LGA_SEGS", segs)
    try:
        err = pc.check_lga_chain(sim, DEV, x, f, gy, 2, 1, {"y": y, "gx": gx, "gf": gf})
        assert max(err.values()) < 2e-5, err
    finally:
        sim.set_option("GANET_LGA_WAVE", 2)
        sim.set_option("GANET_LGA_SEGS", 0)


@pytest.mark.parametrize("mode", ["guard_end", "guard_start", "late_reversed"])
@pytest.mark.parametrize("shape", [(1, 9, 3, 36), (1, 12, 4, 40), (2, 21, 5, 68), (1, 1, 3, 34), (1, 2, 2, 2), (1, 26, 2, 6),
                                   (1, 40, 3, 32), (1, 41, 7, 64), (1, 14, 9, 34), (3, 12, 4, 40), (4, 2, 2, 2), (3, 13, 4, 40)])
def test_apply_on_pair_interleaved_volumes(sim, port_oracle, shape, mode):
    """ganet_lga_apply_paired (ABI v7): one pass and its data-backward with the input OR the output volume in the
    pair-interleaved layout (the private intermediate of an LGA2): equal to the oracle on the API layout, zero padding of the
    last pair for odd D, depths on both sides of the steady body, border tiles, widths that are not a multiple of the tile;
    batches of three and four with an even D (the batch stride of an interleaved volume is ceil(D/2) pairs: ADVICE r2)."""
    B, D, H, W = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
    dev = pc.NumpyDev("start" if mode == "guard_start" else "end")
    if mode == "late_reversed":
        sim.set_option("HIPSIM_LATE_DMA", 1)
        sim.set_option("HIPSIM_LANE_ORDER", 1)
    try:
        df = dev.to(f)
        for tr in (0, 1):
            want = port_oracle.lga_forward(x, f, 2) if tr == 0 else port_oracle.lga_backward(np.zeros_like(x), f, x, 2)[0]
            xp, y = dev.to(pc.to_paired(x)), dev.empty(shape)
            sim.call("ganet_lga_apply_paired", dev.ptr(xp), dev.ptr(df), dev.ptr(y), B, D, H, W, 2, tr, 1, 0, None)
            assert np.abs(y - want).max() < 2e-5, ("paired in", tr)
            dx, yp = dev.to(x), dev.empty((B, (D + 1) // 2, H, W, 2))
            sim.call("ganet_lga_apply_paired", dev.ptr(dx), dev.ptr(df), dev.ptr(yp), B, D, H, W, 2, tr, 0, 1, None)
            assert np.abs(pc.from_paired(yp, D) - want).max() < 2e-5, ("paired out", tr)
            assert D % 2 == 0 or (yp[:, -1, :, :, 1] == 0).all(), "odd half of the last pair must be zero"
    finally:
        sim.set_option("HIPSIM_LATE_DMA", 0)
        sim.set_option("HIPSIM_LANE_ORDER", 0)


def test_apply_paired_argument_errors(sim):
    from ganet_amd._native import GanetError
    x = DEV.zeros((1, 4, 2, 4))
    with pytest.raises(GanetError, match="at most one"):
        sim.call("ganet_lga_apply_paired", DEV.ptr(x), DEV.ptr(x), DEV.ptr(DEV.zeros((1, 4, 2, 4))), 1, 4, 2, 4, 2, 0, 1, 1, None)
    with pytest.raises(GanetError, match="radius 2"):
        sim.call("ganet_lga_apply_paired", DEV.ptr(x), DEV.ptr(x), DEV.ptr(DEV.zeros((1, 4, 2, 4))), 1, 4, 2, 4, 1, 0, 1, 0, None)
    with pytest.raises(GanetError, match="W even"):
        sim.call("ganet_lga_apply_paired", DEV.ptr(x), DEV.ptr(x), DEV.ptr(DEV.zeros((1, 4, 2, 3))), 1, 4, 2, 3, 2, 0, 1, 0, None)



@pytest.mark.parametrize("mode", ["guard_end", "guard_start", "late_reversed"])
@pytest.mark.parametrize("shape", [(1, 9, 3, 36), (1, 12, 4, 40), (2, 21, 5, 68), (1, 1, 3, 34), (1, 2, 2, 2), (1, 26, 2, 6),
                                   (1, 40, 3, 32), (1, 41, 7, 64), (3, 12, 4, 40), (4, 2, 2, 2),
                                   (1, 9, 7, 100), (2, 6, 8, 70)])      # (tiles whose every tap is inside the image: seeded accumulators)
def test_filter_gradient_on_pair_interleaved_x(sim, port_oracle, shape, mode):
    """ganet_lga_filter_grad_paired (ABI v7): gf of one pass with x (or gy) in the pair-interleaved layout, write and
    accumulate mode; batches of three and four with an even D."""
    B, D, H, W = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
    _, want = port_oracle.lga_backward(x, f, gy, 2)
    dev = pc.NumpyDev("start" if mode == "guard_start" else "end")
    if mode == "late_reversed":
        sim.set_option("HIPSIM_LATE_DMA", 1)
        sim.set_option("HIPSIM_LANE_ORDER", 1)
    try:
        xp, dgy, gf = dev.to(pc.to_paired(x)), dev.to(gy), dev.empty(f.shape)
        sim.call("ganet_lga_filter_grad_paired", dev.ptr(xp), dev.ptr(dgy), dev.ptr(gf), B, D, H, W, 2, 0, 1, 0, None)
        assert np.abs(gf - want).max() < 3e-5
        sim.call("ganet_lga_filter_grad_paired", dev.ptr(xp), dev.ptr(dgy), dev.ptr(gf), B, D, H, W, 2, 1, 1, 0, None)
        assert np.abs(gf - 2 * want).max() < 6e-5
        # gy interleaved instead (x in the API layout), and neither
        dx, gyp = dev.to(x), dev.to(pc.to_paired(gy))
        sim.call("ganet_lga_filter_grad_paired", dev.ptr(dx), dev.ptr(gyp), dev.ptr(gf), B, D, H, W, 2, 0, 0, 1, None)
        assert np.abs(gf - want).max() < 3e-5
        sim.call("ganet_lga_filter_grad_paired", dev.ptr(dx), dev.ptr(dgy), dev.ptr(gf), B, D, H, W, 2, 1, 0, 0, None)
        assert np.abs(gf - 2 * want).max() < 6e-5
    finally:
        sim.set_option("HIPSIM_LATE_DMA", 0)
        sim.set_option("HIPSIM_LANE_ORDER", 0)


@pytest.mark.parametrize("mix", [1, 3, 0])
@pytest.mark.parametrize("shape", [(1, 9, 3, 36), (2, 21, 5, 68), (1, 40, 3, 32), (1, 41, 7, 64), (1, 2, 2, 2), (3, 12, 4, 40), (4, 2, 2, 2), (1, 47, 2, 100),
                                   (1, 11, 7, 100), (1, 3, 2, 4), (1, 5, 1, 8), (2, 7, 9, 12), (1, 1, 1, 4), (1, 2, 3, 16)])      # (planar staging of the API-layout operands, narrow images)
def test_two_pass_chain_with_pair_interleaved_intermediate(sim, port_oracle, shape, mix):
    """The call sequence of Lga2Function (default; GANET_LGA_PAIRED=0 switches it off) (ganet_amd/functions/GANet.py: _LgaChain): forward
    x -> t1 (interleaved) -> y; backward gf = gF(t1 interleaved, gy); g_t1 (interleaved) = gX(gy); gf += gF(x, g_t1); gx = gX(g_t1)."""
    B, D, H, W = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
    y_want, ins = port_oracle.lga_chain_forward(x, f, 2, 2)
    gx_want, gf_want = port_oracle.lga_chain_backward(ins, f, gy, 2)
    dev = DEV
    sim.set_option("GANET_LGA_MIX", mix)          # item lists of the interleaved kernels: default / three SIMDs assumed / whole tiles
    sim.set_option("HIPSIM_LATE_DMA", 1)
    try:
        err = pc.check_lga2_paired(sim, pc.NumpyDev("start" if mix == 3 else "end"), x, f, gy, 2, 2, {"y": y_want, "gx": gx_want, "gf": gf_want})
        assert max(err.values()) < 5e-5, err
    finally:
        sim.set_option("GANET_LGA_MIX", 1)
        sim.set_option("HIPSIM_LATE_DMA", 0)
    dx, df, dgy = dev.to(x), dev.to(f), dev.to(gy)
    t1p, y = dev.empty((B, (D + 1) // 2, H, W, 2)), dev.empty(shape)
    sim.call("ganet_lga_apply_paired", dev.ptr(dx), dev.ptr(df), dev.ptr(t1p), B, D, H, W, 2, 0, 0, 1, None)
    sim.call("ganet_lga_apply_paired", dev.ptr(t1p), dev.ptr(df), dev.ptr(y), B, D, H, W, 2, 0, 1, 0, None)
    assert np.abs(y - y_want).max() < 2e-5
    gf, gt1p, gx = dev.empty(f.shape), dev.empty((B, (D + 1) // 2, H, W, 2)), dev.empty(shape)
    sim.call("ganet_lga_filter_grad_paired", dev.ptr(t1p), dev.ptr(dgy), dev.ptr(gf), B, D, H, W, 2, 0, 1, 0, None)
    sim.call("ganet_lga_apply_paired", dev.ptr(dgy), dev.ptr(df), dev.ptr(gt1p), B, D, H, W, 2, 1, 0, 1, None)
    sim.call("ganet_lga_filter_grad_paired", dev.ptr(dx), dev.ptr(gt1p), dev.ptr(gf), B, D, H, W, 2, 1, 0, 1, None)
    sim.call("ganet_lga_apply_paired", dev.ptr(gt1p), dev.ptr(df), dev.ptr(gx), B, D, H, W, 2, 1, 1, 0, None)
    assert np.abs(gx - gx_want).max() < 2e-5 and np.abs(gf - gf_want).max() < 5e-5


@pytest.mark.parametrize("simds", [2, 3, 5, 7])
@pytest.mark.parametrize("shape", [(1, 40, 6, 64), (2, 33, 5, 68), (1, 64, 9, 36), (1, 47, 2, 100)])
def test_mixed_item_list_of_the_plane_pair_apply(sim, port_oracle, shape, simds):
    """GANET_LGA_MIX (LgaSegMix): whole tiles first, the tiles beyond a whole number per SIMD cut into depth segments --
    forward and data-backward equal to the oracle whatever the SIMD count assumed (2..7 here; late-landing copies on)."""
    B, D, H, W = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
    y = port_oracle.lga_forward(x, f, 2)
    gx, gf = port_oracle.lga_backward(x, f, gy, 2)
    sim.set_option("GANET_LGA_MIX", simds)
    sim.set_option("HIPSIM_LATE_DMA", 1)
    try:
        err = pc.check_lga_chain(sim, pc.NumpyDev("start" if simds % 2 else "end"), x, f, gy, 2, 1, {"y": y, "gx": gx, "gf": gf})
        assert max(err.values()) < 2e-5, err
    finally:
        sim.set_option("GANET_LGA_MIX", 1)
        sim.set_option("HIPSIM_LATE_DMA", 0)
