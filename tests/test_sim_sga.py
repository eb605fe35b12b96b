"""SGA kernel logic on the CPU: the product's HIP kernel source compiled against the
lockstep wave64 emulator (tests/hipsim) and driven through the C ABI, compared with the
oracle and the reference-generated golden fixtures.  (The same checks run on the real
gfx950 build in tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

import parity_cases as pc
from golden_util import load, sga_case_names


@pytest.fixture(scope="module")
def sim():
    from sim_util import sim_api
    return sim_api()


DEV = pc.NumpyDev()


def _oracle_want(oracle, x, gs, go):
    out, tmp, mask = oracle.sga_forward(x, *gs)
    grads = oracle.sga_backward(x, *gs, tmp, mask, go)
    want = {"out": out, "mask": mask.astype(np.uint8), "tmp": tmp, "gx": grads[0]}
    for d in range(4):
        want[f"gw{d}"] = grads[1 + d]
        want[f"A{d}"] = oracle.sga_scan(x, gs[d], d)
    return want


@pytest.mark.parametrize("direction", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(1, 2, 5, 4, 8), (2, 1, 17, 3, 6), (1, 3, 1, 2, 4), (1, 1, 35, 5, 4)])
def test_scan_matches_oracle_bit_exact(sim, port_oracle, shape, direction):
    x, gs, _ = pc.sga_inputs(shape, seed=7 + direction)
    pc.check_sga_scan(sim, DEV, port_oracle, x, gs[direction], direction)


@pytest.mark.parametrize("name", sga_case_names())
def test_forward_backward_match_golden(sim, name):
    z = load("sga_golden.npz")
    x, go = z[f"{name}.x"], z[f"{name}.go"]
    gs = [z[f"{name}.g{d}"] for d in range(4)]
    want = {"out": z[f"{name}.out"], "mask": z[f"{name}.mask"], "gx": z[f"{name}.gx"]}
    for d in range(4):
        want[f"A{d}"] = z[f"{name}.A{d}"]
        want[f"gw{d}"] = z[f"{name}.gw{d}"]
    pc.check_sga_forward_backward(sim, DEV, x, gs, go, want)


@pytest.mark.parametrize("name", ["tiny", "ties", "w1", "d33"])
def test_reference_buffer_contract(sim, name):
    z = load("sga_golden.npz")
    x, go = z[f"{name}.x"], z[f"{name}.go"]
    gs = [z[f"{name}.g{d}"] for d in range(4)]
    want = {"out": z[f"{name}.out"], "mask": z[f"{name}.mask"], "tmp": z[f"{name}.tmp"], "gx": z[f"{name}.gx"]}
    for d in range(4):
        want[f"gw{d}"] = z[f"{name}.gw{d}"]
    pc.check_sga_compat(sim, DEV, x, gs, go, want)


@pytest.mark.parametrize("rowwave", [0, 1])
@pytest.mark.parametrize("shape", [(1, 2, 6, 3, 4), (1, 1, 20, 2, 20), (2, 1, 65, 2, 36), (1, 1, 130, 1, 8), (1, 2, 33, 3, 48)])
def test_horizontal_kernel_families(sim, port_oracle, shape, rowwave):
    """Both horizontal implementations (float4-per-lane segments / one wavefront per row with
    LDS-staged tiles) against the oracle, incl. partial batches and D > 64 (2+ disparities/lane)."""
    sim.set_option("GANET_SGA_ROWWAVE", rowwave)
    try:
        x, gs, go = pc.sga_inputs(shape, seed=11 + rowwave)
        err = pc.check_sga_forward_backward(sim, DEV, x, gs, go, _oracle_want(port_oracle, x, gs, go))
        assert max(err.values()) < 2e-5, err
    finally:
        sim.set_option("GANET_SGA_ROWWAVE", 1)


@pytest.mark.parametrize("D", [1, 2, 39, 40, 41, 47, 49, 57, 64, 66, 71, 72, 73])
def test_horizontal_depth_over_wavefront_boundaries(sim, port_oracle, D):
    """The row kernels carry the depth axis over the whole wavefront for D <= 40 (1 disparity per lane, 5 staged pieces),
    D <= 48 (1, 6), D <= 64 (1, 8) and D <= 72 (2, 9; even D: lanes wholly inside / outside) and in one mirrored 16-lane DPP row otherwise
    (sga_row_tu.hip): every boundary of that dispatch, rows of two batches with a partial one, against the oracle."""
    shape = (1, 2, D, 2, 40)
    x, gs, go = pc.sga_inputs(shape, seed=100 + D)
    err = pc.check_sga_forward_backward(sim, DEV, x, gs, go, _oracle_want(port_oracle, x, gs, go))
    assert max(err.values()) < 2e-5, err


@pytest.mark.parametrize("mode", ["plain", "guard_start", "late_reversed"])
@pytest.mark.parametrize("shape", [(1, 1, 192, 5, 16), (1, 2, 65, 7, 20), (2, 1, 7, 9, 8), (1, 1, 100, 6, 36), (1, 1, 150, 3, 12),
                                   (1, 1, 3, 1, 4), (1, 1, 191, 2, 40)])
def test_vertical_wide_column_blocks(sim, port_oracle, shape, mode):
    """GANET_SGA_WIDE_COL=2 (forced; automatic for inputs with few column blocks): the LDS-staged column blocks with one wavefront per column (1,024-thread blocks, 3 disparities per
    lane, D <= 192): down / up forward and adjoint scans bit-exact / within tolerance of the oracle, incl. partial column
    blocks, H not a multiple of the 4-row batch, D not a multiple of 3; also with buffers behind a guard page and with the
    emulator's reversed thread order."""
    sim.set_option("GANET_SGA_WIDE_COL", 2)
    if mode == "late_reversed":
        sim.set_option("HIPSIM_LANE_ORDER", 1)
    try:
        x, gs, go = pc.sga_inputs(shape, seed=sum(shape))
        dev = pc.NumpyDev("start" if mode == "guard_start" else "end")
        err = pc.check_sga_forward_backward(sim, dev, x, gs, go, _oracle_want(port_oracle, x, gs, go))
        assert max(err.values()) < 3e-5, err
    finally:
        sim.set_option("GANET_SGA_WIDE_COL", 1)
        sim.set_option("HIPSIM_LANE_ORDER", 0)


@pytest.mark.parametrize("colblock", [0, 1])
@pytest.mark.parametrize("shape", [(1, 2, 6, 3, 4), (2, 1, 20, 9, 20), (1, 1, 65, 5, 36), (1, 1, 130, 2, 8), (1, 2, 33, 13, 48)])
def test_vertical_kernel_families(sim, port_oracle, shape, colblock):
    """Register-only segment scans vs LDS-staged 16-column blocks (down / up), incl. partial
    column blocks, H not a multiple of the 4-row batch and D > 64."""
    sim.set_option("GANET_SGA_COLBLOCK", colblock)
    try:
        x, gs, go = pc.sga_inputs(shape, seed=31 + colblock)
        err = pc.check_sga_forward_backward(sim, DEV, x, gs, go, _oracle_want(port_oracle, x, gs, go))
        assert max(err.values()) < 2e-5, err
    finally:
        sim.set_option("GANET_SGA_COLBLOCK", 1)


def test_dpp_selftest(sim):
    scratch = np.zeros(8 * 64, np.int32)
    host = np.zeros(8 * 64, np.int32)
    sim.call("ganet_selftest_dpp", scratch.ctypes.data, host.ctypes.data, None)
    sim.call("ganet_selftest_dpp_wave", scratch.ctypes.data, host.ctypes.data, None)


def test_errors_are_reported(sim):
    from ganet_amd._native import GanetError
    x = np.zeros((1, 1, 1100, 1, 1), np.float32)
    g = np.zeros((1, 1, 5, 1, 1), np.float32)
    with pytest.raises(GanetError, match="exceeds"):          # 64 lanes x 17 disparities = 1,088 is the compiled maximum
        sim.call("ganet_sga_scan_forward", x.ctypes.data, g.ctypes.data, x.ctypes.data, 1, 1, 1100, 1, 1, 0, None)
    with pytest.raises(GanetError, match="null"):
        sim.call("ganet_sga_scan_forward", None, g.ctypes.data, x.ctypes.data, 1, 1, 3, 1, 1, 0, None)
    with pytest.raises(GanetError, match="dir"):
        sim.call("ganet_sga_scan_forward", x.ctypes.data, g.ctypes.data, x.ctypes.data, 1, 1, 3, 1, 1, 7, None)


@pytest.mark.parametrize("shape", [(1, 2, 6, 3, 4), (1, 1, 20, 2, 20), (2, 2, 3, 5, 36), (1, 1, 70, 2, 8),
                                   (1, 2, 48, 3, 12), (1, 1, 9, 33, 3)])
def test_forward_backward_vs_oracle_float4_rows(sim, port_oracle, shape):
    """W % 4 == 0 shapes take the float4 row kernels (multi-batch, partial last batch)."""
    x, gs, go = pc.sga_inputs(shape, seed=sum(shape))
    err = pc.check_sga_forward_backward(sim, DEV, x, gs, go, _oracle_want(port_oracle, x, gs, go))
    assert max(err.values()) < 2e-5, err


@pytest.mark.parametrize("shape", [(1, 1, 65, 5, 12), (1, 2, 193, 3, 8), (1, 1, 300, 3, 4), (1, 1, 49, 4, 7), (1, 1, 577, 2, 4)])
def test_wave_wide_scanlines(sim, port_oracle, shape):
    """GANET_SGA_WIDE_SCAN=2: the segment kernels with the WHOLE wavefront on one scanline (wave_shl / wave_shr DPP across the
    16-lane rows, row maxima / sums combined through v_readlane) -- what inputs with few scanlines and D > 272 use.
    Forward volumes / mask / arg-max bit-exact, gradients within 1e-4, including D = 300 and 577 (beyond the 16-lane limit)."""
    x, gs, go = pc.sga_inputs(shape, seed=sum(shape))
    out, tmp, mask = port_oracle.sga_forward(x, *gs)
    grads = port_oracle.sga_backward(x, *gs, tmp, mask, go)
    want = {"out": out, "mask": mask.astype(np.uint8), "tmp": tmp, "gx": grads[0]}
    for d in range(4):
        want[f"gw{d}"] = grads[1 + d]
    sim.set_option("GANET_SGA_WIDE_SCAN", 2)
    try:
        pc.check_sga_forward_backward(sim, DEV, x, gs, go, want)
        if shape[2] in (300, 49):
            pc.check_sga_compat(sim, DEV, x, gs, go, want)          # reference buffer contract (float mask) as well
    finally:
        sim.set_option("GANET_SGA_WIDE_SCAN", 1)


@pytest.mark.parametrize("tiled", [0, 1])
@pytest.mark.parametrize("shape", [(1, 2, 33, 8, 32), (1, 1, 65, 4, 48), (2, 1, 9, 12, 16), (1, 1, 20, 8, 80), (1, 3, 6, 4, 64)])
def test_tiled_private_workspace(sim, port_oracle, shape, tiled):
    """GANET_SGA_TILED (sga_col_kernels.h): the vertical directions' ADJOINT volumes of ganet_sga_backward's private workspace
    tiled [slice][W/16][H/4][D][4][16] -- written by the column adjoint scans as contiguous bursts, read by
    sga_bwd_point<.., TG>.  Both settings: same gradients (within 1e-4 of the oracle), and the tiled volume itself, un-tiled by
    the checker, equals what ganet_sga_backward_scan writes in the API layout.  Shapes: several row batches, one to five column
    blocks, depths that do and do not fill their lanes (GPU: the full cfg2 size as well)."""
    was = sim.get_option("GANET_SGA_TILED")
    sim.set_option("GANET_SGA_TILED", tiled)
    try:
        N, C, D, H, W = shape
        assert sim.query("ganet_sga_workspace_layout", N, C, D, H, W) == tiled
        x, gs, go = pc.sga_inputs(shape, seed=sum(shape) + tiled)
        err = pc.check_sga_forward_backward(sim, DEV, x, gs, go, _oracle_want(port_oracle, x, gs, go))
        assert max(err.values()) < 2e-5, err
    finally:
        sim.set_option("GANET_SGA_TILED", was)


def test_backward_with_a_4_byte_aligned_gradient_on_a_tiled_shape(sim, port_oracle):
    """on a shape where the private adjoint workspace is tiled (W % 16 == 0, H % 4 == 0), a gradient that is contiguous
    but only 4-byte aligned must take the API layout and the generic scans, not fail: ganet_sga_backward decides the layout
    once, from the dimensions, the options AND the alignment of what the tiled kernels would touch."""
    shape = (1, 1, 5, 4, 16)
    N, C, D, H, W = shape
    assert sim.query("ganet_sga_workspace_layout", N, C, D, H, W) == 1
    x, gs, go = pc.sga_inputs(shape, seed=3)
    got = pc.run_sga_backward_only(sim, pc.NumpyDev("end"), x, gs, go, go_offset=1)
    out, tmp, mask = port_oracle.sga_forward(x, *gs)
    grads = port_oracle.sga_backward(x, *gs, tmp, mask, go)
    assert np.abs(got["gx"] - grads[0]).max() <= pc.TOL
    for d in range(4):
        assert np.abs(got[f"gw{d}"] - grads[1 + d]).max() <= pc.TOL


def test_tiled_workspace_falls_back_where_it_does_not_apply(sim, port_oracle):
    """W % 16 != 0 or H % 4 != 0: the workspace keeps the API layout whatever the option says."""
    was = sim.get_option("GANET_SGA_TILED")
    sim.set_option("GANET_SGA_TILED", 1)
    try:
        for shape in [(1, 1, 9, 5, 32), (1, 1, 9, 8, 20)]:
            N, C, D, H, W = shape
            assert sim.query("ganet_sga_workspace_layout", N, C, D, H, W) == 0
            x, gs, go = pc.sga_inputs(shape, seed=3)
            err = pc.check_sga_forward_backward(sim, DEV, x, gs, go, _oracle_want(port_oracle, x, gs, go))
            assert max(err.values()) < 2e-5, err
    finally:
        sim.set_option("GANET_SGA_TILED", was)


def test_tiled_workspace_random_shapes(sim, port_oracle):
    """Seeded random shapes with W % 16 == 0 and H % 4 == 0 (one to four column blocks, one to four row batches, depths from one
    lane's worth to several, one to three slices): the tiled adjoint workspace against the oracle and against its own API-layout
    twin (parity_cases.check_sga_forward_backward does both)."""
    rng = np.random.default_rng(20260925)
    was = sim.get_option("GANET_SGA_TILED")
    sim.set_option("GANET_SGA_TILED", 1)
    try:
        for _ in range(6):
            shape = (int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(2, 40)), 4 * int(rng.integers(1, 5)), 16 * int(rng.integers(1, 5)))
            N, C, D, H, W = shape
            assert sim.query("ganet_sga_workspace_layout", N, C, D, H, W) == 1
            x, gs, go = pc.sga_inputs(shape, seed=int(rng.integers(1 << 30)))
            err = pc.check_sga_forward_backward(sim, DEV, x, gs, go, _oracle_want(port_oracle, x, gs, go))
            assert max(err.values()) < 3e-5, (shape, err)
    finally:
        sim.set_option("GANET_SGA_TILED", was)
