"""Bounds sweeps on the CPU emulator: every buffer ends at (or begins behind) an inaccessible page
(parity_cases.guarded_empty), so an access outside a tensor is a crash, whatever the values read would have been used for.

Why: the plane-pair LGA kernels of round 2 requested one plane past the end of the volume in their last steady step -- the
data was never used, every parity test passed, and the GPU faulted only once a plane spanned whole pages and the tensor
happened to be the last one of a mapped range (528x960, inside a training step).  The depth sweeps below walk every phase of
the march against its ring / look-ahead / steady-body boundaries; the results are compared with the oracle as well."""
import numpy as np
import pytest

import parity_cases as pc


@pytest.fixture(scope="module")
def sim():
    from sim_util import sim_api
    return sim_api()


@pytest.mark.parametrize("guard", ["end", "start"])
@pytest.mark.parametrize("wave,segs,mix", [(2, 0, 0), (2, 2, 0), (2, 0, 3), (1, 0, 0), (1, 2, 0), (1, 0, 3), (0, 0, 0)])
def test_lga_depth_sweep_guarded(sim, port_oracle, wave, segs, mix, guard):
    """plane-pair kernels (whole tiles, two depth segments, the mixed item list with three SIMDs assumed) and the tile
    kernels, every depth 1..27"""
    dev = pc.NumpyDev(guard)
    sim.set_option("GANET_LGA_WAVE", wave)
    sim.set_option("GANET_LGA_SEGS", segs)
    sim.set_option("GANET_LGA_MIX", mix)
    try:
        for D in range(1, 28):
            for B, H, W in ((1, 3, 36),) + (((2, 2, 7),) if D % 4 == 1 else ()):
                rng = np.random.default_rng(100 * D + W)
                x = rng.standard_normal((B, D, H, W)).astype(np.float32)
                f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
                gy = rng.standard_normal((B, D, H, W)).astype(np.float32)
                y = port_oracle.lga_forward(x, f, 2)
                gx, gf = port_oracle.lga_backward(x, f, gy, 2)
                err = pc.check_lga_chain(sim, dev, x, f, gy, 2, 1, {"y": y, "gx": gx, "gf": gf})
                assert max(err.values()) < 2e-5, (D, B, H, W, err)
    finally:
        sim.set_option("GANET_LGA_WAVE", 2)
        sim.set_option("GANET_LGA_SEGS", 0)
        sim.set_option("GANET_LGA_MIX", 1)


@pytest.mark.parametrize("guard", ["end", "start"])
@pytest.mark.parametrize("r", [1, 3])
def test_lga_other_radii_guarded(sim, port_oracle, r, guard):
    dev = pc.NumpyDev(guard)
    for D in (1, 2, 9, 12, 13):
        rng = np.random.default_rng(D + r)
        shape = (1, D, 4, 34)
        x = rng.standard_normal(shape).astype(np.float32)
        f = pc.l1norm(rng.standard_normal((1, 3 * (2 * r + 1) ** 2, 4, 34)), 1)
        gy = rng.standard_normal(shape).astype(np.float32)
        y = port_oracle.lga_forward(x, f, r)
        gx, gf = port_oracle.lga_backward(x, f, gy, r)
        err = pc.check_lga_chain(sim, dev, x, f, gy, r, 1, {"y": y, "gx": gx, "gf": gf})
        assert max(err.values()) < 2e-5, (D, err)


def _sga_want(oracle, x, gs, go):
    out, tmp, mask = oracle.sga_forward(x, *gs)
    grads = oracle.sga_backward(x, *gs, tmp, mask, go)
    want = {"out": out, "mask": mask.astype(np.uint8), "tmp": tmp, "gx": grads[0]}
    for d in range(4):
        want[f"gw{d}"] = grads[1 + d]
    return want


# column-block kernels (W % 4 == 0, any H), row kernels (W % 4 == 0), strided fallbacks (odd W), D across the lane-group
# sizes (<= 16 DPL 1, 17..80 DPL <= 5, > 80), single rows / columns
@pytest.mark.parametrize("guard", ["end", "start"])
@pytest.mark.parametrize("shape", [(1, 1, 1, 1, 4), (1, 2, 3, 2, 8), (1, 1, 16, 5, 16), (2, 1, 17, 3, 20), (1, 1, 33, 7, 12),
                                   (1, 2, 65, 2, 36), (1, 1, 81, 3, 8), (1, 1, 5, 4, 7), (1, 1, 9, 1, 33), (1, 1, 7, 33, 1),
                                   (1, 3, 4, 17, 32), (1, 1, 48, 9, 48)])
def test_sga_shapes_guarded(sim, port_oracle, shape, guard):
    x, gs, go = pc.sga_inputs(shape, seed=sum(shape))
    pc.check_sga_forward_backward(sim, pc.NumpyDev(guard), x, gs, go, _sga_want(port_oracle, x, gs, go))


# ---- hand-counted s_waitcnt vmcnt(n) under the emulator's LATE-LANDING copy model --------------------------------------
# (tests/hipsim/hipsim.h: a global -> LDS copy lands only when a wait of its lane leaves no room for it any more, i.e. as late
# as the kernel's own waits permit; a thread that ends with copies in flight aborts).  A count that is one too loose, a wait
# placed behind the first read of a slot, or a missing final wait gives wrong results / an abort HERE instead of a
# timing-dependent stale read on the GPU ("the emulator compiles GA_VMCNT to a no-op").
@pytest.mark.parametrize("segs,mix,paired", [(0, 0, 0), (2, 0, 0), (3, 0, 0), (0, 2, 0), (0, 0, 1), (0, 3, 1), (2, 0, 1)])
def test_lga_counted_waits_with_late_landing_copies(sim, port_oracle, segs, mix, paired):
    dev = pc.NumpyDev()
    sim.set_option("GANET_LGA_SEGS", segs)
    sim.set_option("GANET_LGA_MIX", mix)
    sim.set_option("HIPSIM_LATE_DMA", 1)
    try:
        for shape in [(1, D, 3, 36) for D in (1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 19, 20, 21, 26, 27, 40, 41)] + \
                     [(2, 13, 5, 68), (1, 22, 2, 8), (1, 15, 9, 40), (1, 33, 1, 4)]:
            rng = np.random.default_rng(sum(shape))
            B, D, H, W = shape
            x = rng.standard_normal(shape).astype(np.float32)
            f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
            gy = rng.standard_normal(shape).astype(np.float32)
            y, ins = port_oracle.lga_chain_forward(x, f, 2, 2)
            gx, gf = port_oracle.lga_chain_backward(ins, f, gy, 2)
            chain = pc.check_lga2_paired if paired and W % 2 == 0 else pc.check_lga_chain
            err = chain(sim, dev, x, f, gy, 2, 2, {"y": y, "gx": gx, "gf": gf})
            assert max(err.values()) < 5e-5, (shape, err)
    finally:
        sim.set_option("HIPSIM_LATE_DMA", 0)
        sim.set_option("GANET_LGA_SEGS", 0)
        sim.set_option("GANET_LGA_MIX", 1)


# ---- the planar staging's asm LDS reads (ds_read2_b32) and their counted lgkmcnt waits ---------------------------------------
# the default LGA kernels for W % 4 == 0 read their window rows with ds_read2_b32 written in asm, which the
# compiler does not wait for; the kernel's own s_waitcnt lgkmcnt(LA * 5) -- now the first instruction of the row's FMA statement
# -- does.  HIPSIM_LATE_LDS: such a read lands in its registers only when a wait of the lane covers it (they hold NaNs until
# then).  All shapes below have W % 4 == 0, i.e. run the planar instantiations (lga_apply_pp_x / _xo, lga_filter_grad_pp_x /
# _gypx) of both call sequences, copies landing late as well.
_PLANAR_SHAPES = [(1, D, 3, 36) for D in (1, 2, 3, 8, 9, 13, 14, 21, 26, 27)] + [(2, 13, 5, 68), (1, 22, 2, 8), (1, 15, 9, 40), (1, 33, 1, 4)]


def _planar_chain(sim, port_oracle, shape, paired):
    dev = pc.NumpyDev()
    rng = np.random.default_rng(sum(shape))
    B, D, H, W = shape
    x = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
    gy = rng.standard_normal(shape).astype(np.float32)
    y, ins = port_oracle.lga_chain_forward(x, f, 2, 2)
    gx, gf = port_oracle.lga_chain_backward(ins, f, gy, 2)
    chain = pc.check_lga2_paired if paired else pc.check_lga_chain
    return chain(sim, dev, x, f, gy, 2, 2, {"y": y, "gx": gx, "gf": gf})


@pytest.mark.parametrize("paired,segs,mix", [(0, 0, 1), (1, 0, 1), (0, 2, 0), (1, 3, 0), (1, 0, 2)])
def test_lga_planar_reads_with_late_landing_lds(sim, port_oracle, paired, segs, mix):
    """(segs / mix: whole tiles, every tile cut into depth segments, the mixed item list -- a segment starts and ends its ring,
    its row look-ahead and its counted waits somewhere inside the volume)"""
    sim.set_option("GANET_LGA_SEGS", segs)
    sim.set_option("GANET_LGA_MIX", mix)
    sim.set_option("HIPSIM_LATE_LDS", 1)
    sim.set_option("HIPSIM_LATE_DMA", 1)
    try:
        for shape in _PLANAR_SHAPES:
            err = _planar_chain(sim, port_oracle, shape, paired)
            assert max(err.values()) < 5e-5, (shape, err)
    finally:
        sim.set_option("HIPSIM_LATE_LDS", 0)
        sim.set_option("HIPSIM_LATE_DMA", 0)
        sim.set_option("GANET_LGA_SEGS", 0)
        sim.set_option("GANET_LGA_MIX", 1)


@pytest.mark.parametrize("which,slack", [("HIPSIM_LGKM_SLACK", 1), ("HIPSIM_VMCNT_SLACK", 1)])
def test_a_counted_wait_loosened_by_one_fails(sim, port_oracle, which, slack):
    """The point of the late-landing models: with every counted wait of one kind one operation too loose, the same chain
    must NOT reproduce the oracle (NaNs from unlanded LDS reads / stale ring slots from unlanded copies)."""
    sim.set_option("HIPSIM_LATE_LDS", 1)
    sim.set_option("HIPSIM_LATE_DMA", 1)
    sim.set_option(which, slack)
    try:
        bad = 0
        for shape in [(1, 13, 3, 36), (1, 26, 3, 36), (2, 13, 5, 68)]:
            try:
                err = _planar_chain(sim, port_oracle, shape, 1)
                bad += not (max(err.values()) < 5e-5)
            except AssertionError:
                bad += 1
        assert bad > 0, f"{which}={slack} went unnoticed"
    finally:
        sim.set_option(which, 0)
        sim.set_option("HIPSIM_LATE_LDS", 0)
        sim.set_option("HIPSIM_LATE_DMA", 0)


# ---- thread scheduling order of the emulator ----------------------------------------------------------------------------
# Between two barriers the emulator runs the threads of a block one after the other; a hand-off through LDS that lacks a
# barrier (or a wave barrier where a workgroup barrier is needed) is then decided by the order.  Everything below also
# runs with the threads resumed in DESCENDING order (and the copies landing late): same results required.
@pytest.fixture()
def reversed_lanes(sim):
    sim.set_option("HIPSIM_LANE_ORDER", 1)
    sim.set_option("HIPSIM_LATE_DMA", 1)
    yield
    sim.set_option("HIPSIM_LANE_ORDER", 0)
    sim.set_option("HIPSIM_LATE_DMA", 0)


@pytest.mark.parametrize("wave,segs", [(2, 0), (2, 2), (1, 0), (1, 2), (0, 0)])
def test_lga_families_with_reversed_thread_order(sim, port_oracle, reversed_lanes, wave, segs):
    dev = pc.NumpyDev()
    sim.set_option("GANET_LGA_WAVE", wave)
    sim.set_option("GANET_LGA_SEGS", segs)
    try:
        for shape in [(1, 1, 3, 36), (1, 9, 3, 36), (1, 12, 4, 40), (2, 21, 5, 68), (1, 26, 2, 7), (1, 14, 9, 34)]:
            rng = np.random.default_rng(sum(shape))
            B, D, H, W = shape
            x = rng.standard_normal(shape).astype(np.float32)
            f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
            gy = rng.standard_normal(shape).astype(np.float32)
            y, ins = port_oracle.lga_chain_forward(x, f, 2, 2)
            gx, gf = port_oracle.lga_chain_backward(ins, f, gy, 2)
            err = pc.check_lga_chain(sim, dev, x, f, gy, 2, 2, {"y": y, "gx": gx, "gf": gf})
            assert max(err.values()) < 2e-5, (shape, err)
    finally:
        sim.set_option("GANET_LGA_WAVE", 2)
        sim.set_option("GANET_LGA_SEGS", 0)


@pytest.mark.parametrize("shape", [(1, 2, 3, 2, 8), (1, 1, 16, 5, 16), (2, 1, 17, 3, 20), (1, 2, 65, 2, 36), (1, 1, 81, 3, 8),
                                   (1, 1, 5, 4, 7), (1, 3, 4, 17, 32), (1, 1, 48, 9, 48), (1, 1, 300, 3, 8)])
def test_sga_with_reversed_thread_order(sim, port_oracle, reversed_lanes, shape):
    x, gs, go = pc.sga_inputs(shape, seed=sum(shape))
    pc.check_sga_forward_backward(sim, pc.NumpyDev(), x, gs, go, _sga_want(port_oracle, x, gs, go))


# ---- the workgroup-shared ring (GANET_LGA_WAVE=2: lga_apply_pp_wx / _wxo / _wpi, four waves on a 32 x 8 tile) ------------------------
# What is new against the one-wave kernels is a hand-off BETWEEN waves: every wave stages a quarter of each ring slot, waits for
# its own copies (counted) and meets the others at one workgroup barrier per pair-step.  The emulator's wave barrier covers the
# caller's wavefront only, its copies land as late as the issuing lane's own waits allow, and threads between two barriers run
# one after the other: a missing or misplaced workgroup barrier reads a slot quarter another wave has not even requested.
_WG_SHAPES = [(1, D, 11, 36) for D in (1, 2, 3, 8, 9, 13, 14, 21, 26, 27)] + \
             [(2, 13, 5, 68), (1, 22, 2, 8), (1, 15, 9, 40), (1, 33, 1, 4), (1, 12, 8, 32), (1, 10, 17, 64), (1, 7, 16, 4), (1, 40, 9, 36),
              (1, 41, 3, 36), (1, 61, 8, 32)]      # (D >= 34: the filter gradient's steady groups of LGAP_WG_NR = 8 steps; 61: two of them)


# Both forms: the workgroup rings (GANET_LGA_WAVE=2, the default since round 5) and the one-wave rings they fall back to (1).
@pytest.fixture(params=[1, 0], ids=["workgroup-ring", "one-wave-ring"])
def wg_ring(sim, request):
    sim.set_option("GANET_LGA_WAVE", 1 + request.param)
    yield request.param
    sim.set_option("GANET_LGA_WAVE", 2)


@pytest.mark.parametrize("guard", ["end", "start"])
@pytest.mark.parametrize("paired,segs,mix", [(0, 0, 1), (1, 0, 1), (0, 2, 0), (1, 3, 0), (1, 0, 2), (0, 0, 3)])
def test_lga_workgroup_ring_late_landing_guarded(sim, port_oracle, wg_ring, paired, segs, mix, guard):
    sim.set_option("GANET_LGA_SEGS", segs)
    sim.set_option("GANET_LGA_MIX", mix)
    sim.set_option("HIPSIM_LATE_LDS", 1)
    sim.set_option("HIPSIM_LATE_DMA", 1)
    try:
        for shape in _WG_SHAPES:
            dev = pc.NumpyDev(guard)
            rng = np.random.default_rng(sum(shape))
            B, D, H, W = shape
            x = rng.standard_normal(shape).astype(np.float32)
            f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
            gy = rng.standard_normal(shape).astype(np.float32)
            y, ins = port_oracle.lga_chain_forward(x, f, 2, 2)
            gx, gf = port_oracle.lga_chain_backward(ins, f, gy, 2)
            chain = pc.check_lga2_paired if paired else pc.check_lga_chain
            err = chain(sim, dev, x, f, gy, 2, 2, {"y": y, "gx": gx, "gf": gf})
            assert max(err.values()) < 5e-5, (shape, err)
    finally:
        sim.set_option("HIPSIM_LATE_LDS", 0)
        sim.set_option("HIPSIM_LATE_DMA", 0)
        sim.set_option("GANET_LGA_SEGS", 0)
        sim.set_option("GANET_LGA_MIX", 1)


def test_lga_workgroup_ring_is_what_runs(sim, wg_ring):
    """the option reaches the launcher (a test of the one-wave kernels under another name would prove nothing)"""
    assert sim.get_option("GANET_LGA_WAVE") == 1 + wg_ring


@pytest.mark.parametrize("paired", [0, 1])
def test_lga_workgroup_ring_reversed_thread_order(sim, port_oracle, wg_ring, reversed_lanes, paired):
    for shape in [(1, 9, 11, 36), (2, 21, 5, 68), (1, 14, 9, 40), (1, 26, 16, 32)]:
        err = _planar_chain(sim, port_oracle, shape, paired)
        assert max(err.values()) < 5e-5, (shape, err)


# HIPSIM_WAVE_GREEDY: one wavefront runs until every one of its threads waits for another wavefront, then the next -- the first
# (with the reversed order: the last) wave of a workgroup is as far ahead of the others as the synchronisation permits.  Resuming
# all threads in turn, which is what the emulator does otherwise, keeps the waves together and would hide a hand-off that
# lets one of them run too far ahead.
@pytest.fixture(params=[0, 1], ids=["first-wave-ahead", "last-wave-ahead"])
def greedy_waves(sim, request):
    sim.set_option("HIPSIM_WAVE_GREEDY", 1)
    sim.set_option("HIPSIM_LANE_ORDER", request.param)
    sim.set_option("HIPSIM_LATE_DMA", 1)
    sim.set_option("HIPSIM_LATE_LDS", 1)
    yield
    sim.set_option("HIPSIM_WAVE_GREEDY", 0)
    sim.set_option("HIPSIM_LANE_ORDER", 0)
    sim.set_option("HIPSIM_LATE_DMA", 0)
    sim.set_option("HIPSIM_LATE_LDS", 0)


@pytest.mark.parametrize("paired,segs", [(0, 0), (1, 0), (1, 2)])
def test_lga_workgroup_ring_with_one_wave_running_ahead(sim, port_oracle, wg_ring, greedy_waves, paired, segs):
    sim.set_option("GANET_LGA_SEGS", segs)
    try:
        for shape in [(1, 9, 11, 36), (2, 21, 5, 68), (1, 14, 9, 40), (1, 41, 16, 32), (1, 1, 8, 4), (1, 26, 3, 36)]:
            err = _planar_chain(sim, port_oracle, shape, paired)
            assert max(err.values()) < 5e-5, (shape, err)
    finally:
        sim.set_option("GANET_LGA_SEGS", 0)


def test_other_multi_wave_kernels_with_one_wave_running_ahead(sim, port_oracle, greedy_waves):
    """the 256-thread LGA tile kernels (GANET_LGA_WAVE=0) and the SGA kernels with several waves per block, under the same schedule"""
    sim.set_option("GANET_LGA_WAVE", 0)
    try:
        for shape in [(1, 9, 11, 36), (2, 13, 5, 68)]:
            err = _planar_chain(sim, port_oracle, shape, 0)
            assert max(err.values()) < 5e-5, (shape, err)
    finally:
        sim.set_option("GANET_LGA_WAVE", 2)
    for shape in [(1, 2, 12, 9, 32), (1, 1, 33, 7, 12), (1, 1, 48, 9, 48), (2, 1, 17, 3, 20)]:
        x, gs, go = pc.sga_inputs(shape, seed=sum(shape))
        pc.check_sga_forward_backward(sim, pc.NumpyDev(), x, gs, go, _sga_want(port_oracle, x, gs, go))


def test_lga_workgroup_ring_loosened_wait_fails(sim, port_oracle):
    """one copy too many left in flight at the per-step wait: the barrier then publishes a slot quarter that has not landed"""
    assert sim.get_option("GANET_LGA_WAVE") == 2
    sim.set_option("HIPSIM_LATE_DMA", 1)
    sim.set_option("HIPSIM_VMCNT_SLACK", 1)
    try:
        bad = 0
        for shape in [(1, 13, 11, 36), (1, 26, 8, 36), (2, 13, 5, 68)]:
            try:
                err = _planar_chain(sim, port_oracle, shape, 1)
                bad += not (max(err.values()) < 5e-5)
            except AssertionError:
                bad += 1
        assert bad > 0, "HIPSIM_VMCNT_SLACK=1 went unnoticed by the workgroup ring"
    finally:
        sim.set_option("HIPSIM_VMCNT_SLACK", 0)
        sim.set_option("HIPSIM_LATE_DMA", 0)


@pytest.mark.parametrize("paired", [0, 1])
def test_lga_workgroup_ring_bit_identical_to_one_wave_kernels(sim, port_oracle, paired):
    """same arithmetic per pixel in the same order: on whole tiles the two forms agree bit for bit"""
    sim.set_option("GANET_LGA_MIX", 0)
    sim.set_option("GANET_LGA_SEGS", 1)
    try:
        for shape in [(1, 13, 11, 36), (2, 8, 5, 68), (1, 27, 16, 32)]:
            rng = np.random.default_rng(sum(shape))
            B, D, H, W = shape
            x = rng.standard_normal(shape).astype(np.float32)
            f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
            gy = rng.standard_normal(shape).astype(np.float32)
            res = []
            for wg in (0, 1):
                sim.set_option("GANET_LGA_WAVE", 1 + wg)
                got = {}
                (pc.check_lga2_paired if paired else pc.check_lga_chain)(sim, pc.NumpyDev(), x, f, gy, 2, 2, None, out=got)
                res.append(got)
            for k in res[0]:
                assert np.array_equal(res[0][k], res[1][k]), (shape, k)
    finally:
        sim.set_option("GANET_LGA_WAVE", 2)
        sim.set_option("GANET_LGA_MIX", 1)
        sim.set_option("GANET_LGA_SEGS", 0)




# ---- the filter gradient with a tile's taps split over a wave pair (GANET_LGA_WAVE=2, x pair-interleaved: lga_filter_grad_pair.inc) ------
# Two waves on the same 64 pixels share one x ring and one gy ring; each issues half of a step's copies, waits for ITS half with a counted
# vmcnt and meets the other at one workgroup barrier per pair-step.  The chains above run it wherever they take the pair-interleaved
# intermediate under GANET_LGA_WAVE=2; here it is taken alone: every schedule the emulator has, and the counted wait must be tight.
def _fg_pair_once(sim, port_oracle, shape, acc_twice=True):
    B, D, H, W = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
    _, want = port_oracle.lga_backward(x, f, gy, 2)
    dev = pc.NumpyDev("end")
    xp, dgy, gf = dev.to(pc.to_paired(x)), dev.to(gy), dev.empty(f.shape)
    sim.call("ganet_lga_filter_grad_paired", dev.ptr(xp), dev.ptr(dgy), dev.ptr(gf), B, D, H, W, 2, 0, 1, 0, None)
    e = float(np.abs(gf - want).max()) if np.isfinite(gf).all() else float("inf")
    if acc_twice:
        sim.call("ganet_lga_filter_grad_paired", dev.ptr(xp), dev.ptr(dgy), dev.ptr(gf), B, D, H, W, 2, 1, 1, 0, None)
        e = max(e, 0.5 * float(np.abs(gf - 2 * want).max()) if np.isfinite(gf).all() else float("inf"))
    return e


_FG_PAIR_SHAPES = [(1, D, 3, 36) for D in (1, 2, 3, 4, 9, 10, 11, 12, 13, 21, 22)] + \
                  [(2, 21, 5, 68), (1, 26, 2, 6), (1, 41, 7, 64), (1, 9, 7, 100), (4, 2, 2, 2), (1, 33, 6, 72), (3, 12, 4, 40)]


@pytest.mark.parametrize("mode", ["plain", "late", "late_reversed", "first_wave_ahead", "last_wave_ahead"])
def test_filter_gradient_wave_pair_under_every_schedule(sim, port_oracle, mode):
    opts = {"plain": {}, "late": {"HIPSIM_LATE_DMA": 1}, "late_reversed": {"HIPSIM_LATE_DMA": 1, "HIPSIM_LANE_ORDER": 1},
            "first_wave_ahead": {"HIPSIM_LATE_DMA": 1, "HIPSIM_WAVE_GREEDY": 1},
            "last_wave_ahead": {"HIPSIM_LATE_DMA": 1, "HIPSIM_WAVE_GREEDY": 1, "HIPSIM_LANE_ORDER": 1}}[mode]
    assert sim.get_option("GANET_LGA_WAVE") == 2
    for k, v in opts.items():
        sim.set_option(k, v)
    try:
        for shape in _FG_PAIR_SHAPES:
            assert _fg_pair_once(sim, port_oracle, shape) < 3e-5, shape
    finally:
        for k in opts:
            sim.set_option(k, 0)


def test_filter_gradient_wave_pair_equals_one_wave_kernel_bit_for_bit(sim, port_oracle):
    """same FMAs per tap in the same order, the centre sums handed over exactly: the two forms agree to the bit"""
    res = []
    try:
        for wave in (1, 2):
            sim.set_option("GANET_LGA_WAVE", wave)
            out = []
            for shape in [(1, 13, 3, 36), (2, 21, 5, 68), (1, 41, 7, 64), (1, 9, 7, 100)]:
                B, D, H, W = shape
                rng = np.random.default_rng(sum(shape))
                x = rng.standard_normal(shape).astype(np.float32)
                gy = rng.standard_normal(shape).astype(np.float32)
                dev = pc.NumpyDev("end")
                xp, dgy, gf = dev.to(pc.to_paired(x)), dev.to(gy), dev.empty((B, 75, H, W))
                sim.call("ganet_lga_filter_grad_paired", dev.ptr(xp), dev.ptr(dgy), dev.ptr(gf), B, D, H, W, 2, 0, 1, 0, None)
                out.append(np.array(gf))
            res.append(out)
    finally:
        sim.set_option("GANET_LGA_WAVE", 2)
    for a, b in zip(*res):
        assert np.array_equal(a, b)


def test_filter_gradient_wave_pair_counted_wait_is_tight(sim, port_oracle):
    """one operation of slack in the counted vmcnt and the late-landing copies are read before they arrive"""
    sim.set_option("HIPSIM_LATE_DMA", 1)
    try:
        bad = {}
        for slack in (0, 1):
            sim.set_option("HIPSIM_VMCNT_SLACK", slack)
            n = 0
            for shape in [(1, 41, 7, 64), (1, 33, 6, 72), (1, 21, 3, 36)]:
                try:
                    n += not (_fg_pair_once(sim, port_oracle, shape, acc_twice=False) < 3e-5)
                except Exception:          # noqa: BLE001  (the emulator aborts a read of an unlanded slot in some modes)
                    n += 1
            bad[slack] = n
        assert bad[0] == 0 and bad[1] > 0, bad
    finally:
        sim.set_option("HIPSIM_VMCNT_SLACK", 0)
        sim.set_option("HIPSIM_LATE_DMA", 0)
