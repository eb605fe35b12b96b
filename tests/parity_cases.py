"""Parity checks shared by the CPU-emulator tests (numpy buffers, tests/hipsim) and the
GPU tests (torch device buffers): every check drives the C ABI of include/ganet_hip.h
and compares with the oracle / golden fixtures.  `dev` abstracts buffer handling."""
import ctypes
import mmap

import os

import numpy as np

TOL = 1e-4   # north_star: fp32 max-abs <= 1e-4 for SGA/LGA forward + backward


def l1norm(g, axis):
    return (g / np.abs(g).sum(axis, keepdims=True)).astype(np.float32)


def sga_inputs(shape, seed):
    rng = np.random.default_rng(seed)
    N, C, D, H, W = shape
    x = rng.standard_normal(shape).astype(np.float32)
    gs = [l1norm(rng.standard_normal((N, C, 5, H, W)), 2) for _ in range(4)]
    go = rng.standard_normal(shape).astype(np.float32)
    return x, gs, go


_PAGE = mmap.PAGESIZE
_libc = ctypes.CDLL(None, use_errno=True)
_libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def guarded_empty(shape, dtype=np.float32, guard="end"):
    """An array whose last byte is followed by an inaccessible page (and whose mapping is preceded by one): a kernel that
    reads or writes past the end of a tensor -- harmless on the GPU as long as the neighbouring memory happens to be mapped,
    a memory access fault once it is not (the 528x960 volumes of round 2) -- dies right here on the emulator.
    guard="start": the array BEGINS right behind the leading inaccessible page instead (accesses in front of a tensor)."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64))
    nbytes = count * dtype.itemsize
    body = -(-max(nbytes, 1) // _PAGE) * _PAGE
    mm = mmap.mmap(-1, body + 2 * _PAGE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    for off in (0, _PAGE + body):
        if _libc.mprotect(base + off, _PAGE, 0) != 0:
            raise OSError(ctypes.get_errno(), "mprotect")
    start = _PAGE if guard == "start" else (_PAGE + body - nbytes) & ~15     # 16-byte aligned like any device allocation
    return np.frombuffer(mm, dtype, count, start).reshape(shape)


class NumpyDev:
    """Host buffers for the emulator build (every buffer ends at a guard page, see guarded_empty)."""
    stream = None

    def __init__(self, guard="end"):
        self.guard = guard

    def to(self, a):
        a = np.asarray(a)
        g = guarded_empty(a.shape, a.dtype, self.guard)
        g[...] = a
        return g

    def empty(self, shape, dtype=np.float32):
        a = guarded_empty(shape, dtype, self.guard)
        a.fill(np.nan if dtype == np.float32 else 113)     # poison: unwritten elements get noticed
        return a

    def zeros(self, shape, dtype=np.float32):
        a = guarded_empty(shape, dtype, self.guard)
        a.fill(0)
        return a

    def ptr(self, a):
        return a.ctypes.data

    def host(self, a):
        return a

    def sync(self):
        pass


class _PtrAt:
    """a device address inside a buffer that is kept alive alongside it"""
    def __init__(self, addr, keep):
        self.addr, self.keep = addr, keep


class _OffsetDev:
    """a device whose ptr() also understands _PtrAt"""
    def __init__(self, dev):
        self._dev = dev

    def __getattr__(self, name):
        return getattr(self._dev, name)

    def ptr(self, a):
        return a.addr if isinstance(a, _PtrAt) else self._dev.ptr(a)


def check_sga_scan(api, dev, oracle, x, g, direction):
    N, C, D, H, W = x.shape
    dx, dg = dev.to(x), dev.to(g)
    A = dev.empty(x.shape)
    api.call("ganet_sga_scan_forward", dev.ptr(dx), dev.ptr(dg), dev.ptr(A), N, C, D, H, W, direction, dev.stream)
    dev.sync()
    want = oracle.sga_scan(x, g, direction)
    got = dev.host(A)
    assert np.array_equal(got, want), f"dir {direction}: {int((got != want).sum())} of {got.size} differ, max {np.abs(got - want).max()}"


def untile_ws(t):
    """[N,C,D,H,W] array holding a vertical direction's volume in SgaFunction's private tiled layout
    ([slice][W/16][H/4][D][4][16], include/ganet_hip.h: ganet_sga_workspace_layout) -> the API layout."""
    N, C, D, H, W = t.shape
    v = t.reshape(N * C, W // 16, H // 4, D, 4, 16)          # (s, cb, rb, d, rj, cw)
    return np.ascontiguousarray(v.transpose(0, 3, 2, 4, 1, 5)).reshape(N, C, D, H, W)


def run_sga_forward(api, dev, x, gs):
    N, C, D, H, W = x.shape
    dx = dev.to(x)
    dg = [dev.to(g) for g in gs]
    A = dev.empty((4,) + x.shape)
    out = dev.empty(x.shape)
    mask = dev.empty(x.shape, np.uint8)
    kp = dev.empty((4, N, C, H, W), np.uint16)
    api.call("ganet_sga_forward", dev.ptr(dx), *[dev.ptr(g) for g in dg], dev.ptr(A), dev.ptr(out), dev.ptr(mask),
             dev.ptr(kp), N, C, D, H, W, dev.stream)
    dev.sync()
    return dx, dg, A, out, mask, kp


def check_sga_forward_backward(api, dev, x, gs, go, want, per_dir=True, results=None):
    """want: dict(out, mask(uint8), gx, gw0..gw3, optional A0..A3) from the oracle/golden.
    per_dir=False skips the cross-check of the per-direction entry point (large volumes)."""
    N, C, D, H, W = x.shape
    dx, dg, A, out, mask, kp = run_sga_forward(api, dev, x, gs)
    hA = dev.host(A)
    for d in range(4):
        if f"A{d}" in want:
            assert np.array_equal(hA[d], want[f"A{d}"]), f"A{d}"
    assert np.array_equal(dev.host(out), want["out"]), "out"
    assert np.array_equal(dev.host(mask), want["mask"]), "direction mask must be bit-exact"
    # arg-max indices: first maximum over d of each directional volume (MaxDepth semantics)
    assert np.array_equal(dev.host(kp).astype(np.int64), np.argmax(hA, axis=3)), "arg-max must be bit-exact"
    dgo = dev.to(go)
    gx = dev.empty(x.shape)
    gw = [dev.empty(gs[0].shape) for _ in range(4)]
    G = dev.empty((4,) + x.shape)
    api.call("ganet_sga_backward", dev.ptr(dx), *[dev.ptr(g) for g in dg], dev.ptr(A), dev.ptr(mask), dev.ptr(kp),
             dev.ptr(dgo), dev.ptr(G), dev.ptr(gx), *[dev.ptr(g) for g in gw], N, C, D, H, W, dev.stream)
    dev.sync()
    # the per-direction entry point must agree with the fused one
    gx1 = dev.empty(x.shape) if per_dir else gx
    G1 = dev.empty(x.shape) if per_dir else None
    for d in range(4 if per_dir else 0):
        gw1 = dev.empty(gs[0].shape)
        api.call("ganet_sga_backward_dir", dev.ptr(dx), dev.ptr(dg[d]), dev.ptr(A) + 4 * d * x.size,
                 dev.ptr(mask), dev.ptr(kp) + 2 * d * (N * C * H * W), dev.ptr(dgo), dev.ptr(G1), dev.ptr(gx1),
                 dev.ptr(gw1), N, C, D, H, W, d, 1 if d else 0, dev.stream)
        dev.sync()
        assert np.abs(dev.host(gw1) - dev.host(gw[d])).max() <= 1e-6
    assert np.abs(dev.host(gx1) - dev.host(gx)).max() <= 1e-5
    if per_dir:
        # ABI 8: the composite entries are their steps -- 4 x scan + ganet_sga_merge, 4 x adjoint scan +
        # ganet_sga_backward_point -- bit for bit (bench.py times the steps in place)
        A2, out2 = dev.empty((4,) + x.shape), dev.empty(x.shape)
        mask2, kp2 = dev.empty(x.shape, np.uint8), dev.empty((4, N, C, H, W), np.uint16)
        for d in range(4):
            api.call("ganet_sga_scan_forward_ws", dev.ptr(dx), dev.ptr(dg[d]), dev.ptr(A2), N, C, D, H, W, d, dev.stream)
        api.call("ganet_sga_merge", dev.ptr(A2), dev.ptr(out2), dev.ptr(mask2), dev.ptr(kp2), N, C, D, H, W, dev.stream)
        dev.sync()
        assert np.array_equal(dev.host(out2), dev.host(out)) and np.array_equal(dev.host(mask2), dev.host(mask))
        assert np.array_equal(dev.host(kp2), dev.host(kp))
        G2, gx2 = dev.empty((4,) + x.shape), dev.empty(x.shape)
        gw2 = [dev.empty(gs[0].shape) for _ in range(4)]
        for d in range(4):
            api.call("ganet_sga_backward_scan_ws", dev.ptr(dg[d]), dev.ptr(mask), dev.ptr(kp), dev.ptr(dgo),
                     dev.ptr(G2), N, C, D, H, W, d, dev.stream)
        api.call("ganet_sga_backward_point", dev.ptr(dx), *[dev.ptr(g) for g in dg], dev.ptr(A), dev.ptr(G2), dev.ptr(gx2),
                 *[dev.ptr(g) for g in gw2], N, C, D, H, W, dev.stream)
        dev.sync()
        # the adjoint volumes themselves: the vertical directions' may be tiled in the private workspace
        hG, hG2 = dev.host(G), dev.host(G2)
        assert np.array_equal(hG, hG2)
        if api.query("ganet_sga_workspace_layout", N, C, D, H, W):
            G3 = dev.empty(x.shape)
            for d in range(2):
                api.call("ganet_sga_backward_scan", dev.ptr(dg[d]), dev.ptr(mask), dev.ptr(kp) + 2 * d * (N * C * H * W), dev.ptr(dgo),
                         dev.ptr(G3), N, C, D, H, W, d, dev.stream)
                dev.sync()
                assert np.array_equal(untile_ws(hG[d]), dev.host(G3)), f"tiled adjoint volume of direction {d}"
        assert np.array_equal(dev.host(gx2), dev.host(gx))
        for d in range(4):
            assert np.array_equal(dev.host(gw2[d]), dev.host(gw[d]))
    err = {"gx": float(np.abs(dev.host(gx) - want["gx"]).max())}
    for d in range(4):
        err[f"gw{d}"] = float(np.abs(dev.host(gw[d]) - want[f"gw{d}"]).max())
    assert max(err.values()) <= TOL, err
    if results is not None:                  # (`results`: dict that receives the gradients)
        results["gx"] = dev.host(gx)
        for d in range(4):
            results[f"gw{d}"] = dev.host(gw[d])
    return err


def run_sga_backward_only(api, dev, x, gs, go, go_offset=0):
    """forward + backward through the composite entries, gradients as host arrays (no oracle: the caller compares runs);
    go_offset: the incoming gradient starts that many floats into its buffer (contiguous, 4-byte aligned)"""
    N, C, D, H, W = x.shape
    dx, dg, A, out, mask, kp = run_sga_forward(api, dev, x, gs)
    if go_offset:
        dgo_buf = dev.to(np.concatenate([np.zeros(go_offset, np.float32), go.ravel()]))
        dgo = _PtrAt(dev.ptr(dgo_buf) + 4 * go_offset, dgo_buf)
        dev = _OffsetDev(dev)
    else:
        dgo = dev.to(go)
    gx = dev.empty(x.shape)
    gw = [dev.empty(gs[0].shape) for _ in range(4)]
    G = dev.empty((4,) + x.shape)
    api.call("ganet_sga_backward", dev.ptr(dx), *[dev.ptr(g) for g in dg], dev.ptr(A), dev.ptr(mask), dev.ptr(kp),
             dev.ptr(dgo), dev.ptr(G), dev.ptr(gx), *[dev.ptr(g) for g in gw], N, C, D, H, W, dev.stream)
    dev.sync()
    res = {"gx": dev.host(gx)}
    for d in range(4):
        res[f"gw{d}"] = dev.host(gw[d])
    return res


def check_sga_compat(api, dev, x, gs, go, want):
    """Reference buffer contract (sga_cuda_forward / sga_cuda_backward)."""
    N, C, D, H, W = x.shape
    dx = dev.to(x)
    dg = [dev.to(g) for g in gs]
    tmp, out, mask = dev.zeros(x.shape), dev.zeros(x.shape), dev.zeros(x.shape)
    api.call("ganet_sga_forward_compat", dev.ptr(dx), *[dev.ptr(g) for g in dg], dev.ptr(tmp), dev.ptr(out),
             dev.ptr(mask), N, C, D, H, W, dev.stream)
    dev.sync()
    assert np.array_equal(dev.host(out), want["out"])
    assert np.array_equal(dev.host(mask).astype(np.uint8), want["mask"])
    assert np.array_equal(dev.host(tmp), want["tmp"]), "temp_out must hold A_left"
    dgo = dev.to(go)
    gx = dev.zeros(x.shape)
    gw = [dev.zeros(gs[0].shape) for _ in range(4)]
    tgrad = dev.zeros(x.shape)
    idx = dev.zeros((N, C, H, W))
    api.call("ganet_sga_backward_compat", dev.ptr(dx), *[dev.ptr(g) for g in dg], dev.ptr(tmp), dev.ptr(mask),
             dev.ptr(idx), dev.ptr(dgo), dev.ptr(tgrad), dev.ptr(gx), *[dev.ptr(g) for g in gw],
             N, C, D, H, W, dev.stream)
    dev.sync()
    err = {"gx": float(np.abs(dev.host(gx) - want["gx"]).max())}
    for d in range(4):
        err[f"gw{d}"] = float(np.abs(dev.host(gw[d]) - want[f"gw{d}"]).max())
    assert max(err.values()) <= TOL, err
    return err


def lga_dims(x):
    if x.ndim == 5:
        return x.shape[0] * x.shape[1], x.shape[2], x.shape[3], x.shape[4]
    return x.shape


def _lga_errs(got, want):
    """max abs error per result against the oracle's (`want` None: no oracle at this size, the caller compares runs)"""
    if want is None:
        return {k: 0.0 for k in got}
    err = {k: float(np.abs(got[k] - want[k]).max()) for k in got}
    assert max(err.values()) <= TOL, err
    return err


def check_lga_chain(api, dev, x, f, gy, r, passes, want, out=None):
    """Chained LGA passes (Lga/Lga2/Lga3 and the 3d forms) through the one-pass ABI.  `out`: dict that receives the results."""
    B, D, H, W = lga_dims(x)
    df = dev.to(f)
    ins = [dev.to(x)]
    for _ in range(passes):
        y = dev.empty(x.shape)
        api.call("ganet_lga_forward", dev.ptr(ins[-1]), dev.ptr(df), dev.ptr(y), B, D, H, W, r, dev.stream)
        ins.append(y)
    dev.sync()
    got = {"y": dev.host(ins[-1])}
    g = dev.to(gy)
    gf = dev.empty(f.shape)
    for k, xin in enumerate(reversed(ins[:-1])):
        gx = dev.empty(x.shape)
        api.call("ganet_lga_backward", dev.ptr(xin), dev.ptr(df), dev.ptr(g), dev.ptr(gx), dev.ptr(gf),
                 B, D, H, W, r, 1 if k > 0 else 0, dev.stream)
        g = gx
    dev.sync()
    got["gx"], got["gf"] = dev.host(g), dev.host(gf)
    if out is not None:
        out.update(got)
    return _lga_errs(got, want)


def check_lga2_paired(api, dev, x, f, gy, r, passes, want, out=None):
    """The call sequence of Lga2Function with its private intermediate (and the intermediate's gradient) pair-interleaved
    (ganet_amd/functions/GANet.py: _LgaChain): x -> t1 (interleaved) -> y;  gf = gF(t1, gy);  g_t1 (interleaved) = gX(gy);
    gf += gF(x, g_t1);  gx = gX(g_t1).  Radius 2, two passes, even W."""
    assert r == 2 and passes == 2
    B, D, H, W = lga_dims(x)
    dx, df, dgy = dev.to(x), dev.to(f), dev.to(gy)
    pshape = (B, (D + 1) // 2, H, W, 2)
    t1p, y = dev.empty(pshape), dev.empty(x.shape)
    api.call("ganet_lga_apply_paired", dev.ptr(dx), dev.ptr(df), dev.ptr(t1p), B, D, H, W, 2, 0, 0, 1, dev.stream)
    api.call("ganet_lga_apply_paired", dev.ptr(t1p), dev.ptr(df), dev.ptr(y), B, D, H, W, 2, 0, 1, 0, dev.stream)
    gf, gt1p, gx = dev.empty(f.shape), dev.empty(pshape), dev.empty(x.shape)
    api.call("ganet_lga_filter_grad_paired", dev.ptr(t1p), dev.ptr(dgy), dev.ptr(gf), B, D, H, W, 2, 0, 1, 0, dev.stream)
    api.call("ganet_lga_apply_paired", dev.ptr(dgy), dev.ptr(df), dev.ptr(gt1p), B, D, H, W, 2, 1, 0, 1, dev.stream)
    api.call("ganet_lga_filter_grad_paired", dev.ptr(dx), dev.ptr(gt1p), dev.ptr(gf), B, D, H, W, 2, 1, 0, 1, dev.stream)
    api.call("ganet_lga_apply_paired", dev.ptr(gt1p), dev.ptr(df), dev.ptr(gx), B, D, H, W, 2, 1, 1, 0, dev.stream)
    dev.sync()
    got = {"y": dev.host(y), "gx": dev.host(gx), "gf": dev.host(gf)}
    if out is not None:
        out.update(got)
    err = _lga_errs(got, want)
    # The same chain with the filters' edge sums as a side channel (ganet_lga_apply_paired_edges, what Lga2Function runs when a
    # backward will follow): written by the first pass, read by both data-backward launches.  The sums are the same loads added
    # in the same order on either side, so every result is bit-identical; the buffer itself is checked against its definition.
    edge = dev.empty((B, 3, H, W))
    t1e, gt1e, gxe = dev.empty(pshape), dev.empty(pshape), dev.empty(x.shape)
    api.call("ganet_lga_apply_paired_edges", dev.ptr(dx), dev.ptr(df), dev.ptr(t1e), dev.ptr(edge), B, D, H, W, 2, 0, 0, 1, dev.stream)
    api.call("ganet_lga_apply_paired_edges", dev.ptr(dgy), dev.ptr(df), dev.ptr(gt1e), dev.ptr(edge), B, D, H, W, 2, 1, 0, 1, dev.stream)
    api.call("ganet_lga_apply_paired_edges", dev.ptr(gt1e), dev.ptr(df), dev.ptr(gxe), dev.ptr(edge), B, D, H, W, 2, 1, 1, 0, dev.stream)
    dev.sync()
    assert np.array_equal(dev.host(t1e), dev.host(t1p)) and np.array_equal(dev.host(gt1e), dev.host(gt1p))
    assert np.array_equal(dev.host(gxe), dev.host(gx))
    he = dev.host(edge)
    inside = np.zeros((25, H, W), bool)
    for a in range(-2, 3):
        for b in range(-2, 3):
            ii, jj = np.arange(H)[:, None] + a, np.arange(W)[None, :] + b
            inside[(a + 2) * 5 + (b + 2)] = (ii >= 0) & (ii < H) & (jj >= 0) & (jj < W)
    f3 = f.reshape(B, 3, 25, H, W)
    want_edge = np.stack([(f3 * ~inside).sum((1, 2)), (f3[:, 0] * inside).sum(1), (f3[:, 2] * inside).sum(1)], 1)
    assert np.abs(he - want_edge).max() <= 1e-5, float(np.abs(he - want_edge).max())
    return err


def to_paired(v):
    """[B, D, H, W] -> the pair-interleaved layout [B, ceil(D/2), H, W, 2] of ganet_lga_apply_paired (odd D: zero odd half)."""
    B, D, H, W = v.shape
    out = np.zeros((B, (D + 1) // 2, H, W, 2), np.float32)
    out[..., 0] = v[:, 0::2]
    out[:, :D // 2, :, :, 1] = v[:, 1::2]
    return out


def from_paired(p, D):
    B, _, H, W, _ = p.shape
    v = np.empty((B, D, H, W), np.float32)
    v[:, 0::2] = p[..., 0][:, :(D + 1) // 2]
    v[:, 1::2] = p[..., 1][:, :D // 2]
    return v

