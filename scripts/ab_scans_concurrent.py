"""Do the four forward (adjoint) scans of an SGA pass run faster side by side than one after the other?  They read the same x
(gradOut + mask): on four streams the second to fourth reader may find it in the L2 / Infinity Cache.  Both forms are captured
into hipGraphs and replayed.  python scripts/ab_scans_concurrent.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ganet_amd import _native

lib = _native.lib()
dev = torch.device("cuda:0")
inp = bench.make_inputs(dev)
x, gs, go, xl, f, gy = [t.detach() if torch.is_tensor(t) else [u.detach() for u in t] for t in inp]
N, C, D, H, W = x.shape
npix = N * C * H * W
A = torch.empty((4,) + tuple(x.shape), device=dev)
out = torch.empty_like(x); mask = torch.empty(x.shape, dtype=torch.uint8, device=dev)
kp = torch.empty((4, N, C, H, W), dtype=torch.int16, device=dev)
G = torch.empty_like(A)
lib.call("ganet_sga_forward", x.data_ptr(), *[g.data_ptr() for g in gs], A.data_ptr(), out.data_ptr(), mask.data_ptr(), kp.data_ptr(),
         N, C, D, H, W, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(4)]


def fwd(d, st):
    lib.call("ganet_sga_scan_forward", x.data_ptr(), gs[d].data_ptr(), A[d].data_ptr(), N, C, D, H, W, d, st.cuda_stream)


def bwd(d, st):
    lib.call("ganet_sga_backward_scan", gs[d].data_ptr(), mask.data_ptr(), kp.data_ptr() + 2 * d * npix, go.data_ptr(), G[d].data_ptr(),
             N, C, D, H, W, d, st.cuda_stream)


def capture(fn, concurrent, order=(0, 1, 2, 3)):
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for d in order:
            fn(d, cap)
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if concurrent:
            for d in order:
                streams[d].wait_stream(cur)
                fn(d, streams[d])
            for d in order:
                cur.wait_stream(streams[d])
        else:
            for d in order:
                fn(d, cur)
    return g


def timeit(g, n=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for name, fn in (("forward scans", fwd), ("adjoint scans", bwd)):
    gseq, gcon = capture(fn, False), capture(fn, True)
    gpair = None
    r = []
    for rep in range(3):
        r.append((timeit(gseq), timeit(gcon)))
    print(name, "sequential %.4f ms   four streams %.4f ms" % tuple(sorted(v)[1] for v in zip(*r)), flush=True)
    # two at a time: a column scan beside a row scan (different bottlenecks)
    def two(pairs):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cur = torch.cuda.current_stream()
            for a, b in pairs:
                streams[0].wait_stream(cur); streams[1].wait_stream(cur)
                fn(a, streams[0]); fn(b, streams[1])
                cur.wait_stream(streams[0]); cur.wait_stream(streams[1])
        return g
    g2 = two([(0, 2), (1, 3)])
    print(name, "two at a time (column beside row) %.4f ms" % sorted(timeit(g2) for _ in range(3))[1], flush=True)
    # round 5: the row scans are bound by what a SIMD can issue (time follows the waves on the fullest SIMD: 2,560 rows on 1,024
    # SIMDs = 3 rounds for 2.5), so the two row scans side by side (5,120 waves = 5 per SIMD, balanced) could save a round of six;
    # likewise the two column scans (832 workgroups = 3.25 per CU instead of 2 x 1.625 -> 2 x 2)
    g3 = two([(2, 3)])
    gseq_rows = capture(fn, False, order=(2, 3))
    print(name, "rows only: sequential %.4f ms   right beside left %.4f ms" % (sorted(timeit(gseq_rows) for _ in range(3))[1], sorted(timeit(g3) for _ in range(3))[1]), flush=True)
    g4 = two([(0, 1)])
    gseq_cols = capture(fn, False, order=(0, 1))
    print(name, "columns only: sequential %.4f ms   down beside up %.4f ms" % (sorted(timeit(gseq_cols) for _ in range(3))[1], sorted(timeit(g4) for _ in range(3))[1]), flush=True)
    g5 = two([(0, 1), (2, 3)])
    print(name, "all four: (down beside up) then (right beside left) %.4f ms" % sorted(timeit(g5) for _ in range(3))[1], flush=True)
