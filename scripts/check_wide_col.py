"""First GPU run of the wide column-block kernels (GANET_SGA_WIDE_COL, sga_col_fwd_wide / sga_col_bwdg_wide: 1,024-thread
blocks, > 64 KB of dynamic LDS at D = 192) -- built and checked on the CPU emulator at the end of round 2, NOT yet run on a GPU.
  1. parity: vertical scans with the option on vs off, forward bit-exact, adjoint scans within 1e-5, at small shapes and
     at SURVEY 8d's stress shape [1,1,192,240,624];
  2. timing of the vertical scans on the stress shape, option off / on (scripts/bench_sga_shape.py does the full table).
python scripts/check_wide_col.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd import _native

lib = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def run(shape, wide):
    N, C, D, H, W = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(shape, device=dev)
    gs = [F.normalize(torch.randn(N, C, 5, H, W, device=dev), p=1, dim=2) for _ in range(2)]
    go = torch.randn_like(x)
    mask = torch.randint(0, 4, shape, dtype=torch.uint8, device=dev)
    kp = torch.randint(0, D, (2, N, C, H, W), dtype=torch.int16, device=dev)
    A = torch.full((2,) + tuple(shape), float("nan"), device=dev)
    G = torch.full((2,) + tuple(shape), float("nan"), device=dev)
    lib.set_option("GANET_SGA_WIDE_COL", wide)
    t = {}
    try:
        for d in range(2):
            def fwd(d=d):
                lib.call("ganet_sga_scan_forward", x.data_ptr(), gs[d].data_ptr(), A[d].data_ptr(), N, C, D, H, W, d, st)

            def bwd(d=d):
                lib.call("ganet_sga_backward_scan", gs[d].data_ptr(), mask.data_ptr(), kp[d].data_ptr(), go.data_ptr(), G[d].data_ptr(),
                         N, C, D, H, W, d, st)
            for name, fn in ((f"fwd{d}", fwd), (f"bwd{d}", bwd)):
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record(); e1.synchronize()
                t[name] = round(e0.elapsed_time(e1) / 10, 4)
    finally:
        lib.set_option("GANET_SGA_WIDE_COL", 0)
    return A, G, t


ok = True
for shape in [(1, 2, 65, 7, 20), (2, 1, 7, 9, 8), (1, 1, 150, 3, 12), (1, 1, 192, 5, 16), (1, 1, 192, 240, 624)]:
    A0, G0, t0 = run(shape, 0)
    A1, G1, t1 = run(shape, 1)
    same = bool(torch.equal(A0, A1))
    eg = float((G0 - G1).abs().max())
    ok &= same and eg <= 1e-5
    print(shape, "forward bit-exact:", same, " adjoint max diff:", eg, "\n   ms off:", t0, "\n   ms on: ", t1, flush=True)
print("WIDE_COL_OK" if ok else "WIDE_COL_MISMATCH")
sys.exit(0 if ok else 1)
