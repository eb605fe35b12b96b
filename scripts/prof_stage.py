"""Runs ONE stage of the hot path a few times (for rocprofv3 --pmc / --kernel-trace runs).
python scripts/prof_stage.py <stage> [iters]   stage in: sga_fwd_v sga_fwd_h sga_bwd_v sga_bwd_h sga_fwd sga_bwd lga_fwd lga_bwd lga_api all step
(step = exactly the 16 launches of one benchmark step, the private workspaces as the Functions keep them: what the PMC passes
behind profiles/traffic_pmc.json run)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ganet_amd import _native  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if os.environ.get("GANET_PROF_LIB"):      # (development: a tagged build of the library, scripts/build_variants.py)
    _native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", os.environ["GANET_PROF_LIB"]))
lib = _native.lib()
inp = bench.make_inputs(torch.device("cuda:0"))
x, gs, go, xl, f, gy = [t.detach() if torch.is_tensor(t) else [u.detach() for u in t] for t in inp]
N, C, D, H, W = x.shape
st = torch.cuda.current_stream().cuda_stream
A = torch.empty((4,) + tuple(x.shape), device=x.device)
out = torch.empty_like(x)
mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
kp = torch.empty((4, N, C, H, W), dtype=torch.int16, device=x.device)
G = torch.empty_like(A)
gx = torch.empty_like(x)
gw = [torch.empty_like(g) for g in gs]
B, DL, HL, WL = xl.shape
t1, gxl = torch.empty_like(xl), torch.empty_like(xl)
gf = torch.empty_like(f)
tp = torch.empty(B * ((DL + 1) // 2) * HL * WL * 2, device=xl.device)
gtp = torch.empty_like(tp)
edge = torch.empty((B, 3, HL, WL), device=xl.device)
lib.call("ganet_sga_forward", x.data_ptr(), *[g.data_ptr() for g in gs], A.data_ptr(), out.data_ptr(), mask.data_ptr(),
         kp.data_ptr(), N, C, D, H, W, st)
torch.cuda.synchronize()


def fwd(d):
    lib.call("ganet_sga_scan_forward", x.data_ptr(), gs[d].data_ptr(), A[d].data_ptr(), N, C, D, H, W, d, st)


def bwd(d):
    lib.call("ganet_sga_backward_scan", gs[d].data_ptr(), mask.data_ptr(), kp.data_ptr() + 2 * d * N * C * H * W,
             go.data_ptr(), G[d].data_ptr(), N, C, D, H, W, d, st)


def fullfwd():
    lib.call("ganet_sga_forward", x.data_ptr(), *[g.data_ptr() for g in gs], A.data_ptr(), out.data_ptr(),
             mask.data_ptr(), kp.data_ptr(), N, C, D, H, W, st)


def fullbwd():
    lib.call("ganet_sga_backward", x.data_ptr(), *[g.data_ptr() for g in gs], A.data_ptr(), mask.data_ptr(),
             kp.data_ptr(), go.data_ptr(), G.data_ptr(), gx.data_ptr(), *[g.data_ptr() for g in gw],
             N, C, D, H, W, st)


for _ in range(iters):
    if stage in ("sga_fwd_v", "all"):
        fwd(0)
    if stage in ("sga_fwd_h", "all"):
        fwd(2)
    if stage in ("sga_bwd_v", "all"):
        bwd(0)
    if stage in ("sga_bwd_h", "all"):
        bwd(2)
    if stage in ("sga_fwd", "all", "step"):
        fullfwd()
    if stage in ("sga_bwd", "all", "step"):
        fullbwd()
    # LGA2 as Lga2Function runs it (pair-interleaved private intermediate and intermediate gradient)
    if stage in ("lga_fwd", "all", "step"):
        lib.call("ganet_lga_apply_paired_edges", xl.data_ptr(), f.data_ptr(), tp.data_ptr(), edge.data_ptr(), B, DL, HL, WL, 2, 0, 0, 1, st)
        lib.call("ganet_lga_apply_paired", tp.data_ptr(), f.data_ptr(), t1.data_ptr(), B, DL, HL, WL, 2, 0, 1, 0, st)
    if stage in ("lga_bwd", "all", "step"):
        lib.call("ganet_lga_filter_grad_paired", tp.data_ptr(), gy.data_ptr(), gf.data_ptr(), B, DL, HL, WL, 2, 0, 1, 0, st)
        lib.call("ganet_lga_apply_paired_edges", gy.data_ptr(), f.data_ptr(), gtp.data_ptr(), edge.data_ptr(), B, DL, HL, WL, 2, 1, 0, 1, st)
        lib.call("ganet_lga_filter_grad_paired", xl.data_ptr(), gtp.data_ptr(), gf.data_ptr(), B, DL, HL, WL, 2, 1, 0, 1, st)
        lib.call("ganet_lga_apply_paired_edges", gtp.data_ptr(), f.data_ptr(), gxl.data_ptr(), edge.data_ptr(), B, DL, HL, WL, 2, 1, 1, 0, st)
    if stage in ("lga_api", "all"):                      # one pass on the API layout (LgaFunction; mixed item list)
        lib.call("ganet_lga_forward", xl.data_ptr(), f.data_ptr(), t1.data_ptr(), B, DL, HL, WL, 2, st)
        lib.call("ganet_lga_backward", xl.data_ptr(), f.data_ptr(), gy.data_ptr(), gxl.data_ptr(), gf.data_ptr(),
                 B, DL, HL, WL, 2, 0, st)
torch.cuda.synchronize()
