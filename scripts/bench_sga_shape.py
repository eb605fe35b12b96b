"""Per-kernel stage timings of SGA at an arbitrary shape: python scripts/bench_sga_shape.py N C D H W"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd import _native
N, C, D, H, W = (int(v) for v in sys.argv[1:6])
if os.environ.get('GANET_VARIANT_LIB'):
    _native._LIB = _native.CApi(os.path.join(ROOT, 'ganet_amd', os.environ['GANET_VARIANT_LIB']))      # A/B builds of scripts/build_variants.py
lib = _native.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(N, C, D, H, W, device=dev)
gs = [F.normalize(torch.randn(N, C, 5, H, W, device=dev), p=1, dim=2) for _ in range(4)]
go = torch.randn_like(x)
st = torch.cuda.current_stream().cuda_stream
A = torch.empty((4,) + tuple(x.shape), device=dev); out = torch.empty_like(x)
mask = torch.empty(x.shape, dtype=torch.uint8, device=dev); kp = torch.empty((4, N, C, H, W), dtype=torch.int16, device=dev)
G = torch.empty_like(A); gx = torch.empty_like(x); gw = [torch.empty_like(g) for g in gs]


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


res = {}
names = ["down", "up", "right", "left"]
npix = N * C * H * W
for d in range(4):
    res[f"fwd_{names[d]}"] = timed(lambda d=d: lib.call("ganet_sga_scan_forward", x.data_ptr(), gs[d].data_ptr(), A[d].data_ptr(), N, C, D, H, W, d, st))
res["forward_call"] = timed(lambda: lib.call("ganet_sga_forward", x.data_ptr(), *[g.data_ptr() for g in gs], A.data_ptr(), out.data_ptr(), mask.data_ptr(), kp.data_ptr(), N, C, D, H, W, st))
for d in range(4):
    res[f"bwd_{names[d]}"] = timed(lambda d=d: lib.call("ganet_sga_backward_scan", gs[d].data_ptr(), mask.data_ptr(), kp.data_ptr() + 2 * d * npix, go.data_ptr(), G[d].data_ptr(), N, C, D, H, W, d, st))
res["backward_call"] = timed(lambda: lib.call("ganet_sga_backward", x.data_ptr(), *[g.data_ptr() for g in gs], A.data_ptr(), mask.data_ptr(), kp.data_ptr(), go.data_ptr(), G.data_ptr(), gx.data_ptr(), *[g.data_ptr() for g in gw], N, C, D, H, W, st))
V = x.numel() * 4
print({k: round(v, 4) for k, v in res.items()}, "V_MB", round(V / 1e6, 1), "fwd+bwd 32.25V at", round(32.25 * V / (res["forward_call"] + res["backward_call"]) / 1e9, 2), "TB/s")
