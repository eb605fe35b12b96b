"""GetCostVolume / DisparityRegression kernels at the cfg2 shapes through the C ABI: ms and GB/s."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ganet_amd import _native

dev = torch.device("cuda:0")
lib = _native.lib()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


N, C, Dn, H, W = 1, 32, 65, 80, 208
x = torch.randn(N, C, H, W, device=dev); y = torch.randn(N, C, H, W, device=dev)
cost = torch.empty(N, 2 * C, Dn, H, W, device=dev); gc = torch.randn_like(cost)
gx, gy = torch.empty_like(x), torch.empty_like(y)
res = {}
t = timed(lambda: lib.call("ganet_cost_volume_forward", x.data_ptr(), y.data_ptr(), cost.data_ptr(), N, C, Dn, H, W, st))
res["cost_volume_fwd"] = (t, cost.numel() * 4 / t / 1e6)
t = timed(lambda: lib.call("ganet_cost_volume_backward", gc.data_ptr(), gx.data_ptr(), gy.data_ptr(), N, C, Dn, H, W, st))
res["cost_volume_bwd"] = (t, cost.numel() * 4 / t / 1e6)
N, Dn, H, W = 1, 193, 240, 624
p = torch.rand(N, Dn, H, W, device=dev); out = torch.empty(N, H, W, device=dev); go = torch.randn(N, H, W, device=dev); gp = torch.empty_like(p)
t = timed(lambda: lib.call("ganet_disparity_regression_forward", p.data_ptr(), out.data_ptr(), N, Dn, H, W, st))
res["disp_regression_fwd"] = (t, p.numel() * 4 / t / 1e6)
t = timed(lambda: lib.call("ganet_disparity_regression_backward", go.data_ptr(), gp.data_ptr(), N, Dn, H, W, st))
res["disp_regression_bwd"] = (t, p.numel() * 4 / t / 1e6)
print(json.dumps({k: {"ms": round(v[0], 4), "GBps": round(v[1], 1)} for k, v in res.items()}))
