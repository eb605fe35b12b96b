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

# --- the fused caller-side kernels
res = {}
N, C, H, W = 1, 32, 80, 208
g = torch.randn(N, 20 * C, H, W, device=dev); ys = torch.empty(4, N, C, 5, H, W, device=dev); gg = torch.empty_like(g)
t = timed(lambda: lib.call("ganet_l1_normalize_forward", g.data_ptr(), *[ys[i].data_ptr() for i in range(4)], N, 4, C, 5, H, W, st))
res["guidance_normalize_fwd"] = (t, 2 * g.numel() * 4 / t / 1e6)
t = timed(lambda: lib.call("ganet_l1_normalize_backward", g.data_ptr(), *[ys[i].data_ptr() for i in range(4)], gg.data_ptr(), N, 4, C, 5, H, W, st))
res["guidance_normalize_bwd"] = (t, 3 * g.numel() * 4 / t / 1e6)
N, K, H, W = 1, 75, 240, 624
f = torch.randn(N, K, H, W, device=dev); fy = torch.empty_like(f); fg = torch.empty_like(f)
t = timed(lambda: lib.call("ganet_l1_normalize_forward", f.data_ptr(), fy.data_ptr(), None, None, None, N, 1, 1, K, H, W, st))
res["filter_normalize_fwd"] = (t, 2 * f.numel() * 4 / t / 1e6)
t = timed(lambda: lib.call("ganet_l1_normalize_backward", f.data_ptr(), fy.data_ptr(), None, None, None, fg.data_ptr(), N, 1, 1, K, H, W, st))
res["filter_normalize_bwd"] = (t, 3 * f.numel() * 4 / t / 1e6)
N, Dn, H, W = 1, 193, 240, 624
sn = torch.empty(N, H, W, device=dev)
t = timed(lambda: lib.call("ganet_norm_disparity_regression_forward", p.data_ptr(), out.data_ptr(), sn.data_ptr(), N, Dn, H, W, st))
res["norm_regression_fwd"] = (t, p.numel() * 4 / t / 1e6)
t = timed(lambda: lib.call("ganet_norm_disparity_regression_backward", p.data_ptr(), out.data_ptr(), sn.data_ptr(), go.data_ptr(), gp.data_ptr(), N, Dn, H, W, st))
res["norm_regression_bwd"] = (t, 2 * p.numel() * 4 / t / 1e6)
sy = torch.empty_like(p)
t = timed(lambda: lib.call("ganet_softmin_forward", p.data_ptr(), sy.data_ptr(), N, Dn, H, W, st))
res["softmin_fwd"] = (t, 2 * p.numel() * 4 / t / 1e6)
t = timed(lambda: lib.call("ganet_softmin_backward", sy.data_ptr(), p.data_ptr(), gp.data_ptr(), N, Dn, H, W, st))
res["softmin_bwd"] = (t, 3 * p.numel() * 4 / t / 1e6)
print(json.dumps({k: {"ms": round(v[0], 4), "GBps_compulsory": round(v[1], 1)} for k, v in res.items()}))
res = {}
mxb, ssb = torch.empty(N, H, W, device=dev), torch.empty(N, H, W, device=dev)
t = timed(lambda: lib.call("ganet_softmin_regression_forward", p.data_ptr(), out.data_ptr(), mxb.data_ptr(), ssb.data_ptr(), N, Dn, H, W, st))
res["softmin_regression_fwd"] = (t, p.numel() * 4 / t / 1e6)
t = timed(lambda: lib.call("ganet_softmin_regression_backward", p.data_ptr(), out.data_ptr(), mxb.data_ptr(), ssb.data_ptr(), go.data_ptr(), gp.data_ptr(), N, Dn, H, W, st))
res["softmin_regression_bwd"] = (t, 2 * p.numel() * 4 / t / 1e6)
import torch.nn.functional as F
xr = p.clone().requires_grad_()
disp = torch.arange(Dn, device=dev, dtype=torch.float32).view(1, Dn, 1, 1)
def ref():
    o = torch.sum(F.softmin(xr, dim=1) * disp, 1)
    torch.autograd.grad(o, [xr], go)
res["torch_softmin_regression_fwd_bwd"] = (timed(ref), 0.0)
print(json.dumps({k: {"ms": round(v[0], 4), "GBps_compulsory": round(v[1], 1)} for k, v in res.items()}))
