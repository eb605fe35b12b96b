"""What of an LGA launch does not scale with the depth?  The plane-pair kernels (API-layout forward = lga_apply_pp_wx, filter
gradient = lga_filter_grad_pp_x) at 240 x 624 over D = 25 .. 385, each timed back to back with a cache-flushing write in between
(so that a launch does not find its operands in L2 / the Infinity Cache): time(D) = a + b D, a = tap gather + ring fill + first /
last general steps + dispatch ramp.  python scripts/lga_depth_sweep.py [H W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ganet_amd import _native
lib = _native.lib()
H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (240, 624)
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
rows = []
for D in (25, 49, 97, 193, 289, 385):
    x = torch.randn((1, D, H, W), device="cuda"); f = torch.randn((1, 75, H, W), device="cuda")
    y = torch.empty_like(x); gf = torch.empty_like(f)
    gx = torch.empty_like(x)
    calls = {"apply": lambda: lib.call("ganet_lga_forward", x.data_ptr(), f.data_ptr(), y.data_ptr(), 1, D, H, W, 2, st),
             "filter_grad": lambda: lib.call("ganet_lga_filter_grad_paired", x.data_ptr(), y.data_ptr(), gf.data_ptr(), 1, D, H, W, 2, 0, 0, 0, st),
             "backward (filter_grad + apply_T)": lambda: lib.call("ganet_lga_backward", x.data_ptr(), f.data_ptr(), y.data_ptr(), gx.data_ptr(), gf.data_ptr(), 1, D, H, W, 2, 0, st)}
    res = {}
    for name, fn in calls.items():
        for _ in range(2): fn()
        ts = []
        for _ in range(7):
            flush.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[name] = sorted(ts)[len(ts) // 2]
    rows.append((D, res))
    print(D, {k: round(v, 1) for k, v in res.items()}, flush=True)
import numpy as np
Ds = np.array([r[0] for r in rows], float)
for name in rows[0][1]:
    t = np.array([r[1][name] for r in rows])
    b, a = np.polyfit(Ds, t, 1)
    print(f"{name:34s} time(D) = {a:6.1f} us + {b:.4f} us x D   (D = 193: {a + 193 * b:.1f} us, fixed share {a / (a + 193 * b):.0%})")
