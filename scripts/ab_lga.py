"""Same-box A/B of the LGA kernel families on the cfg2 shape [1,193,240,624] (or argv shape B D H W): forward pass, data-backward
and filter-gradient times per GANET_LGA_WAVE / GANET_LGA_SEGS / GANET_LGA_SPLIT setting, plus the max difference of each
variant's results from the first one.   python scripts/ab_lga.py [B D H W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd import _native

args = [a for a in sys.argv[1:] if not a.startswith("--lib=")]
libname = [a[6:] for a in sys.argv[1:] if a.startswith("--lib=")]
if libname:                                     # a variant build (scripts/build_variants.py): ganet_amd/<name>
    _native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", libname[0]))
lib = _native.lib()
print("library:", lib.path)
shape = tuple(int(v) for v in args[0:4]) if len(args) >= 4 else (1, 193, 240, 624)
B, D, H, W = shape
torch.manual_seed(0)
x = torch.randn(shape, device="cuda")
f = F.normalize(torch.randn(B, 75, H, W, device="cuda"), p=1, dim=1)
gy = torch.randn(shape, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


variants = [dict(GANET_LGA_WAVE=2), dict(GANET_LGA_WAVE=3), dict(GANET_LGA_WAVE=3, GANET_LGA_BWD_STREAMS=1), dict(GANET_LGA_WAVE=3, GANET_LGA_FG_WPS=2),
            dict(GANET_LGA_WAVE=3, GANET_LGA_FG_WPS=2, GANET_LGA_BWD_STREAMS=1),
            dict(GANET_LGA_WAVE=3, GANET_LGA_VMCNT_SAFE=1),
            dict(GANET_LGA_WAVE=3, GANET_LGA_SEGS=1), dict(GANET_LGA_WAVE=3, GANET_LGA_SEGS=2), dict(GANET_LGA_WAVE=3, GANET_LGA_SEGS=3)]
base = None
for rep in range(2):
    for v in variants:
        for k in ("GANET_LGA_WAVE", "GANET_LGA_SEGS", "GANET_LGA_SPLIT", "GANET_LGA_VMCNT_SAFE", "GANET_LGA_FG_WPS", "GANET_LGA_BWD_STREAMS"):
            lib.set_option(k, {"GANET_LGA_WAVE": 2, "GANET_LGA_SEGS": 0, "GANET_LGA_SPLIT": 1, "GANET_LGA_VMCNT_SAFE": 0, "GANET_LGA_FG_WPS": 3,
                               "GANET_LGA_BWD_STREAMS": 0}[k])
        for k, val in v.items():
            lib.set_option(k, val)
        y, gx, gf = torch.empty_like(x), torch.empty_like(x), torch.empty_like(f)
        t_f = timed(lambda: lib.call("ganet_lga_forward", x.data_ptr(), f.data_ptr(), y.data_ptr(), B, D, H, W, 2, st))
        t_b = timed(lambda: lib.call("ganet_lga_backward", x.data_ptr(), f.data_ptr(), gy.data_ptr(), gx.data_ptr(), gf.data_ptr(), B, D, H, W, 2, 0, st))
        t_acc = timed(lambda: lib.call("ganet_lga_backward", x.data_ptr(), f.data_ptr(), gy.data_ptr(), gx.data_ptr(), gf.data_ptr(), B, D, H, W, 2, 1, st))
        lib.call("ganet_lga_backward", x.data_ptr(), f.data_ptr(), gy.data_ptr(), gx.data_ptr(), gf.data_ptr(), B, D, H, W, 2, 0, st)
        torch.cuda.synchronize()
        if base is None:
            base = (y.clone(), gx.clone(), gf.clone())
        diff = [float((a - b).abs().max()) for a, b in zip((y, gx, gf), base)]
        print(f"rep{rep} {str(v):70s} fwd {t_f:.4f} ms  bwd(gF+gX) {t_b:.4f} ms (accumulate {t_acc:.4f})  maxdiff y/gx/gf {diff[0]:.2e} {diff[1]:.2e} {diff[2]:.2e}", flush=True)
lib.set_option("GANET_LGA_WAVE", 2); lib.set_option("GANET_LGA_SEGS", 0)
