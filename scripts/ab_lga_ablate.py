"""Forward-pass time of lga_apply_pp (whole tiles, one depth segment) for several builds of the library
(scripts/build_variants.py tags with -DLGAP_ABLATE=<bits>): python scripts/ab_lga_ablate.py lib1.so lib2.so ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd import _native
B, D, H, W = 1, 193, 240, 624
torch.manual_seed(0)
x = torch.randn(B, D, H, W, device="cuda")
f = F.normalize(torch.randn(B, 75, H, W, device="cuda"), p=1, dim=1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for name in sys.argv[1:]:
        lib = _native.CApi(os.path.join(ROOT, "ganet_amd", name))
        for segs in (1, 0):
            lib.set_option("GANET_LGA_MIX", 0); lib.set_option("GANET_LGA_SEGS", segs)
            fn = lambda: lib.call("ganet_lga_forward", x.data_ptr(), f.data_ptr(), y.data_ptr(), B, D, H, W, 2, st)
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); e1.synchronize()
            print(f"rep{rep} {name:36s} segs={segs}  fwd {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
