"""LGA forward pass at an arbitrary shape: python scripts/bench_lga_shape.py D H W [iters] -> ms, ns per pixel-plane."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ganet_amd import _native
lib = _native.lib()
D, H, W = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
x = torch.randn((1, D, H, W), device="cuda"); f = torch.randn((1, 75, H, W), device="cuda"); y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
def run(): lib.call("ganet_lga_forward", x.data_ptr(), f.data_ptr(), y.data_ptr(), 1, D, H, W, 2, st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"D={D} H={H} W={W} waves64={H*W/64:.0f}: {ms:.4f} ms  {ms*1e6/(D*H*W):.4f} ns/px-plane")
