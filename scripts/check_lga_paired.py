"""First GPU run of the pair-interleaved LGA2 intermediate (GANET_LGA_PAIRED=1; kernels lga_apply_pp_po / lga_apply_pp_pi /
lga_filter_grad_pp_xp, ABI v7) -- built and checked on the CPU emulator at the end of round 2, NOT yet run on a GPU.
  1. parity of Lga2Function forward + backward, option on vs off, at small shapes and the cfg2 / cfg5 model shapes;
  2. timing of Lga2Function forward and forward+backward at [1,193,240,624], option off / on.
python scripts/check_lga_paired.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd.functions.GANet import Lga2Function

dev = torch.device("cuda:0")


def run(shape, paired, iters=0):
    os.environ["GANET_LGA_PAIRED"] = "1" if paired else "0"
    B, D, H, W = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(shape, device=dev, requires_grad=True)
    f = F.normalize(torch.randn(B, 75, H, W, device=dev), p=1, dim=1).requires_grad_()
    gy = torch.randn(shape, device=dev)
    y = Lga2Function.apply(x, f, 2)
    y.backward(gy)
    torch.cuda.synchronize()
    out = (y.detach().clone(), x.grad.clone(), f.grad.clone())
    t = None
    if iters:
        def fwd():
            with torch.no_grad():
                Lga2Function.apply(x, f, 2)

        def both():
            x.grad = None; f.grad = None
            Lga2Function.apply(x, f, 2).backward(gy)
        t = {}
        for name, fn in (("fwd", fwd), ("fwd+bwd", both)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); e1.synchronize()
            t[name] = round(e0.elapsed_time(e1) / iters, 4)
    return out, t


ok = True
for shape, iters in [((1, 9, 3, 36), 0), ((2, 21, 5, 68), 0), ((1, 41, 7, 64), 0), ((1, 193, 24, 624), 0), ((1, 193, 240, 624), 30),
                     ((2, 193, 528, 960), 5)]:
    (y0, gx0, gf0), t0 = run(shape, False, iters)
    (y1, gx1, gf1), t1 = run(shape, True, iters)
    e = [float((a - b).abs().max()) for a, b in ((y0, y1), (gx0, gx1), (gf0, gf1))]
    good = e[0] <= 2e-6 and e[1] <= 2e-6 and e[2] <= 1e-4
    ok &= good
    print(shape, "max |diff| y / gx / gf:", e, "OK" if good else "MISMATCH", "\n   ms off:", t0, "\n   ms on: ", t1, flush=True)
print("LGA_PAIRED_OK" if ok else "LGA_PAIRED_MISMATCH")
sys.exit(0 if ok else 1)
