"""Fused caller-side chains vs the reference's op-by-op statements on the same GPU, cfg2 shapes.
Both sides use this library's SGA / LGA kernels; the difference is only the torch op chains around them."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd.modules.fused import DispAggTail, GuidedSGA, GuidedSGABnRelu
from ganet_amd.modules.GANet import SGA, LGA2, DisparityRegression

dev = torch.device("cuda:0")
torch.manual_seed(123)


def timed(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


res = {}
# --- SGABlock guidance + SGA (models/GANet_deep.py:263-269), fwd+bwd
x = torch.randn(1, 32, 65, 80, 208, device=dev, requires_grad=True)
g = torch.randn(1, 640, 80, 208, device=dev, requires_grad=True)
go = torch.randn_like(x)
sga = SGA()


def ref_sgablock():
    k1, k2, k3, k4 = torch.split(g, (160, 160, 160, 160), 1)
    ks = [F.normalize(k.view(1, 32, 5, 80, 208), p=1, dim=2) for k in (k1, k2, k3, k4)]
    out = sga(x, *ks)
    torch.autograd.grad(out, [x, g], go)


def fused_sgablock():
    out = GuidedSGA()(x, g)
    torch.autograd.grad(out, [x, g], go)


res["sgablock_ref_ms"] = timed(ref_sgablock)
res["sgablock_fused_ms"] = timed(fused_sgablock)

# --- inference: normalise + SGA + BatchNorm3d(eval) + ReLU (models/GANet_deep.py:263-271), forward only
bn = torch.nn.BatchNorm3d(32).to(dev).eval()
fused_inf = GuidedSGABnRelu(bn).eval()


def ref_infer():
    with torch.no_grad():
        k1, k2, k3, k4 = torch.split(g, (160, 160, 160, 160), 1)
        ks = [F.normalize(k.view(1, 32, 5, 80, 208), p=1, dim=2) for k in (k1, k2, k3, k4)]
        return torch.relu_(bn(sga(x, *ks)))


def fused_infer():
    with torch.no_grad():
        return fused_inf(x, g)


res["sgablock_infer_ref_ms"] = timed(ref_infer)
res["sgablock_infer_fused_ms"] = timed(fused_infer)

# --- DispAgg tail (models/GANet_deep.py:243-247), fwd+bwd
xl = torch.randn(1, 193, 240, 624, device=dev, requires_grad=True)
lg1 = torch.randn(1, 75, 240, 624, device=dev, requires_grad=True)
lg2 = torch.randn(1, 75, 240, 624, device=dev, requires_grad=True)
gd = torch.randn(1, 240, 624, device=dev)
lga2, softmin, disparity = LGA2(radius=2), torch.nn.Softmin(dim=1), DisparityRegression(192)


def ref_tail():
    t = lga2(xl, F.normalize(lg1, p=1, dim=1))
    t = softmin(t)
    t = lga2(t, F.normalize(lg2, p=1, dim=1))
    t = F.normalize(t, p=1, dim=1)
    out = disparity(t)
    torch.autograd.grad(out, [xl, lg1, lg2], gd)


tail = DispAggTail(192)


def fused_tail():
    out = tail(xl, lg1, lg2)
    torch.autograd.grad(out, [xl, lg1, lg2], gd)


res["dispagg_tail_ref_ms"] = timed(ref_tail)
res["dispagg_tail_fused_ms"] = timed(fused_tail)
print(json.dumps({k: round(v, 4) for k, v in res.items()}))
