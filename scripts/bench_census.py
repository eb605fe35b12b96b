"""All guided-aggregation work of ONE GANet-deep training step at the cfg2/cfg4 per-GPU shapes (SURVEY.md Appendix B:
3 x SGABlock on [1,32,65,80,208], 4 x SGABlock on [1,48,33,40,104], GetCostVolume, DispAgg tail (2 x LGA2 + Softmin +
normalise + regression), 2 x Disp tail (Softmin + regression)), forward + backward, op-by-op as the reference writes
it vs the fused modules of ganet_amd.modules.fused.  Both sides use this library's SGA / LGA kernels."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd.modules.fused import DispAggTail, GuidedSGA, SoftminDisparityRegression
from ganet_amd.modules.GANet import SGA, LGA2, DisparityRegression, GetCostVolume

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, iters=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


xa = [torch.randn(1, 32, 65, 80, 208, device=dev, requires_grad=True) for _ in range(3)]
ga = [torch.randn(1, 640, 80, 208, device=dev, requires_grad=True) for _ in range(3)]
xb = [torch.randn(1, 48, 33, 40, 104, device=dev, requires_grad=True) for _ in range(4)]
gb = [torch.randn(1, 960, 40, 104, device=dev, requires_grad=True) for _ in range(4)]
fl, fr_ = (torch.randn(1, 32, 80, 208, device=dev, requires_grad=True) for _ in range(2))
vol = torch.randn(1, 193, 240, 624, device=dev, requires_grad=True)
lg1, lg2 = (torch.randn(1, 75, 240, 624, device=dev, requires_grad=True) for _ in range(2))
aux = [torch.randn(1, 193, 240, 624, device=dev, requires_grad=True) for _ in range(2)]
sga, lga2, softmin, disparity, cv = SGA(), LGA2(radius=2), torch.nn.Softmin(dim=1), DisparityRegression(192), GetCostVolume(64)
gsga, tail, sdr = GuidedSGA(), DispAggTail(192), SoftminDisparityRegression(192)
leaves = xa + ga + xb + gb + [fl, fr_, vol, lg1, lg2] + aux


def sgablock_ref(x, g):
    C = x.shape[1]
    ks = torch.split(g, (C * 5,) * 4, 1)
    ks = [F.normalize(k.view(x.shape[0], C, 5, x.shape[3], x.shape[4]), p=1, dim=2) for k in ks]
    return sga(x, *ks)


def step(fused):
    outs = []
    for x, g in zip(xa + xb, ga + gb):
        outs.append((gsga(x, g) if fused else sgablock_ref(x, g)).sum())
    outs.append(cv(fl, fr_).sum())
    if fused:
        outs.append(tail(vol, lg1, lg2).sum())
        outs += [sdr(a).sum() for a in aux]
    else:
        t = lga2(vol, F.normalize(lg1, p=1, dim=1)); t = softmin(t); t = lga2(t, F.normalize(lg2, p=1, dim=1))
        outs.append(disparity(F.normalize(t, p=1, dim=1)).sum())
        outs += [disparity(softmin(a)).sum() for a in aux]
    torch.autograd.grad(sum(outs), leaves)


res = {"ga_ops_per_step_ref_ms": timed(lambda: step(False)), "ga_ops_per_step_fused_ms": timed(lambda: step(True))}
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
