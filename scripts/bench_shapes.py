"""SGA fwd+bwd and LGA2 fwd+bwd through the autograd Functions at every shape the models feed the ops
(cfg2/cfg4 per-GPU, cfg3 KITTI 1248x384, cfg5 SceneFlow 2x960x528), ms + algorithmic GB/s."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd.functions.GANet import Lga2Function, SgaFunction

dev = torch.device("cuda:0")
SGA = {"cfg2 sga1-3 [1,32,65,80,208]": (1, 32, 65, 80, 208), "cfg2 sga11-14 [1,48,33,40,104]": (1, 48, 33, 40, 104),
       "cfg3 sga1-3 [1,32,65,128,416]": (1, 32, 65, 128, 416), "cfg3 sga11-14 [1,48,33,64,208]": (1, 48, 33, 64, 208),
       "cfg5 sga1-3 [2,32,65,176,320]": (2, 32, 65, 176, 320), "cfg5 sga11-14 [2,48,33,88,160]": (2, 48, 33, 88, 160)}
LGA = {"cfg2 lga [1,193,240,624]": (1, 193, 240, 624), "cfg3 lga [1,193,384,1248]": (1, 193, 384, 1248),
       "cfg5 lga [2,193,528,960]": (2, 193, 528, 960)}


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


rows = []
for name, s in SGA.items():
    torch.manual_seed(1)
    N, C, D, H, W = s
    x = torch.randn(s, device=dev, requires_grad=True)
    gs = [F.normalize(torch.randn(N, C, 5, H, W, device=dev), p=1, dim=2).requires_grad_() for _ in range(4)]
    go = torch.randn(s, device=dev)
    def step():
        out = SgaFunction.apply(x, *gs)
        torch.autograd.grad(out, [x] + gs, go)
    ms = timed(step)
    V, G = 4 * x.numel(), 4 * gs[0].numel()
    alg = 5 * V + 12 * G
    rows.append({"op": "SGA fwd+bwd", "shape": name, "ms": round(ms, 4), "alg_GB": round(alg / 1e9, 4), "alg_GBps": round(alg / ms / 1e6, 1)})
    del x, gs, go
    torch.cuda.empty_cache()
for name, s in LGA.items():
    torch.manual_seed(2)
    N, D, H, W = s
    x = torch.randn(s, device=dev, requires_grad=True)
    f = F.normalize(torch.randn(N, 75, H, W, device=dev), p=1, dim=1).requires_grad_()
    gy = torch.randn(s, device=dev)
    def step():
        y = Lga2Function.apply(x, f, 2)
        torch.autograd.grad(y, [x, f], gy)
    ms = timed(step)
    V, Fb = 4 * x.numel(), 4 * f.numel()
    alg = 7 * V + 3 * Fb
    rows.append({"op": "LGA2 fwd+bwd", "shape": name, "ms": round(ms, 4), "alg_GB": round(alg / 1e9, 4), "alg_GBps": round(alg / ms / 1e6, 1)})
    del x, f, gy
    torch.cuda.empty_cache()
for r in rows:
    print(json.dumps(r))
