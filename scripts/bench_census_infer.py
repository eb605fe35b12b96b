"""All guided-aggregation work of one GANet-deep INFERENCE pass at the cfg3 shapes (KITTI 1248x384: 1/3-resolution
volumes [1,32,65,128,416], 1/6-resolution [1,48,33,64,208], full-resolution [1,193,384,1248]), forward only under
no_grad: op by op as the reference writes it vs ganet_amd.modules.fused (eval mode).  Round 6: the seven SGABlocks' tails
behind conv_refine's convolution (BatchNorm3d + `x += rem` + relu, models/GANet_deep.py:270-277) are part of the census, as
stock statements vs ResidualBnRelu; `--no-tail` gives the census of rounds 1-5."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ganet_amd.modules.fused import DispAggTail, GuidedSGABnRelu, ResidualBnRelu
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


xa = [torch.randn(1, 32, 65, 128, 416, device=dev) for _ in range(3)]
ga = [torch.randn(1, 640, 128, 416, device=dev) for _ in range(3)]
xb = [torch.randn(1, 48, 33, 64, 208, device=dev) for _ in range(4)]
gb = [torch.randn(1, 960, 64, 208, device=dev) for _ in range(4)]
fl, fr_ = (torch.randn(1, 32, 128, 416, device=dev) for _ in range(2))
vol = torch.randn(1, 193, 384, 1248, device=dev)
lg1, lg2 = (torch.randn(1, 75, 384, 1248, device=dev) for _ in range(2))
sga, lga2, softmin, disparity, cv = SGA(), LGA2(radius=2), torch.nn.Softmin(dim=1), DisparityRegression(192), GetCostVolume(64)
bna = [torch.nn.BatchNorm3d(32).to(dev).eval() for _ in range(3)]
bnb = [torch.nn.BatchNorm3d(48).to(dev).eval() for _ in range(4)]
fa = [GuidedSGABnRelu(b).eval() for b in bna]
fb = [GuidedSGABnRelu(b).eval() for b in bnb]
tail = DispAggTail(192)
TAIL = "--no-tail" not in sys.argv
# conv_refine's output (a temporary the tail may overwrite) and its BatchNorm3d, per block
ta = [torch.randn_like(x) for x in xa] if TAIL else []
tb = [torch.randn_like(x) for x in xb] if TAIL else []
bn2 = [torch.nn.BatchNorm3d(c).to(dev).eval() for c in [32] * 3 + [48] * 4]
ftail = [ResidualBnRelu(b) for b in bn2]


def sgablock_ref(x, g, bn):
    C = x.shape[1]
    ks = torch.split(g, (C * 5,) * 4, 1)
    ks = [F.normalize(k.view(x.shape[0], C, 5, x.shape[3], x.shape[4]), p=1, dim=2) for k in ks]
    return torch.relu_(bn(sga(x, *ks)))


def step(fused):
    with torch.no_grad():
        for x, g, bn, m in zip(xa + xb, ga + gb, bna + bnb, fa + fb):
            (m(x, g) if fused else sgablock_ref(x, g, bn))
        for t, x, bn, m in zip(ta + tb, xa + xb, bn2, ftail):
            if fused:
                m(t, x)
            else:
                y = bn(t); y += x; torch.relu_(y)
        cv(fl, fr_)
        if fused:
            tail(vol, lg1, lg2)
        else:
            t = lga2(vol, F.normalize(lg1, p=1, dim=1)); t = softmin(t); t = lga2(t, F.normalize(lg2, p=1, dim=1))
            disparity(F.normalize(t, p=1, dim=1))


res = {"ga_ops_per_inference_ref_ms": timed(lambda: step(False)), "ga_ops_per_inference_fused_ms": timed(lambda: step(True))}
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
