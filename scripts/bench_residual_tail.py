"""SGABlock's tail behind conv_refine's convolution (models/GANet_deep.py:270-277: BatchNorm3d, `x += rem`, relu) at the block
shapes of cfg2/cfg4, cfg3 and cfg5: the reference's statements on stock PyTorch kernels vs ganet_amd.modules.fused.ResidualBnRelu.
eval = forward under no_grad (predict.py), train = forward + backward with batch statistics.  GB/s = the fused form's
algorithmic bytes (eval 3 V; train: bn 2 V + 3 V + 3 V backward of the tail itself) over its time."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ganet_amd.modules.fused import ResidualBnRelu

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, iters=20):
    """device time of fn: captured once into a hipGraph and replayed (these tails are 20 - 500 us of kernels: an eager loop
    measures the host -- a Python autograd.Function + ctypes call against three ATen dispatches -- not the GPU, which is what
    bounds a model pass whose host runs far ahead of 60 - 500 ms of convolutions)"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


def stock(bn, t, rem):
    x = bn(t)
    x += rem
    return torch.relu_(x)


out = {}
for name, shape in (("cfg2_A", (1, 32, 65, 80, 208)), ("cfg2_B", (1, 48, 33, 40, 104)), ("cfg3_A", (1, 32, 65, 128, 416)),
                    ("cfg5_A", (2, 32, 65, 176, 320))):
    C = shape[1]
    V = 4 * torch.Size(shape).numel()
    bn = torch.nn.BatchNorm3d(C).to(dev)
    m = ResidualBnRelu(bn)
    t, rem, gy = (torch.randn(shape, device=dev) for _ in range(3))
    bn.eval()
    for p in bn.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        e_ref = timed(lambda: stock(bn, t, rem))
        e_fus = timed(lambda: m(t.copy_(gy), rem)) - timed(lambda: t.copy_(gy))    # in place over t: refill it, and take the refill out
    bn.train()
    for p in bn.parameters():
        p.requires_grad_(True)
    tl, rl = t.clone().requires_grad_(), rem.clone().requires_grad_()
    tr_ref = timed(lambda: torch.autograd.grad(stock(bn, tl, rl), [tl, rl, bn.weight, bn.bias], gy))
    tr_fus = timed(lambda: torch.autograd.grad(m(tl, rl), [tl, rl, bn.weight, bn.bias], gy))
    out[name] = {"shape": list(shape), "V_MB": round(V / 1e6, 1),
                 "eval_stock_ms": round(e_ref, 4), "eval_fused_ms": round(e_fus, 4), "eval_fused_GBs": round(3 * V / e_fus / 1e6, 0),
                 "train_stock_ms": round(tr_ref, 4), "train_fused_ms": round(tr_fus, 4)}
print(json.dumps(out, indent=1))
