"""Does a faster filter-gradient kernel (stage timings: -11 %) shorten the step?  Per library: the whole step as a hipGraph, the
16 C-ABI launches of the step eagerly without events in between, and with them.  python scripts/diag_fg.py libA.so libB.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from ganet_amd import _native
dev = torch.device("cuda:0")
inp = bench.make_inputs(dev)
def ev_time(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n
for rep in range(2):
    for name in sys.argv[1:]:
        _native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", name), strict=False)
        st = bench.stage_timings(inp, iters=10)
        lga = {k: round(v, 4) for k, v in st.items() if k.startswith("lga_bwd_f")}
        eager = ev_time(lambda: bench.one_step(inp), 20)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            bench.one_step(inp)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            keep = bench.one_step(inp)
        graph = ev_time(g.replay, 50)
        print(f"{name:28s} sum of kernels (events between) {st['step_sum_of_kernels']:.4f}  eager step {eager:.4f}  graph {graph:.4f}  {lga}", flush=True)
        del keep, g
