"""Stage timings of the hot path (C ABI + HIP events): python scripts/bench_stages.py [prefix] -> one dict."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ganet_amd import _native
lib = _native.lib()
inp = bench.make_inputs(torch.device("cuda:0"))
st = bench.stage_timings(inp, iters=10)
pre = sys.argv[1] if len(sys.argv) > 1 else ""
print({k: round(v, 4) for k, v in st.items() if k.startswith(pre)})
