"""Times SGA backward for several builds of the library (variants named on the command line)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ganet_amd import _native
for name in sys.argv[1:]:
    _native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", name))
    inp = bench.make_inputs(torch.device("cuda:0"))
    st = bench.stage_timings(inp, iters=10)
    print(name, {k: round(v, 4) for k, v in st.items() if k.startswith("sga_")}, flush=True)
