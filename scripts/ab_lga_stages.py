import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from ganet_amd import _native
for rep in range(2):
    for name in sys.argv[1:]:
        _native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", name), strict=False)
        inp = bench.make_inputs(torch.device("cuda:0"))
        st = bench.stage_timings(inp, iters=10, only="lga")
        print(name, {k: round(v, 4) for k, v in st.items() if k.startswith("lga")}, flush=True)
