"""Same-box A/B of library builds on the stage timings: python scripts/ab_stage.py <key-prefix> libA.so libB.so ... (paths under ganet_amd/; 'env:NAME=V' entries set an option for the following libs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ganet_amd import _native
pre = sys.argv[1]
inp = bench.make_inputs(torch.device("cuda:0"))
for rep in range(2):
    for name in sys.argv[2:]:
        lib, _, opt = name.partition("@")
        _native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", lib))
        if opt:
            k, v = opt.split("=")
            _native._LIB.set_option(k, int(v))
        st = bench.stage_timings(inp, iters=10)
        print(name, {k: round(v, 4) for k, v in st.items() if k.startswith(pre)}, flush=True)
