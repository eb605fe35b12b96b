"""GPU tuning sweep for the SGA scan kernels (lanes per scanline, block sizes).
python scripts/tune_sga.py  -> prints stage timings per configuration."""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ganet_amd import _native  # noqa: E402

lib = _native.lib()
inp = bench.make_inputs(torch.device("cuda:0"))
shape = os.environ.get("TUNE_SHAPE")
rows = []
for gd, bv, rw in itertools.product([1, 2, 4], [64, 128, 256], [1]):
    lib.set_option("GANET_SGA_GD_V", gd)
    lib.set_option("GANET_SGA_GD_H", 16)
    lib.set_option("GANET_SGA_BLOCK_V", bv)
    lib.set_option("GANET_SGA_ROWWAVE", rw)
    bh = 64
    st = bench.stage_timings(inp, iters=3)
    row = {"gd_v": gd, "block_v": bv, "rowwave": rw,
           "fwd_v": round(st["sga_scan_fwd_down"] + st["sga_scan_fwd_up"], 3),
           "fwd_h": round(st["sga_scan_fwd_right"] + st["sga_scan_fwd_left"], 3),
           "bwdg_v": round(st["sga_bwd_scan_down"] + st["sga_bwd_scan_up"], 3),
           "bwdg_h": round(st["sga_bwd_scan_right"] + st["sga_bwd_scan_left"], 3),
           "fwd_call": round(st["sga_forward_call"], 3), "bwd_call": round(st["sga_backward_call"], 3)}
    rows.append(row)
    print(json.dumps(row), flush=True)
