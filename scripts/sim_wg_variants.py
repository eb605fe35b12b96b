"""The workgroup-ring LGA chains on emulator builds with other compile-time ring parameters -- a variant library must be right
before GPU minutes are spent on timing it:  python scripts/sim_wg_variants.py -DLGAP_WG_NR=10
(round-robin and one-wave-ahead schedules, both thread orders, copies and LDS reads landing late, guard pages).
Rounds 4 - 5: NR = 5 / 6 / 7 / 10 -- no failures."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import sim_util, parity_cases as pc
from ganet_amd._native import CApi
from oracle.oracle import Oracle
flags = sys.argv[1].split(',')
so = '/tmp/libganet_sim_variant.so'
cmd = ["g++", "-std=c++17", "-O1", "-DGA_HIPSIM"] + flags + ["-I", os.path.join(sim_util.HERE, "hipsim"), "-I", sim_util.CSRC,
       "-x", "c++", os.path.join(sim_util.CSRC, "ganet_capi.hip"), os.path.join(sim_util.CSRC, "sga_row_tu.hip"), "-shared", "-fPIC", "-o", so]
subprocess.run(cmd, check=True)
sim = CApi(so); o = Oracle("port")
bad = 0
for greedy, order in ((0, 0), (1, 0), (1, 1)):
    sim.set_option("HIPSIM_WAVE_GREEDY", greedy); sim.set_option("HIPSIM_LANE_ORDER", order)
    sim.set_option("HIPSIM_LATE_DMA", 1); sim.set_option("HIPSIM_LATE_LDS", 1)
    for wg in (1,):
        for shape in [(1, 9, 11, 36), (2, 21, 5, 68), (1, 41, 16, 32), (1, 61, 8, 32), (1, 1, 8, 4), (1, 26, 3, 36), (1, 14, 9, 40)]:
            rng = np.random.default_rng(sum(shape)); B, D, H, W = shape
            x = rng.standard_normal(shape).astype(np.float32); f = pc.l1norm(rng.standard_normal((B, 75, H, W)), 1)
            gy = rng.standard_normal(shape).astype(np.float32)
            y, ins = o.lga_chain_forward(x, f, 2, 2); gx, gf = o.lga_chain_backward(ins, f, gy, 2)
            for chain in (pc.check_lga2_paired, pc.check_lga_chain):
                try:
                    chain(sim, pc.NumpyDev("end"), x, f, gy, 2, 2, {"y": y, "gx": gx, "gf": gf})
                except AssertionError as e:
                    bad += 1; print("FAIL", flags, greedy, order, wg, shape, chain.__name__, str(e)[:100])
print(flags, "failures:", bad)
