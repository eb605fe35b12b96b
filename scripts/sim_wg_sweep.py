"""One-off emulator sweep of the workgroup-ring LGA chains (GANET_LGA_WG = 1 | 2): every depth 1 .. 71, pair-interleaved and API-layout chains,
guard pages at either end, copies and LDS reads landing late, round-robin / one-wave-ahead schedules in both thread orders.  Round 4: 340 runs, 0 failures.
python scripts/sim_wg_sweep.py   (about three minutes; the test suite holds a subset: tests/test_sim_bounds.py)"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import parity_cases as pc
from sim_util import sim_api
from oracle.oracle import Oracle
o=Oracle("port"); sim=sim_api()
sim.set_option("HIPSIM_LATE_DMA",1); sim.set_option("HIPSIM_LATE_LDS",1)
bad=0; n=0
for D in range(1,72):
    for (B,H,W) in ((1,11,36),) + (((2,3,8),) if D%5==0 else ()):
        shape=(B,D,H,W)
        rng=np.random.default_rng(D*7+W)
        x=rng.standard_normal(shape).astype(np.float32); f=pc.l1norm(rng.standard_normal((B,75,H,W)),1); gy=rng.standard_normal(shape).astype(np.float32)
        y,ins=o.lga_chain_forward(x,f,2,2); gx,gf=o.lga_chain_backward(ins,f,gy,2)
        for wg in (1,2):
            sim.set_option("GANET_LGA_WG",wg)
            sim.set_option("HIPSIM_WAVE_GREEDY", D%2); sim.set_option("HIPSIM_LANE_ORDER", (D//2)%2)
            for chain in (pc.check_lga2_paired, pc.check_lga_chain):
                n+=1
                try: chain(sim, pc.NumpyDev("end" if D%3 else "start"), x,f,gy,2,2,{"y":y,"gx":gx,"gf":gf})
                except AssertionError as e: bad+=1; print("FAIL",shape,wg,chain.__name__,str(e)[:120])
print("runs",n,"failures",bad)
