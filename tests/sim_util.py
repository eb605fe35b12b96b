"""TEST INFRASTRUCTURE: builds and binds tests/hipsim/libganet_sim.so -- the product's
kernel sources (ganet_amd/csrc/*.h, ganet_capi.hip) compiled with g++ against the
lockstep wave64 emulator in tests/hipsim/hipsim.h -- so kernel logic can be checked on a
machine without a GPU.  Never imported by ganet_amd itself."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "ganet_amd", "csrc")
SIM_SO = os.path.join(HERE, "hipsim", "libganet_sim.so")


def build_sim():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hipsim", "hipsim.h"),
                                                                os.path.join(ROOT, "include", "ganet_hip.h")]
    if os.path.exists(SIM_SO) and all(os.path.getmtime(d) <= os.path.getmtime(SIM_SO) for d in deps):
        return SIM_SO
    cmd = ["g++", "-std=c++17", "-O1", "-DGA_HIPSIM", "-I", os.path.join(HERE, "hipsim"), "-I", CSRC,
           "-x", "c++", os.path.join(CSRC, "ganet_capi.hip"), os.path.join(CSRC, "sga_row_tu.hip"),
           "-shared", "-fPIC", "-o", SIM_SO]
    subprocess.run(cmd, check=True)
    return SIM_SO


def sim_api():
    from ganet_amd._native import CApi
    api = CApi(build_sim())
    assert api.is_simulator
    return api


def ptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data
