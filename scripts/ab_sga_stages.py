"""Same-box SGA stage timings (bench.stage_timings: every SGA kernel of the step in sequence) for several builds / option
settings of the library: python scripts/ab_sga_stages.py libA.so libB.so@GANET_SGA_TILED=2 ...  (timing-only ablation builds
included: their results are not checked here)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from ganet_amd import _native
DEFAULTS = {}
RESET = ("GANET_SGA_TILED", "GANET_LGA_MIX", "GANET_LGA_SEGS", "GANET_LGA_WAVE")


def reset_options(lib, libname, defaults):
    """options are process-wide in a loaded library (and a library loaded twice is ONE library): every entry starts from that
    library's own defaults, whatever an earlier entry of the list set"""
    if libname not in defaults:
        defaults[libname] = {}
        for k in RESET:
            try:
                defaults[libname][k] = lib.get_option(k)
            except Exception:                                  # (an older build of the library: no such option / no ganet_get_option)
                pass
    for k, v in defaults[libname].items():
        lib.set_option(k, v)


for rep in range(2):
    for name in sys.argv[1:]:
        libname, _, optstr = name.partition("@")          # lib.so@OPTION=value,OPTION=value
        _native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", libname), strict=False)
        reset_options(_native._LIB, libname, DEFAULTS)
        for kv in filter(None, optstr.split(",")):
            k, v = kv.split("=")
            _native._LIB.set_option(k, int(v))
        inp = bench.make_inputs(torch.device("cuda:0"))
        st = bench.stage_timings(inp, iters=10, only="sga")
        print(name, {k: round(v, 4) for k, v in st.items() if k.startswith("sga")}, flush=True)
