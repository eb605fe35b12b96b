"""Same-box A/B of the WHOLE benchmark step (bench.one_step captured into a hipGraph, replayed) for several builds of the library
(scripts/build_variants.py):  python scripts/ab_step.py libA.so libB.so ...   -> ms per step per library, interleaved repetitions.
Every graph uses the SAME input tensors and is captured into the SAME memory pool, so corresponding buffers of all variants sit
at the same addresses: the same library captured twice with buffers of its own differs by up to 2.6 % (placement of the volumes in
the memory channels, profiles/r3z_*), which would drown a 1 % effect."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ganet_amd import _native

dev = torch.device("cuda:0")
graphs = {}
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


inp = bench.make_inputs(dev)
pool = torch.cuda.graph_pool_handle()
for idx, name in enumerate(sys.argv[1:]):
    libname, _, optstr = name.partition("@")          # lib.so@OPTION=value,OPTION=value: ganet_set_option before the capture
    _native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", libname), strict=False)
    os.environ.pop("GANET_LGA_EDGES", None)
    # options are process-wide in a loaded library: every entry starts from that library's own defaults
    reset_options(_native._LIB, libname, DEFAULTS)
    for kv in filter(None, optstr.split(",")):
        k, v = kv.split("=")
        if k in ("GANET_LGA_PAIRED", "GANET_SGA_SAVE", "GANET_LGA_EDGES"):      # read by the Python layer from the environment at every call
            os.environ[k] = v
        else:
            _native._LIB.set_option(k, int(v))
    name = f"{idx}:{name}"
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            bench.one_step(inp)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool):
        keep = bench.one_step(inp)
    del keep                      # the next capture reuses the same blocks of the shared pool
    graphs[name] = (g, inp, None)
res = {n: [] for n in graphs}
for rep in range(5):
    for name, (g, _, _) in graphs.items():
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record(); e1.synchronize()
        res[name].append(e0.elapsed_time(e1) / 50)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:32s} median {v[len(v)//2]:.4f} ms  min {v[0]:.4f}  max {v[-1]:.4f}", flush=True)
