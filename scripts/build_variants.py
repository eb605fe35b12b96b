"""Builds tagged variants of the library for same-box A/B runs:
python scripts/build_variants.py tag1:-DFOO=1,-DBAR=2 tag2:-DFOO=3 ...  -> ganet_amd/libganet_hip_<tag>.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from ganet_amd import build
def one(spec):
    tag, _, flags = spec.partition(":")
    out = os.path.join(ROOT, "ganet_amd", f"libganet_hip_{tag}.so")
    # objects are written next to the sources: give every variant its own object names
    import subprocess, shutil
    objs = []
    for src, extra in build.SOURCES.items():
        obj = os.path.join(build.CSRC, f"{src}.{tag}.o")
        cmd = ["hipcc"] + build.HIPCC_FLAGS + extra + [f for f in flags.split(",") if f] + ["-I", build.CSRC, "-c", os.path.join(build.CSRC, src), "-o", obj]
        subprocess.run(cmd, check=True); objs.append(obj)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC", "-Wl,-soname," + os.path.basename(out)] + objs + ["-o", out], check=True)
    for o in objs: os.remove(o)
    return out
with ThreadPoolExecutor(4) as ex:
    for o in ex.map(one, sys.argv[1:]): print(o)
