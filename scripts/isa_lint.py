"""Lint of the gfx950 ISA hipcc emits for the kernels (compiler quirks that cost whole memory round trips):
  * global_load followed within six instructions by s_waitcnt vmcnt(0)  -> a load under a condition
  * scratch_load / scratch_store                                        -> spills or dynamically indexed local arrays
python scripts/isa_lint.py [name-substring ...]   (compiles both translation units with -S)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ganet_amd import build

want = sys.argv[1:]
for src, extra in build.SOURCES.items():
    with tempfile.NamedTemporaryFile(suffix=".s") as tf:
        subprocess.run(["hipcc"] + build.HIPCC_FLAGS + extra + ["-I", os.environ.get("ISA_CSRC", build.CSRC), "-S", "--cuda-device-only",
                        os.path.join(os.environ.get("ISA_CSRC", build.CSRC), src), "-o", tf.name], check=True, stderr=subprocess.DEVNULL)
        txt = open(tf.name).read().split("\n")
    cur, res = None, {}
    for i, l in enumerate(txt):
        m = re.match(r"^(_ZN2ga\S+):", l)
        if m:
            cur = m.group(1); res[cur] = {"loads": 0, "serial": 0, "scratch": 0}
            continue
        if cur is None:
            continue
        if "scratch_" in l:
            res[cur]["scratch"] += 1
        if "global_load" in l and "lds" not in l:
            res[cur]["loads"] += 1
            k, j = 0, i + 1
            while k < 6 and j < len(txt):
                t = txt[j].strip()
                if t and not t.startswith(";") and not t.startswith("."):
                    k += 1
                    if "vmcnt(0)" in t:
                        res[cur]["serial"] += 1
                        break
                    if "global_load" in t:
                        break
                j += 1
        if "s_endpgm" in l:
            cur = None
    for k, v in res.items():
        if want and not any(w in k for w in want):
            continue
        if v["serial"] or v["scratch"] or want:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"{name[:90]:90s} loads={v['loads']:4d} serialised={v['serial']:3d} scratch_ops={v['scratch']:3d}")
