"""BASELINE configs[4], per-GPU slice: ONE GANet-deep training step (forward, loss mix, backward, Adam) at the SceneFlow frame size
960x528, max_disp 192, two samples per GPU (batch 16 over 8 GPUs), fused call sites, MIOpen find mode (the first step times
MIOpen's solvers for ~200 new convolution shapes: minutes; the winners go to miopen_cache/ and come back with gpurun_out/).
    python scripts/check_cfg5_train.py <out.json> [steps] [warmup]"""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
steps = sys.argv[2] if len(sys.argv) > 2 else "2"
warmup = sys.argv[3] if len(sys.argv) > 3 else "1"
t0 = time.time()
r = subprocess.run([sys.executable, "-m", "harness.train", "--crop_height", "528", "--crop_width", "960", "--batch", "2",
                    "--steps", steps, "--warmup", warmup, "--fused"], cwd=ROOT, capture_output=True, text=True)
wall = time.time() - t0
lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
res = json.loads(lines[-1]) if lines else {"error": r.stderr[-2000:]}
res["wall_s_including_miopen_find"] = round(wall, 1)
res["rc"] = r.returncode
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
# bring MIOpen's user db back with the results (gpurun merges gpurun_out/ only)
src = os.path.join(ROOT, "miopen_cache")
dst = os.path.join(os.path.dirname(os.path.abspath(out)), "miopen_cache")
if os.path.isdir(src) and sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(src) for f in fs) < 40 << 20:
    shutil.copytree(src, dst, dirs_exist_ok=True)
