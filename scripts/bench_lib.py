"""bench.py's own measurement (hipGraph of one step, 20 replays between synchronisations) with a chosen build of the library:
   python scripts/bench_lib.py libganet_hip_<tag>.so [bench.py flags]      (same-box comparisons of what the driver runs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the library: one HIP runtime)
from ganet_amd import _native
_native._LIB = _native.CApi(os.path.join(ROOT, "ganet_amd", sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + (sys.argv[2:] or ["--no-cpu-baseline", "--no-roofline", "--no-overlap"])
import bench
bench.main()
