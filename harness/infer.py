#!/usr/bin/env python
"""BASELINE.json configs[2]: full GANet-deep inference (9 GA layers... 7 SGA + 2 LGA2) on a KITTI-2015-sized pair
(1248x384, max_disp 192) on one MI355X, the reference's model on this repository's ops.

    python -m harness.infer [--model GANet_deep] [--height 384] [--width 1248] [--max_disp 192] [--fused] [--iters 5]

Prints one JSON line: ms per forward pass (median of --iters, HIP events), peak device memory, parameter count.
Random-init weights and synthetic standardised images (no checkpoint / dataset offline); predict.py:100-114 is the
step.  For the GA-op share of the pass run it under `rocprofv3 --kernel-trace --stats` and feed the kernel stats to
scripts/model_kernel_share.py."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from harness.miopen_env import use_repo_miopen_cache  # noqa: E402

use_repo_miopen_cache()            # before torch loads MIOpen

import torch  # noqa: E402

from harness import fuse, steps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="GANet_deep")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=1248)
    ap.add_argument("--max_disp", type=int, default=192)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--fused", action="store_true", help="ganet_amd.modules.fused op chains instead of the stock call forms")
    ap.add_argument("--no_miopen_find", dest="miopen_find", action="store_false",
                    help="MIOpen immediate mode (heuristic solver choice) instead of timing its solvers per shape")
    ap.add_argument("--kernel_share", action="store_true", help="add the per-group device-time table (torch.profiler over 2 extra passes)")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "harness.infer needs a GPU"
    from ganet_amd import _native
    assert not _native.lib().is_simulator
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = bool(args.miopen_find)     # MIOpen find mode (harness/miopen_env.py)
    torch.manual_seed(123)
    model = steps.build_model(args.model, args.max_disp, dev)
    n_fused = fuse.use_fused_ops(model) if args.fused else 0
    left, right, _ = steps.synthetic_batch(args.batch, args.height, args.width, args.max_disp, dev)
    for _ in range(args.warmup):
        out = steps.predict(model, left, right)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    times = []
    for _ in range(args.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = steps.predict(model, left, right)
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
    times.sort()
    assert out.shape == (args.batch, args.height, args.width) and bool(torch.isfinite(out).all())
    share = None
    if args.kernel_share:
        from harness.kernel_share import profile_passes
        share = profile_passes(lambda: steps.predict(model, left, right), 2)
    print(json.dumps({
        "what": "full-model inference, reference model on the drop-in ops", "model": args.model,
        "input": [args.batch, 3, args.height, args.width], "max_disp": args.max_disp,
        "ops": "ganet_amd.modules.fused (%d call sites)" % n_fused if args.fused else "drop-in call forms (libs/)",
        "ms_per_pair": round(times[len(times) // 2], 3), "ms_min": round(times[0], 3), "iters": args.iters,
        "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 3),
        "miopen_find": bool(args.miopen_find), "params": sum(p.numel() for p in model.parameters()), "dtype": "f32", "weights": "random init",
        "disp_range": [round(float(out.min()), 3), round(float(out.max()), 3)],
        "device": torch.cuda.get_device_name(0), "kernel_share": share}))


if __name__ == "__main__":
    main()
