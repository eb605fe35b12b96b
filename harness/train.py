#!/usr/bin/env python
"""BASELINE.json configs[3] (per-GPU slice of it on one GPU): the GANet-deep training step -- forward, loss mix,
backward, DDP gradient all-reduce over RCCL, Adam -- at crop 240x624, max_disp 192, one sample per GPU (batch 8 over
8 GPUs), the reference's model on this repository's ops.

    python -m harness.train [--gpus N] [--model GANet_deep] [--crop_height 240] [--crop_width 624] [--max_disp 192]
                            [--batch 1] [--steps 5] [--warmup 2] [--sync_bn] [--fused] [--resume CKPT] [--save CKPT]

N > 1: launched under torch.distributed.run (one rank per GPU) or, without a launcher, re-executes itself that way.
Rank 0 prints one JSON line: ms per step (max over ranks), samples/s of the whole job, peak memory, loss values."""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from harness.miopen_env import use_repo_miopen_cache  # noqa: E402

use_repo_miopen_cache()            # before torch loads MIOpen

import torch  # noqa: E402

from ganet_amd import dist as gdist  # noqa: E402
from harness import fuse, steps  # noqa: E402


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--model", default="GANet_deep")
    ap.add_argument("--crop_height", type=int, default=240)
    ap.add_argument("--crop_width", type=int, default=624)
    ap.add_argument("--max_disp", type=int, default=192)
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU (train.sh: 8 samples over 8 GPUs)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--kitti", type=int, default=1)
    ap.add_argument("--sync_bn", action="store_true")
    ap.add_argument("--no_miopen_find", dest="miopen_find", action="store_false",
                    help="MIOpen immediate mode (heuristic solver choice) instead of timing its solvers per shape")
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--kernel_share", action="store_true", help="add the per-group device-time table (torch.profiler over 2 extra steps, rank 0)")
    ap.add_argument("--resume", default="")
    ap.add_argument("--save", default="")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    return ap.parse_args(argv)


def run(args, hook=None):
    """The job of one rank.  hook(model): see steps.build_model (tests only)."""
    cpu = args.device == "cpu"
    ctx = gdist.init(args.gpus, backend="gloo" if cpu else None)
    if cpu:
        dev = torch.device("cpu")
    else:
        dev = torch.device("cuda", ctx.local_rank)
        torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)     # MIOpen find mode (harness/miopen_env.py)
    torch.manual_seed(123)                      # same initial weights on every rank (DDP broadcasts rank 0's anyway)
    model = steps.build_model(args.model, args.max_disp, dev, sync_bn=args.sync_bn and ctx.world_size > 1,
                              ddp=ctx.world_size > 1, local_rank=ctx.local_rank, hook=hook)
    if args.fused:
        fuse.use_fused_ops(steps.unwrap(model))
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(0.9, 0.999))
    epoch0 = 0
    if args.resume:
        epoch0, _ = steps.load_checkpoint(args.resume, model, None)
    crit = steps.criterion(bool(args.kitti))
    # every rank its own samples: the batch dimension is what shards (no data-path collective inside the ops)
    left, right, target = steps.synthetic_batch(args.batch, args.crop_height, args.crop_width, args.max_disp, dev,
                                                seed=123 + ctx.rank)
    sync = (lambda: None) if cpu else torch.cuda.synchronize
    losses = []
    for _ in range(args.warmup):
        steps.train_step(model, opt, args.model, left, right, target, args.max_disp, crit)
    if not cpu:
        torch.cuda.reset_peak_memory_stats()

    def timed():
        for _ in range(args.steps):
            loss, err = steps.train_step(model, opt, args.model, left, right, target, args.max_disp, crit)
            if loss is not None:      # (a step every rank skipped: no valid pixel anywhere)
                losses.append(float(loss))

    elapsed = gdist.timed_region(ctx, timed, sync=sync)
    share = None
    if args.kernel_share and ctx.rank == 0 and not cpu and ctx.world_size == 1:
        from harness.kernel_share import profile_passes
        share = profile_passes(lambda: steps.train_step(model, opt, args.model, left, right, target, args.max_disp, crit), 2)
    if args.save and ctx.rank == 0:
        steps.save_checkpoint(args.save, model, opt, epoch0 + 1)
    line = {"what": "training step, reference model on the drop-in ops", "model": args.model, "n_gpus": ctx.world_size,
            "per_gpu_batch": args.batch, "crop": [args.crop_height, args.crop_width], "max_disp": args.max_disp,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "samples_per_sec": round(ctx.world_size * args.batch * args.steps / elapsed, 3),
            "steps": args.steps, "warmup": args.warmup, "miopen_find": bool(args.miopen_find), "sync_bn": bool(args.sync_bn and ctx.world_size > 1),
            "ops": "ganet_amd.modules.fused" if args.fused else "drop-in call forms (libs/)",
            "grad_allreduce": "DistributedDataParallel (RCCL)" if ctx.world_size > 1 and not cpu else
                              ("DistributedDataParallel (gloo)" if ctx.world_size > 1 else "none (1 rank)"),
            "peak_mem_GB": None if cpu else round(torch.cuda.max_memory_allocated() / 2 ** 30, 3),
            "loss_first_last": [round(losses[0], 5), round(losses[-1], 5)] if losses else None, "dtype": "f32",
            "data": "synthetic", "weights": "random init" if not args.resume else args.resume, "kernel_share": share}
    gdist.finish(ctx)
    if ctx.rank == 0:
        print(json.dumps(line))
    return line


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)
    run(args)


if __name__ == "__main__":
    main()
