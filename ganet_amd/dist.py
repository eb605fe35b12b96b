"""One-process-per-GPU plumbing for the benchmark and for data-parallel callers.

The guided-aggregation ops have no cross-sample term (every index in the reference kernels is
per-(n,c): GANet_kernel.cu:77-78, 1148-1152), so the hot path shards over the batch with NO
data-path collective: each rank owns whole samples.  What is collective is only (a) the timing
protocol of bench.py (barrier + max-over-ranks) and (b), for training callers, the parameter
gradient all-reduce, for which `all_reduce_mean_` below is the bucketed RCCL form
(backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests)."""
import datetime
import os
import time
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class DistCtx:
    rank: int
    local_rank: int
    world_size: int
    initialized_here: bool


def init(expected_world=None, backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); world 1 needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if expected_world is not None and expected_world != world:
        if world == 1 and expected_world > 1:
            raise RuntimeError(f"--gpus {expected_world} needs a torch.distributed.run launch "
                               f"(one rank per GPU); WORLD_SIZE is {world}")
        raise RuntimeError(f"--gpus {expected_world} but WORLD_SIZE={world}")
    started = False
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        started = True
    return DistCtx(rank, local, world, started)


def barrier(ctx):
    if ctx.world_size > 1:
        dist.barrier()


def max_over_ranks(ctx, value, device=None):
    if ctx.world_size == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", ctx.local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_region(ctx, fn, sync=lambda: None):
    """barrier + device sync, run fn, device sync + barrier; returns the MAX wall time over ranks."""
    barrier(ctx)
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    local = time.perf_counter() - t0
    barrier(ctx)
    return max_over_ranks(ctx, local)


def shard_range(n_items, ctx):
    """Contiguous, near-equal split of n_items independent units (samples) over ranks."""
    base, extra = divmod(n_items, ctx.world_size)
    start = ctx.rank * base + min(ctx.rank, extra)
    return start, start + base + (1 if ctx.rank < extra else 0)


def all_reduce_mean_(tensors, ctx, bucket_bytes=64 << 20):
    """In-place mean of gradient tensors across ranks, in flat buckets (few, large collectives:
    xGMI rings are per-link bound, so bucket size -- not call count -- sets the rate)."""
    if ctx.world_size == 1:
        return
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(ctx.world_size)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        bucket, size = [], 0

    for t in tensors:
        bucket.append(t)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


GRAD_ELEMS_GANET_DEEP = 6580112      # fp32 parameters of GANet-deep = its gradient all-reduce (26.3 MB)


def allreduce_probe(ctx, backend, device, nelem=GRAD_ELEMS_GANET_DEEP, iters=10, bucket_bytes=64 << 20):
    """The one collective a data-parallel caller of this path runs per step -- the parameter-gradient mean (the reference:
    nn.DataParallel's reduce, train.py:73) -- measured on its own: `nelem` fp32 values through all_reduce_mean_ on a process
    group of `backend` ("nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).  Every rank calls it.  Returns
    {backend, ranks, bytes, allreduce_ms, algbw_GBs, busbw_GBs} (busbw = 2 (N-1)/N x bytes / time: per-link traffic of a
    ring), max over ranks, or {"error": ...} when the group cannot be made (the benchmark's own number never depends on it)."""
    if ctx.world_size == 1:
        return None
    try:
        # (a bounded wait: a collective that cannot complete becomes an error string after a minute, not a hung benchmark)
        group = dist.new_group(backend=backend, timeout=datetime.timedelta(seconds=60)) if dist.get_backend() != backend else None
        g = torch.ones(nelem, dtype=torch.float32, device=device) * (ctx.rank + 1)

        def once():
            if group is None:
                all_reduce_mean_([g], ctx, bucket_bytes)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
                g.div_(ctx.world_size)
        sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
        once()                                           # connection set-up is not part of the rate
        sync()
        want = (ctx.world_size + 1) / 2.0                # mean of rank + 1
        ok = bool(torch.allclose(g[:16].float().cpu(), torch.full((16,), want)))
        g.fill_(1.0)
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(iters):
            once()
        sync()
        dt = max_over_ranks(ctx, (time.perf_counter() - t0) / iters, device=torch.device("cpu") if dist.get_backend() == "gloo" else None)
        nbytes = nelem * 4
        return {"backend": "rccl (nccl)" if backend == "nccl" else backend, "ranks": ctx.world_size, "bytes": nbytes,
                "allreduce_ms": round(1e3 * dt, 4), "algbw_GBs": round(nbytes / dt / 1e9, 2),
                "busbw_GBs": round(2.0 * (ctx.world_size - 1) / ctx.world_size * nbytes / dt / 1e9, 2), "result_ok": ok}
    except Exception as e:                               # noqa: BLE001  (reported, never fatal for the op benchmark)
        return {"backend": backend, "ranks": ctx.world_size, "error": f"{type(e).__name__}: {e}"[:300]}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def allreduce_probe_isolated(ctx, backend, device_index, timeout_s=150.0, nelem=GRAD_ELEMS_GANET_DEEP, iters=10):
    """allreduce_probe() in CHILD processes, one per rank, on a rendezvous of their own (a fresh port chosen by rank 0 and
    handed round over the caller's group).  Why not in-process: a collective that hangs does not become an error string -- the
    device sync blocks and RCCL's watchdog aborts the process after its timeout -- and a failure on ONE rank (new_group, the
    allocation) leaves the others waiting in the next collective; either way the benchmark's already measured value would be
    lost or badly delayed.  Here every rank waits for its own child for at most `timeout_s`, kills exactly that child if it is
    still alive (by pid), and returns; the caller's process group never sees the probe.  Every rank calls it; rank 0 gets the
    child's JSON object (or {"error": ...}), the others None."""
    import json
    import subprocess
    import sys
    if ctx.world_size == 1:
        return None
    port = [_free_port() if ctx.rank == 0 else None]
    dist.broadcast_object_list(port, src=0)
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port[0]), "RANK": str(ctx.rank), "WORLD_SIZE": str(ctx.world_size),
                "LOCAL_RANK": str(ctx.local_rank)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC between the ranks' GPUs (the host driver has no legacy IPC)
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
        env.pop(k, None)                                    # the child's rendezvous is a plain TCP store of its own
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "ganet_amd.dist", "--probe", backend, str(device_index), str(nelem), str(iters)]
    res = {"backend": backend, "ranks": ctx.world_size}
    try:
        child = subprocess.Popen(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            out, err = child.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            child.kill()                                    # (this child, by pid)
            out, err = child.communicate()
            res["error"] = f"probe child of rank {ctx.rank} still running after {timeout_s:.0f} s: killed"
            out = ""
        if "error" not in res:
            line = next((ln for ln in reversed(out.strip().splitlines()) if ln.startswith("{")), None)
            if line is not None:
                res = json.loads(line)
            elif ctx.rank == 0:
                res["error"] = f"probe child exited with {child.returncode}: {(err or '').strip()[-300:]}"
    except Exception as e:                                   # noqa: BLE001
        res["error"] = f"{type(e).__name__}: {e}"[:300]
    if ctx.rank != 0:
        return None
    res["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY")
    res["isolated"] = "child processes, own rendezvous"
    return res


def _probe_main(argv):
    """child of allreduce_probe_isolated: `python -m ganet_amd.dist --probe <backend> <device index> <nelem> <iters>`"""
    import json
    backend, dev_index, nelem, iters = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    delay = float(os.environ.get("GANET_PROBE_TEST_DELAY_RANK0", "0"))      # tests: rank 0 arrives late
    if delay and int(os.environ.get("RANK", "0")) == 0:
        time.sleep(delay)
    device = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(device)
    try:
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]),
                                timeout=datetime.timedelta(seconds=60))
        # (local_rank = the device this rank computes on: max_over_ranks puts its scalar there under nccl)
        ctx = DistCtx(int(os.environ["RANK"]), dev_index if backend == "nccl" else int(os.environ.get("LOCAL_RANK", "0")),
                      int(os.environ["WORLD_SIZE"]), True)
        res = allreduce_probe(ctx, backend, device, nelem=nelem, iters=iters)
    except Exception as e:                                   # noqa: BLE001
        res = {"backend": backend, "error": f"{type(e).__name__}: {e}"[:300]}
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(res), flush=True)
    try:
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:                                        # noqa: BLE001
        pass


def finish(ctx):
    if ctx.initialized_here and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 6 and sys.argv[1] == "--probe":
        _probe_main(sys.argv[2:])
