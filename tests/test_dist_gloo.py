"""world_size-2 gloo tests (CPU) of the multi-rank plumbing bench.py relies on: the timing
protocol (barrier + max over ranks), unit sharding with no data-path collective, and the
bucketed gradient mean."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from ganet_amd import dist as gdist
    ctx = gdist.init(world, backend="gloo")
    # rank 1 is slower: every rank must report rank 1's time
    t = gdist.timed_region(ctx, lambda: time.sleep(0.05 + 0.2 * rank))
    lo, hi = gdist.shard_range(7, ctx)
    grads = [torch.full((5,), float(rank + 1)), torch.full((3, 2), float(10 * (rank + 1)))]
    gdist.all_reduce_mean_(grads, ctx, bucket_bytes=16)
    probe = gdist.allreduce_probe(ctx, "gloo", torch.device("cpu"), nelem=1000, iters=2)
    assert probe["ranks"] == world and probe["result_ok"] and probe["bytes"] == 4000, probe
    bad = gdist.allreduce_probe(ctx, "no-such-backend", torch.device("cpu"), nelem=10, iters=1)
    assert "error" in bad, bad                      # reported, never fatal
    q.put((rank, t, (lo, hi), [g.tolist() for g in grads]))      # plain lists: a tensor in the queue is handed over
    # through the sender's resource-sharer socket, which is gone if this process exits before the parent reads it
    gdist.finish(ctx)


def test_two_rank_timing_sharding_and_grad_mean():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t0, t1 = res[0][1], res[1][1]
    assert abs(t0 - t1) < 1e-9 and t0 >= 0.25, "every rank must see the max-over-ranks time"
    assert res[0][2] == (0, 4) and res[1][2] == (4, 7)
    for _, _, _, grads in res:
        assert torch.allclose(torch.tensor(grads[0]), torch.full((5,), 1.5))
        assert torch.allclose(torch.tensor(grads[1]), torch.full((3, 2), 15.0))


def test_world_size_mismatch_is_an_error(monkeypatch):
    sys.path.insert(0, ROOT)
    from ganet_amd import dist as gdist
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(RuntimeError, match="torch.distributed.run"):
        gdist.init(4)


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_gpus2_plumbing_with_stub_step(launcher):
    """`python bench.py --gpus 2` must work BOTH ways: on its own (it re-executes itself under torch.distributed.run)
    and when the driver launches it under torch.distributed.run -- here on CPU/gloo with the placeholder step
    (--stub-step); exactly one JSON line, from rank 0, with the protocol's fields."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    tail = [bench, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port())] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert r.stdout.strip() == lines[0], "stdout must carry the JSON line only: " + r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 3e-3)) <= 0.01 * line["value"]
    # the gradient all-reduce probe bench.py runs after the timed region (RCCL on GPUs; here its code path over gloo)
    probe = line["rccl"]
    assert "error" not in probe, probe
    assert probe["ranks"] == 2 and probe["result_ok"] and probe["allreduce_ms"] > 0
    assert abs(probe["busbw_GBs"] - probe["algbw_GBs"]) <= 0.02 * probe["algbw_GBs"] + 0.01      # 2 (N-1)/N = 1 at N = 2


def _run_stub_bench(extra_env, timeout=300):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


def test_probe_survives_a_late_rank0():
    """rank 0 reaches the all-reduce probe 10 s after rank 1 (it used to run its extra measurements first,
    with the other ranks already inside a collective that has a timeout).  The probe runs in child processes on a rendezvous
    of their own and before rank 0's extras: the late rank only makes the children's rendezvous wait."""
    line, err = _run_stub_bench({"GANET_BENCH_TEST_DELAY_RANK0": "10"})
    assert line["value"] > 0
    assert "error" not in line["rccl"], line["rccl"]
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["result_ok"]
    assert "[bench] measured:" in err            # the value was on record before the probe started


def test_hung_probe_costs_the_line_nothing():
    """a collective that never completes must not take the measured value with it.  Rank 0's probe child
    sleeps past the parent's patience: the parent kills exactly that child, reports the error inside `rccl`, and the ONE JSON
    line carries the value all the same."""
    line, _ = _run_stub_bench({"GANET_PROBE_TEST_DELAY_RANK0": "120", "GANET_BENCH_PROBE_TIMEOUT": "8"}, timeout=200)
    assert line["value"] > 0 and line["n_gpus"] == 2
    assert "error" in line["rccl"] and "killed" in line["rccl"]["error"], line["rccl"]


def _skip_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from harness import steps
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Toy(torch.nn.Module):                      # two disparity maps, like GANet11
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(6, 2, 3, padding=1)
            self.never_called = torch.nn.Linear(3, 3)    # like GANet_deep's cost_agg.deconv0b (constructed, never used)
            self.blow = 0.0

        def forward(self, left, right):
            o = self.c(torch.cat([left, right], 1))
            o = o + self.blow * o                        # blow = inf: a non-finite activation on this rank
            return o[:, 0], o[:, 1]
    torch.manual_seed(0)
    model = DDP(Toy(), find_unused_parameters=True)      # what harness/train.py sets for GANet_deep (the empty-shard
    # loss used to touch the unused parameters directly -> "Expected to mark a variable ready only once" on the empty rank)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    max_disp = 24
    left, right, target = steps.synthetic_batch(1, 8, 12, max_disp, "cpu", seed=rank)
    crit = lambda a, b: torch.nn.functional.smooth_l1_loss(a, b)      # noqa: E731
    out = []
    # step 1: rank 1 has no valid pixel (its shard alone would be skipped), rank 0 has: both must run the step
    t1 = torch.full_like(target, float(max_disp)) if rank == 1 else target
    out.append(steps.train_step(model, opt, "GANet11", left, right, t1, max_disp, crit)[0] is not None)
    # step 2: no rank has a valid pixel: both skip
    out.append(steps.train_step(model, opt, "GANet11", left, right, torch.full_like(target, float(max_disp)), max_disp, crit)[0] is None)
    digest = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
    # step 3: the zero-weighted rank's forward overflows: 0 * inf would put NaN into everybody's averaged gradients -- the step
    # is skipped collectively, parameters stay as they were (and finite) on both ranks
    if rank == 1:
        model.module.blow = float("inf")
    out.append(steps.train_step(model, opt, "GANet11", left, right, t1, max_disp, crit)[0] is not None)
    after = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
    out.append(after == digest and bool(all(torch.isfinite(p).all() for p in model.parameters())))
    q.put((rank, out, digest))
    dist.destroy_process_group()


def test_batch_without_valid_pixels_is_skipped_collectively():
    """harness.steps.train_step under DDP: a rank whose shard has no valid pixel must not leave the others
    waiting in the gradient all-reduce; the step is skipped only when every rank is empty."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_skip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [True, True, True, True] and res[1][1] == [True, True, True, True]
    assert res[0][2] == res[1][2], "identical parameters after the shared step"
