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
