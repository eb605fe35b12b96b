"""CPU tests of the full-model harness: the reference's models import on top of the drop-in `libs/`,
run and differentiate on the CPU with the GA ops routed through the C oracle, checkpoints keep the reference's key
format, and the world-size-2 DDP training step works over gloo."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from harness import fuse, refmodel, steps  # noqa: E402

pytestmark = pytest.mark.skipif(not refmodel.available(), reason="no reference model code (GANET_REF_ROOT, "
                                "/root/reference or oracle/_ref/pyref)")


def _hook(ora):
    from oracle.cpu_ops import route_cpu_through_oracle
    return lambda m: route_cpu_through_oracle(m, ora)


@pytest.mark.parametrize("name,n_ops,n_out", [("GANet11", 10, 2), ("GANet_deep", 14, 3)])
def test_reference_models_run_on_cpu_through_the_oracle(port_oracle, name, n_ops, n_out):
    """GANet(48) at a 48x96 crop: SGA x4/x7, LGA2, LGA, LGA3, GetCostVolume and DisparityRegression instances are all
    found and rebound; training mode returns the 2 / 3 disparities of the model, eval mode one; a loss.backward() reaches
    every parameter."""
    torch.manual_seed(0)
    counts = []
    model = steps.build_model(name, 48, "cpu", hook=lambda m: counts.append(_hook(port_oracle)(m)))
    assert counts == [n_ops]
    import libs.GANet.modules.GANet as drop_in
    assert os.path.realpath(drop_in.__file__).startswith(os.path.realpath(ROOT))
    left, right, target = steps.synthetic_batch(1, 48, 96, 48, "cpu")
    model.train()
    outs = model(left, right)
    assert len(outs) == n_out and all(o.shape == (1, 48, 96) for o in outs)
    loss = steps.loss_mix(name, outs, target, target < 48, steps.criterion(True))
    loss.backward()
    missing = [k for k, p in model.named_parameters() if p.grad is None]
    # GANet_deep builds cost_agg.deconv0b and never calls it (models/GANet_deep.py:306); everything else gets a gradient
    assert [k for k in missing if not k.startswith("cost_agg.deconv0b.")] == [], missing
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    d = steps.predict(model, left, right)
    assert d.shape == (1, 48, 96) and bool(torch.isfinite(d).all())


def test_checkpoint_keys_follow_the_reference(tmp_path, port_oracle):
    """train.py:193-197 saves the DataParallel wrapper's state_dict (`module.` prefix) with 'epoch' and 'optimizer';
    predict.py:60 / train.py:79 load it with strict=False.  Round trip through both a bare and a wrapped model."""
    torch.manual_seed(1)
    model = steps.build_model("GANet11", 48, "cpu")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    path = str(tmp_path / "ck" / "kitti_epoch_1.pth")
    steps.save_checkpoint(path, model, opt, 7)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 7
    assert all(k.startswith("module.") for k in ck["state_dict"])
    assert "module.cost_agg.sga1.conv_refine.conv.weight" in ck["state_dict"]
    other = steps.build_model("GANet11", 48, "cpu")
    epoch, res = steps.load_checkpoint(path, other)
    assert epoch == 7 and not res.missing_keys and not res.unexpected_keys
    for (k, a), (_, b) in zip(model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k

    class Wrapper(torch.nn.Module):          # anything with .module, like DataParallel / DDP
        def __init__(self, m):
            super().__init__()
            self.module = m
    third = Wrapper(steps.build_model("GANet11", 48, "cpu"))
    res = steps.load_state_dict_compat(third, {k[len("module."):]: v for k, v in ck["state_dict"].items()})
    assert not res.missing_keys and not res.unexpected_keys


def test_fused_call_sites_leave_parameters_and_keys_alone():
    model = steps.build_model("GANet_deep", 48, "cpu")
    keys = list(model.state_dict())
    n_mod = sum(1 for _ in model.modules())
    assert fuse.use_fused_ops(model) == 7 + 2 + 1      # SGABlocks, Disp x2, DispAgg
    assert list(model.state_dict()) == keys and sum(1 for _ in model.modules()) == n_mod


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    torch.set_num_threads(2)
    from harness import train as htrain
    from oracle.cpu_ops import route_cpu_through_oracle
    from oracle.oracle import Oracle
    ora = Oracle("port")
    args = htrain.parse(["--gpus", str(world), "--model", "GANet11", "--crop_height", "48", "--crop_width", "96",
                         "--max_disp", "24", "--steps", "2", "--warmup", "0", "--device", "cpu"])
    captured = {}

    def hook(m):
        route_cpu_through_oracle(m, ora)
        captured["model"] = m
    line = htrain.run(args, hook=hook)
    params = dict(captured["model"].named_parameters())      # parameters only: BN running statistics are per rank here
    digest = float(sum(v.detach().double().abs().sum() for v in params.values()))
    first = params["conv_start.0.conv.weight"].detach().flatten()[:8].tolist()
    q.put((rank, line["loss_first_last"], digest, first, line["grad_allreduce"]))


def test_two_rank_ddp_training_step_over_gloo():
    """harness.train on 2 ranks (gloo, CPU, GA ops through the oracle): each rank draws its own sample, DDP averages the
    gradients, so both ranks hold identical parameters after two Adam steps although their losses differ."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, loss0, dig0, first0, how), (_, loss1, dig1, first1, _) = res
    assert how == "DistributedDataParallel (gloo)"
    assert dig0 == dig1 and first0 == first1, "ranks must hold identical parameters after DDP steps"
    assert loss0 != loss1, "each rank trains on its own sample"
