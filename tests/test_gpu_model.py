"""GPU tests of the full-model harness: the reference's own models/GANet11.py and models/GANet_deep.py
run on the MI355X through this repository's `libs/` drop-in, compared with the SAME model (same weights, same inputs) on
the CPU with the guided-aggregation ops routed through the C oracle (oracle/cpu_ops.py).

Three arms: (cpu) everything on the CPU, GA ops through the oracle; (gpu) the product: everything on the GPU, GA ops =
libganet_hip.so; (hyb) the GPU model with ONLY the GA ops swapped for the oracle (host round trip per op).
  gpu vs hyb isolates this library: the rest of the model is the same MIOpen / ATen-HIP arithmetic on both sides.  Bars:
      disparities within 1e-2 absolute in eval and 2e-2 in training mode (range 0..48; observed 2e-3 / 5.5e-3),
      gradients cosine >= 0.9995 and median per-tensor rel-L2 <= 2e-2 (observed: GANet11 2.5e-3 / 0.999997, GANet_deep with
      its seven SGA layers 7.3e-3 / 0.99997), both widened to 4x / 3x the gpu-vs-gpu noise floor of the same run when that
      is larger (profiles/r2w_model_and_threads.txt: the same arm twice gave 1 - cosine = 2.9e-4, gpu vs hyb 3.0e-4).  Not tighter: the ops agree with
      the oracle to 2e-7 (LGA) / bit-exactly (SGA forward), but a randomly initialised GANet amplifies that through
      F.normalize(p=1) of a signed LGA output and 48-level regression; MIOpen's weight-gradient kernels use atomics, so
      even gpu vs gpu is not bit-reproducible; and the SGA direction choice / arg-max routing is discontinuous -- a few
      pixels pick another branch after 1e-7 upstream differences.
  gpu vs cpu additionally carries PyTorch's own CPU-vs-MIOpen differences through ~60 layers, which nobody controls to
      1e-4: disparities within 2e-3 of the disparity range, median per-tensor gradient rel-L2 <= 3e-2, cosine >= 0.998
      (observed 0.99957 .. 0.99999).  These are sanity bars on a chaotic system (a broken op gives cosine << 0.99); the
      guarantees are the per-op parity tests."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    sys.path.insert(0, ROOT)
    from harness import refmodel
    if not refmodel.available():
        pytest.skip("no reference model code (GANET_REF_ROOT, /root/reference or oracle/_ref/pyref)")
    return torch


def _arms(torch, name, max_disp, port_oracle):
    from harness import steps
    from oracle.cpu_ops import route_cpu_through_oracle
    torch.manual_seed(7)
    cpu = steps.build_model(name, max_disp, "cpu", hook=lambda m: route_cpu_through_oracle(m, port_oracle))
    gpu = steps.build_model(name, max_disp, "cuda")
    gpu.load_state_dict(cpu.state_dict())
    hyb = steps.build_model(name, max_disp, "cuda", hook=lambda m: route_cpu_through_oracle(m, port_oracle))
    hyb.load_state_dict(cpu.state_dict())
    return {"cpu": cpu, "gpu": gpu, "hyb": hyb}


def _grad_cmp(torch, ga, gb):
    assert set(ga) == set(gb)
    rel = {k: float((ga[k] - gb[k]).norm() / (ga[k].norm() + 1e-12)) for k in ga}
    fa = torch.cat([ga[k].flatten() for k in sorted(ga)]).double()
    fb = torch.cat([gb[k].flatten() for k in sorted(ga)]).double()
    cos = float(torch.dot(fa, fb) / (fa.norm() * fb.norm()))
    return float(np.median(list(rel.values()))), max(rel.items(), key=lambda kv: kv[1]), cos


@pytest.mark.parametrize("name", ["GANet11", "GANet_deep"])
def test_reference_model_on_gpu_matches_cpu_oracle_twin(env, port_oracle, name):
    torch = env
    from ganet_amd import _native
    from harness import steps
    assert not _native.lib().is_simulator
    max_disp, H, W, B = 48, 96, 192, 2
    arms = _arms(torch, name, max_disp, port_oracle)
    left, right, target = steps.synthetic_batch(B, H, W, max_disp, "cpu", seed=5)
    crit = steps.criterion(True)
    dev = {"cpu": "cpu", "gpu": "cuda", "hyb": "cuda"}
    # -- eval / no_grad (predict.py:107-114): SgaFunction takes its inference path here
    d = {tag: steps.predict(m, left.to(dev[tag]), right.to(dev[tag])).cpu() for tag, m in arms.items()}
    # -- train: forward, loss mix of train.py:100-118, backward
    res = {}
    for tag, model in arms.items():
        model.train()
        model.zero_grad()
        outs = model(left.to(dev[tag]), right.to(dev[tag]))
        t = target.to(dev[tag])
        loss = steps.loss_mix(name, outs, t, t < max_disp, crit)
        loss.backward()
        res[tag] = ([o.detach().cpu() for o in outs], float(loss.detach()),
                    {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None})
    # noise floor of the comparison itself: the SAME gpu arm run a second time (MIOpen's weight-gradient kernels accumulate
    # with atomics, so two runs of one model on one input do not give the same gradients)
    gpu = arms["gpu"]
    gpu.zero_grad()
    outs = gpu(left.cuda(), right.cuda())
    t = target.cuda()
    steps.loss_mix(name, outs, t, t < max_disp, crit).backward()
    again = {k: p.grad.detach().cpu() for k, p in gpu.named_parameters() if p.grad is not None}
    n_med, n_worst, n_cos = _grad_cmp(torch, again, res["gpu"][2])
    print(f"{name} gpu vs gpu (same arm twice): grad rel-L2 median {n_med:.3e} worst {n_worst[1]:.3e}  1-cosine {1 - n_cos:.3e}")
    for other, bars in (("hyb", dict(e_eval=1e-2, e_train=2e-2, med=2e-2, cos=0.9995)),
                        ("cpu", dict(e_eval=2e-3 * max_disp, e_train=2e-3 * max_disp, med=3e-2, cos=0.998))):
        e_eval = float((d["gpu"] - d[other]).abs().max())
        e_train = max(float((a - b).abs().max()) for a, b in zip(res["gpu"][0], res[other][0]))
        med, worst, cos = _grad_cmp(torch, res[other][2], res["gpu"][2])
        print(f"{name} gpu vs {other}: eval |dd|max {e_eval:.3e}  train {e_train:.3e}  loss {res['gpu'][1]:.6f} / {res[other][1]:.6f}  "
              f"grad rel-L2 median {med:.3e} worst {worst[1]:.3e} ({worst[0]})  cosine {cos:.8f}")
        assert e_eval <= bars["e_eval"] and e_train <= bars["e_train"], (other, e_eval, e_train)
        assert abs(res["gpu"][1] - res[other][1]) <= 1e-3 * abs(res[other][1])
        # the gradient bars move with the noise floor measured above: on some boxes two runs of the SAME gpu arm already differ
        # by 1 - cosine = 3e-4 (GANet_deep; 3e-5 on others), and the comparison with another arm cannot be better than that
        # ... but the floor itself is capped: a nondeterministic bug in the ops (a race, a miscounted wait) would inflate it and
        # loosen the bars with it.  Observed floors: 1 - cosine <= 3e-4, median rel-L2 <= 9.1e-3 (MIOpen atomics)
        assert 1 - n_cos <= 1.5e-3 and n_med <= 3e-2, ("noise floor of the gpu arm itself", 1 - n_cos, n_med)
        med_bar = max(bars["med"], 3 * n_med)
        cos_bar = min(bars["cos"], 1 - 4 * (1 - n_cos))
        assert med <= med_bar and cos >= cos_bar, (other, med, cos, med_bar, cos_bar)


def test_fused_call_sites_equal_stock_call_forms(env):
    """harness.fuse (ganet_amd.modules.fused in SGABlock / Disp / DispAgg) against the unmodified model: eval forward and
    one training forward+backward on the GPU."""
    torch = env
    from harness import fuse, steps
    torch.manual_seed(3)
    a = steps.build_model("GANet_deep", 48, "cuda")
    b = steps.build_model("GANet_deep", 48, "cuda")
    b.load_state_dict(a.state_dict())
    assert fuse.use_fused_ops(b) == 10
    left, right, target = steps.synthetic_batch(1, 96, 192, 48, "cuda", seed=2)
    da, db = steps.predict(a, left, right), steps.predict(b, left, right)
    assert float((da - db).abs().max()) <= 1e-3
    crit = steps.criterion(True)
    grads = []
    for m in (a, b):
        m.train()
        outs = m(left, right)
        steps.loss_mix("GANet_deep", outs, target, target < 48, crit).backward()
        grads.append(torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).double())
    cos = float(torch.dot(grads[0], grads[1]) / (grads[0].norm() * grads[1].norm()))
    print("fused vs stock: eval max diff", float((da - db).abs().max()), "grad cosine", cos)
    assert cos >= 0.9999


_MODE_B = r'''
import os, shutil, sys, glob, tempfile
import numpy as np
import torch
root = sys.argv[1]
src = os.path.join(root, "oracle", "_ref", "pyref", "modeB")
if os.path.exists("/root/reference/libs/GANet/functions/GANet.py"):
    src_libs = "/root/reference/libs"           # the real files where they exist
else:
    src_libs = os.path.join(src, "libs")
tmp = tempfile.mkdtemp()
dst = os.path.join(tmp, "libs", "GANet")
for sub in ("functions", "modules"):
    os.makedirs(os.path.join(dst, sub))
    for f in glob.glob(os.path.join(src_libs, "GANet", sub, "*.py*")):
        shutil.copy(f, os.path.join(dst, sub))
    open(os.path.join(dst, sub, "__init__.py"), "a").close()
os.makedirs(os.path.join(dst, "build", "lib"))
for d in (os.path.join(tmp, "libs"), dst, os.path.join(dst, "build"), os.path.join(dst, "build", "lib")):
    open(os.path.join(d, "__init__.py"), "a").close()
ext = glob.glob(os.path.join(root, "libs", "GANet", "build", "lib", "GANet*.so"))
assert ext, "pybind module GANet not built"
shutil.copy(ext[0], os.path.join(dst, "build", "lib"))
shutil.copy(os.path.join(root, "ganet_amd", "libganet_hip.so"), os.path.join(dst, "build", "lib"))
sys.path.insert(0, tmp)
sys.path.append(root)                      # for oracle.* only; `libs` is the temp tree
from libs.GANet.functions.GANet import SgaFunction, Lga2Function, Lga3d2Function
F_ = sys.modules["libs.GANet.functions.GANet"]      # (the package attribute of that name is the pybind module: the reference's __init__ star-imports it)
assert os.path.realpath(F_.__file__).startswith(os.path.realpath(tmp)), F_.__file__
assert os.path.realpath(F_.GANet.__file__).startswith(os.path.realpath(tmp)) and F_.GANet.__file__.endswith(".so")
assert "ganet_amd" not in sys.modules, "mode B must not go through the Python layer of ganet_amd"
from libs.GANet.modules.GANet import SGA, LGA2
from oracle.oracle import Oracle
import torch.nn.functional as F
ora = Oracle("port")
torch.manual_seed(11)
x = torch.randn(1, 3, 33, 12, 24, device="cuda", requires_grad=True)
gs = [F.normalize(torch.randn(1, 3, 5, 12, 24, device="cuda"), p=1, dim=2).requires_grad_() for _ in range(4)]
go = torch.randn_like(x)
out = SGA()(x, *gs)
grads = torch.autograd.grad(out, [x] + gs, go)
torch.cuda.synchronize()
n = lambda t: t.detach().cpu().numpy()
o_out, o_tmp, o_mask = ora.sga_forward(n(x), *[n(g) for g in gs])
o_g = ora.sga_backward(n(x), *[n(g) for g in gs], o_tmp, o_mask, n(go))
assert np.array_equal(n(out), o_out), "SGA forward must be bit-exact"
for a, b in zip(grads, o_g):
    assert np.abs(n(a) - b).max() <= 1e-4
xl = torch.randn(2, 9, 20, 40, device="cuda", requires_grad=True)
f = F.normalize(torch.randn(2, 75, 20, 40, device="cuda"), p=1, dim=1).requires_grad_()
gy = torch.randn_like(xl)
y = LGA2(radius=2)(xl, f)
gx, gf = torch.autograd.grad(y, [xl, f], gy.clone())      # the reference's backward overwrites gradOutput in place
torch.cuda.synchronize()
o_y, ins = ora.lga_chain_forward(n(xl), n(f), 2, 2)
o_gx, o_gf = ora.lga_chain_backward(ins, n(f), n(gy), 2)
assert np.abs(n(y) - o_y).max() <= 1e-4 and np.abs(n(gx) - o_gx).max() <= 1e-4 and np.abs(n(gf) - o_gf).max() <= 1e-4
x5 = torch.randn(1, 2, 7, 10, 16, device="cuda", requires_grad=True)
f5 = F.normalize(torch.randn(1, 2, 75, 10, 16, device="cuda"), p=1, dim=2).requires_grad_()
y5 = Lga3d2Function.apply(x5, f5, 2)
o_y5, _ = ora.lga_chain_forward(n(x5), n(f5), 2, 2)
assert np.abs(n(y5) - o_y5).max() <= 1e-4
loaded = [l.split()[-1] for l in open("/proc/self/maps") if "GANet.cpython" in l or "libganet_hip" in l]
print("MODE_B_OK", sorted(set(os.path.basename(p) for p in loaded)))
shutil.rmtree(tmp)
'''


def test_reference_python_layer_unmodified_on_the_pybind_module(env):
    """INTEGRATION.md mode B: the reference's OWN libs/GANet/functions/GANet.py and modules/GANet.py (its files where
    /root/reference exists, their byte-compiled form from oracle/_ref/pyref on the GPU box -- never vendored) import this
    repository's pybind module through their unchanged `from ..build.lib import GANet` and match the oracle: SGA forward
    bit-exact, SGA / LGA2 gradients within 1e-4.  Runs in a fresh interpreter so that `libs` is the reference's package."""
    have_src = os.path.exists("/root/reference/libs/GANet/functions/GANet.py") or \
        os.path.exists(os.path.join(ROOT, "oracle", "_ref", "pyref", "modeB", "libs", "GANet", "functions", "GANet.pyc"))
    if not have_src:
        pytest.skip("the reference's functions/GANet.py is not available in any form")
    env_ = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", _MODE_B, ROOT], capture_output=True, text=True, timeout=900, env=env_, cwd="/tmp")
    print(r.stdout[-1500:], r.stderr[-3000:])
    assert r.returncode == 0 and "MODE_B_OK" in r.stdout
