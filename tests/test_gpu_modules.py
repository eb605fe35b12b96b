"""GPU tests of the drop-in operator API (ganet_amd.functions / ganet_amd.modules / the pybind module GANet,
mirroring libs/GANet/{functions,modules}/GANet.py and the pybind module of GANet_cuda.cpp)."""
import numpy as np
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("wg", [1, 0])
def test_lga2_module_on_workgroup_and_one_wave_rings(torch_mod, port_oracle, wg):
    """LGA2 through autograd with the workgroup rings (default) and with the one-wave rings they fall back to"""
    torch = torch_mod
    import torch.nn.functional as F
    import ganet_amd.modules.GANet as M
    from ganet_amd import _native
    _native.lib().set_option("GANET_LGA_WAVE", 1 + wg)
    try:
        torch.manual_seed(11)
        x = torch.randn((2, 41, 21, 40), device="cuda", requires_grad=True)
        f = F.normalize(torch.randn((2, 75, 21, 40), device="cuda"), p=1, dim=1).requires_grad_()
        gy = torch.randn_like(x)
        y = M.LGA2(radius=2)(x, f)
        y.backward(gy)
        torch.cuda.synchronize()
    finally:
        _native.lib().set_option("GANET_LGA_WAVE", 2)
    o_y, ins = port_oracle.lga_chain_forward(_np(x), _np(f), 2, 2)
    o_gx, o_gf = port_oracle.lga_chain_backward(ins, _np(f), _np(gy), 2)
    assert np.abs(_np(y) - o_y).max() <= pc.TOL
    assert np.abs(_np(x.grad) - o_gx).max() <= pc.TOL
    assert np.abs(_np(f.grad) - o_gf).max() <= pc.TOL


@pytest.mark.parametrize("save_mode", ["", "recompute"])
def test_sga_module_autograd(torch_mod, port_oracle, monkeypatch, save_mode):
    torch = torch_mod
    import torch.nn.functional as F
    from ganet_amd.modules.GANet import SGA
    monkeypatch.setenv("GANET_SGA_SAVE", save_mode)
    torch.manual_seed(123)
    x = torch.randn(2, 3, 33, 10, 24, device="cuda", requires_grad=True)
    gs = [F.normalize(torch.randn(2, 3, 5, 10, 24, device="cuda"), p=1, dim=2).requires_grad_() for _ in range(4)]
    go = torch.randn_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # a non-default torch stream
        out = SGA()(x, *gs)
        out.backward(go)
    side.synchronize()
    o_out, o_tmp, o_mask = port_oracle.sga_forward(_np(x), *[_np(g) for g in gs])
    o_g = port_oracle.sga_backward(_np(x), *[_np(g) for g in gs], o_tmp, o_mask, _np(go))
    assert np.array_equal(_np(out), o_out)
    for t, want in zip([x] + gs, o_g):
        assert np.abs(_np(t.grad) - want).max() <= pc.TOL


@pytest.mark.parametrize("cls_name,passes,five_d,paired", [("LGA", 1, False, "1"), ("LGA2", 2, False, "1"), ("LGA2", 2, False, "0"),
                                                           ("LGA3", 3, False, "1"), ("LGA3D", 1, True, "1"), ("LGA3D2", 2, True, "1"),
                                                           ("LGA3D2", 2, True, "0"), ("LGA3D3", 3, True, "1")])
def test_lga_modules_autograd(torch_mod, port_oracle, monkeypatch, cls_name, passes, five_d, paired):
    """every LGA module through autograd; the two-pass forms both with the pair-interleaved private intermediate (default) and
    with GANET_LGA_PAIRED=0 (intermediate in the API layout)"""
    monkeypatch.setenv("GANET_LGA_PAIRED", paired)
    torch = torch_mod
    import torch.nn.functional as F
    import ganet_amd.modules.GANet as M
    torch.manual_seed(7)
    shape = (2, 3, 9, 12, 40) if five_d else (2, 9, 12, 40)
    fshape = shape[:-3] + (75,) + shape[-2:]
    x = torch.randn(shape, device="cuda", requires_grad=True)
    f = F.normalize(torch.randn(fshape, device="cuda"), p=1, dim=len(shape) - 3).requires_grad_()
    gy = torch.randn_like(x)
    gy_keep = gy.clone()
    y = getattr(M, cls_name)(radius=2)(x, f)
    y.backward(gy)
    torch.cuda.synchronize()
    assert torch.equal(gy, gy_keep), "gradOutput must not be modified (SURVEY F7)"
    o_y, ins = port_oracle.lga_chain_forward(_np(x), _np(f), 2, passes)
    o_gx, o_gf = port_oracle.lga_chain_backward(ins, _np(f), _np(gy), 2)
    assert np.abs(_np(y) - o_y).max() <= pc.TOL
    assert np.abs(_np(x.grad) - o_gx).max() <= pc.TOL
    assert np.abs(_np(f.grad) - o_gf).max() <= pc.TOL


def test_cost_volume_and_regression_modules(torch_mod):
    """Against a torch restatement of modules/GANet.py:119-148 (slice-assign loop / sum(x*disp))."""
    torch = torch_mod
    from ganet_amd.modules.GANet import DisparityRegression, GetCostVolume
    torch.manual_seed(1)
    maxdisp = 12
    x = torch.randn(2, 8, 10, 40, device="cuda", requires_grad=True)
    y = torch.randn(2, 8, 10, 40, device="cuda", requires_grad=True)
    cost = GetCostVolume(maxdisp)(x, y)
    xr, yr = x.detach().clone().requires_grad_(), y.detach().clone().requires_grad_()
    ref = x.new_zeros(2, 16, maxdisp + 1, 10, 40)
    for i in range(maxdisp + 1):
        if i > 0:
            ref[:, :8, i, :, i:] = xr[:, :, :, i:]
            ref[:, 8:, i, :, i:] = yr[:, :, :, :-i]
        else:
            ref[:, :8, i] = xr
            ref[:, 8:, i] = yr
    assert torch.equal(cost, ref)
    g = torch.randn_like(cost)
    cost.backward(g)
    ref.backward(g)
    assert torch.allclose(x.grad, xr.grad, atol=1e-5) and torch.allclose(y.grad, yr.grad, atol=1e-5)
    p = torch.softmax(torch.randn(2, maxdisp + 1, 10, 40, device="cuda"), 1).requires_grad_()
    out = DisparityRegression(maxdisp)(p)
    disp = torch.arange(maxdisp + 1, device="cuda", dtype=torch.float32).view(1, -1, 1, 1)
    pr = p.detach().clone().requires_grad_()
    want = torch.sum(pr * disp, 1)
    assert torch.allclose(out, want, atol=1e-5)
    go = torch.randn_like(out)
    out.backward(go)
    want.backward(go)
    assert torch.allclose(p.grad, pr.grad, atol=1e-6)


def test_ext_module_reference_buffer_contract(torch_mod, port_oracle):
    """The six functions of the reference's pybind module, called exactly as the reference's
    functions/GANet.py calls them (zero-filled caller buffers, aliasing in the LGA2 backward)."""
    torch = torch_mod
    import torch.nn.functional as F
    from libs.GANet.build.lib import GANet          # the pybind module (ganet_amd/csrc/ganet_torch_ext.cpp), where the reference imports it from
    torch.manual_seed(3)
    x = torch.randn(1, 2, 17, 6, 12, device="cuda")
    gs = [F.normalize(torch.randn(1, 2, 5, 6, 12, device="cuda"), p=1, dim=2) for _ in range(4)]
    output, temp_out, mask = (torch.zeros_like(x) for _ in range(3))
    assert GANet.sga_cuda_forward(x, *gs, temp_out, output, mask) == 1
    go = torch.randn_like(x)
    gradInput = torch.zeros_like(x)
    grads = [torch.zeros_like(g) for g in gs]
    temp_grad = torch.zeros_like(x)
    max_idx = torch.zeros(1, 2, 6, 12, device="cuda")
    saved_tmp = temp_out.clone()
    GANet.sga_cuda_backward(x, *gs, temp_out, mask, max_idx, go, temp_grad, gradInput, *grads)
    torch.cuda.synchronize()
    o_out, o_tmp, o_mask = port_oracle.sga_forward(_np(x), *[_np(g) for g in gs])
    o_g = port_oracle.sga_backward(_np(x), *[_np(g) for g in gs], o_tmp, o_mask, _np(go))
    assert np.array_equal(_np(output), o_out) and np.array_equal(_np(mask), o_mask)
    assert np.array_equal(_np(saved_tmp), o_tmp)
    for t, want in zip([gradInput] + grads, o_g):
        assert np.abs(_np(t) - want).max() <= pc.TOL
    # LGA2 exactly as Lga2Function.backward chains it (functions/GANet.py:189-203)
    xl = torch.randn(1, 6, 8, 36, device="cuda")
    f = F.normalize(torch.randn(1, 75, 8, 36, device="cuda"), p=1, dim=1)
    t1, y = torch.zeros_like(xl), torch.zeros_like(xl)
    GANet.lga_cuda_forward(xl, f, t1, 2)
    GANet.lga_cuda_forward(t1, f, y, 2)
    gy = torch.randn_like(xl)
    gy0 = gy.clone()
    gradFilters = torch.zeros_like(f)
    o_y, ins = port_oracle.lga_chain_forward(_np(xl), _np(f), 2, 2)
    o_gx, o_gf = port_oracle.lga_chain_backward(ins, _np(f), _np(gy0), 2)
    GANet.lga_cuda_backward(t1, f, gy, t1, gradFilters, 2)        # gradInput aliases input
    GANet.lga_cuda_backward(xl, f, t1, gy, gradFilters, 2)        # result lands in gradOutput's buffer
    torch.cuda.synchronize()
    assert np.abs(_np(y) - o_y).max() <= pc.TOL
    assert np.abs(_np(gy) - o_gx).max() <= pc.TOL
    assert np.abs(_np(gradFilters) - o_gf).max() <= pc.TOL


def test_cpu_tensors_fail_loudly(torch_mod):
    torch = torch_mod
    from ganet_amd.modules.GANet import LGA2, SGA
    x = torch.randn(1, 1, 3, 2, 4)
    g = torch.randn(1, 1, 5, 2, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        SGA()(x, g, g, g, g)
    with pytest.raises(RuntimeError, match="no CPU path"):
        LGA2(2)(torch.randn(1, 3, 4, 4), torch.randn(1, 75, 4, 4))


@pytest.mark.parametrize("offset", [4, 12, 20])
def test_sga_on_16_byte_aligned_views(torch_mod, port_oracle, offset):
    """Inputs that are only 16-byte aligned (views at an element offset into larger buffers): the row scans count
    their batches from 128-byte line boundaries of the ADDRESS, so every misalignment is a different batch grid."""
    torch = torch_mod
    import torch.nn.functional as F
    from ganet_amd.functions.GANet import SgaFunction
    torch.manual_seed(offset)
    shape, gshape = (1, 2, 33, 6, 40), (1, 2, 5, 6, 40)

    def view_of(t):
        buf = torch.empty(t.numel() + 64, device="cuda")
        v = buf[offset:offset + t.numel()].view(t.shape)
        v.copy_(t)
        return v

    x = view_of(torch.randn(shape, device="cuda")).requires_grad_()
    gs = [view_of(F.normalize(torch.randn(gshape, device="cuda"), p=1, dim=2)).requires_grad_() for _ in range(4)]
    go = view_of(torch.randn(shape, device="cuda"))
    assert x.data_ptr() % 128 != 0 and x.data_ptr() % 16 == 0
    out = SgaFunction.apply(x, *gs)
    grads = torch.autograd.grad(out, [x] + gs, go)
    torch.cuda.synchronize()
    o_out, o_tmp, o_mask = port_oracle.sga_forward(_np(x), *[_np(g) for g in gs])
    o_g = port_oracle.sga_backward(_np(x), *[_np(g) for g in gs], o_tmp, o_mask, _np(go))
    assert np.array_equal(_np(out), o_out)
    for got, want in zip(grads, o_g):
        assert np.abs(_np(got) - want).max() <= pc.TOL


def test_lga2_backward_with_a_4_byte_aligned_gradient(torch_mod, port_oracle):
    """Lga2Function's forward commits to the pair-interleaved private intermediate; an incoming gradient that is
    contiguous but only 4-byte aligned (a slice at an odd element offset of a larger buffer) must still reach the kernels
    (they stage it with 16-byte copies): the backward copies it instead of raising."""
    torch = torch_mod
    import torch.nn.functional as F
    from ganet_amd.functions.GANet import Lga2Function
    torch.manual_seed(5)
    shape = (1, 9, 20, 40)
    x = torch.randn(shape, device="cuda", requires_grad=True)
    f = F.normalize(torch.randn(1, 75, 20, 40, device="cuda"), p=1, dim=1).requires_grad_()
    buf = torch.empty(x.numel() + 8, device="cuda")
    gy = buf[1:1 + x.numel()].view(shape)
    gy.copy_(torch.randn(shape, device="cuda"))
    assert gy.is_contiguous() and gy.data_ptr() % 16 != 0
    y = Lga2Function.apply(x, f, 2)
    gx, gf = torch.autograd.grad(y, [x, f], gy)
    torch.cuda.synchronize()
    o_y, ins = port_oracle.lga_chain_forward(_np(x), _np(f), 2, 2)
    o_gx, o_gf = port_oracle.lga_chain_backward(ins, _np(f), _np(gy), 2)
    assert np.abs(_np(y) - o_y).max() <= pc.TOL
    assert np.abs(_np(gx) - o_gx).max() <= pc.TOL and np.abs(_np(gf) - o_gf).max() <= pc.TOL


@pytest.mark.parametrize("shape", [(1, 3, 33, 10, 24), (1, 2, 9, 5, 7), (1, 1, 240, 4, 12)])
def test_sga_without_grad_takes_the_inference_path(torch_mod, port_oracle, shape):
    """Under torch.no_grad() (predict.py:113) or when no input needs a gradient, SgaFunction keeps nothing for a
    backward and runs the four-launch running-maximum form; its output is the same bit-exact volume.  Shapes: the
    fused scans' range, W % 4 != 0 and D > 208 (both need the scratch volumes)."""
    torch = torch_mod
    import torch.nn.functional as F
    from ganet_amd.functions.GANet import SgaFunction
    torch.manual_seed(sum(shape))
    N, C, D, H, W = shape
    x = torch.randn(shape, device="cuda")
    gs = [F.normalize(torch.randn(N, C, 5, H, W, device="cuda"), p=1, dim=2) for _ in range(4)]
    want, _, _ = port_oracle.sga_forward(_np(x), *[_np(g) for g in gs])
    out_plain = SgaFunction.apply(x, *gs)                    # nothing requires grad
    assert out_plain.grad_fn is None
    with torch.no_grad():
        out_ng = SgaFunction.apply(x.clone().requires_grad_(), *gs)
    torch.cuda.synchronize()
    assert np.array_equal(_np(out_plain), want) and np.array_equal(_np(out_ng), want)
    before = torch.cuda.memory_allocated()
    out = SgaFunction.apply(x, *gs)
    torch.cuda.synchronize()
    if W % 4 == 0 and D <= 208:     # no directional volumes / mask / arg-max were kept or allocated
        assert torch.cuda.memory_allocated() - before <= x.numel() * 4 + 4096
    del out


def test_loss_functions_on_gpu_match_cpu(torch_mod):
    """MyLoss2 / MyLoss (functions/GANet.py:264-310) are plain tensor statements applied in the reference's order
    (its sequential in-place updates are the specification): GPU == CPU, and a target that asks for a gradient gets zeros (the reference hands back a one-element
    zero tensor there)."""
    torch = torch_mod
    from ganet_amd.modules.GANet import MyLoss, MyLoss2
    torch.manual_seed(0)
    a = (torch.randn(4000, device="cuda") * 4).requires_grad_()
    b = torch.randn(4000, device="cuda") * 4
    loss = MyLoss2(thresh=3, alpha=2)(a, b)
    loss.backward()
    ac = a.detach().cpu().requires_grad_()
    lc = MyLoss2(thresh=3, alpha=2)(ac, b.cpu())
    lc.backward()
    assert abs(loss.item() - lc.item()) <= 1e-5 * max(1.0, abs(lc.item()))
    assert (a.grad.cpu() - ac.grad).abs().max().item() <= 1e-7
    a2, b2 = a.detach().clone().requires_grad_(), b.clone().requires_grad_()
    MyLoss()(a2, b2).backward()
    assert b2.grad is not None and float(b2.grad.abs().max()) == 0.0


def test_concurrent_python_threads_on_one_device(torch_mod, port_oracle):
    """SURVEY 8b, threading: under nn.DataParallel (the reference's train.py:73) one Python thread per replica calls the
    ops concurrently and autograd's engine threads call the backwards.  Here: four threads on ONE device, each on its own
    stream with its own inputs, several rounds of SGA + LGA2 forward / backward; every result must equal the oracle's
    (forward bit-exact) -- no shared scratch, no shared error state, options read-only."""
    import threading
    torch = torch_mod
    import torch.nn.functional as F
    from ganet_amd.modules.GANet import SGA, LGA2
    nthreads, rounds = 4, 3
    cases = []
    for t in range(nthreads):
        g = torch.Generator(device="cpu").manual_seed(100 + t)
        x = torch.randn(1, 2, 17 + 8 * t, 9, 24, generator=g)
        gs = [F.normalize(torch.randn(1, 2, 5, 9, 24, generator=g), p=1, dim=2) for _ in range(4)]
        go = torch.randn(x.shape, generator=g)
        lx = torch.randn(1, 11 + 2 * t, 7, 36, generator=g)
        lf = F.normalize(torch.randn(1, 75, 7, 36, generator=g), p=1, dim=1)
        lgy = torch.randn(lx.shape, generator=g)
        cases.append((x, gs, go, lx, lf, lgy))
    results, errors = [None] * nthreads, []
    start = threading.Barrier(nthreads)

    def work(t):
        try:
            x, gs, go, lx, lf, lgy = cases[t]
            stream = torch.cuda.Stream()
            out = []
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(rounds):
                    dx = x.cuda().requires_grad_()
                    dgs = [g.cuda().requires_grad_() for g in gs]
                    o = SGA()(dx, *dgs)
                    o.backward(go.cuda())
                    dl = lx.cuda().requires_grad_()
                    df = lf.cuda().requires_grad_()
                    y = LGA2(radius=2)(dl, df)
                    y.backward(lgy.cuda())
                    stream.synchronize()
                    out.append((_np(o), [_np(v.grad) for v in [dx] + dgs], _np(y), _np(dl.grad), _np(df.grad)))
            results[t] = out
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(nthreads):
        x, gs, go, lx, lf, lgy = cases[t]
        o_out, o_tmp, o_mask = port_oracle.sga_forward(x.numpy(), *[g.numpy() for g in gs])
        o_g = port_oracle.sga_backward(x.numpy(), *[g.numpy() for g in gs], o_tmp, o_mask, go.numpy())
        o_y, ins = port_oracle.lga_chain_forward(lx.numpy(), lf.numpy(), 2, 2)
        o_gx, o_gf = port_oracle.lga_chain_backward(ins, lf.numpy(), lgy.numpy(), 2)
        for out, grads, y, gx, gf in results[t]:
            assert np.array_equal(out, o_out), t
            for got, want in zip(grads, o_g):
                assert np.abs(got - want).max() <= pc.TOL, t
            assert np.abs(y - o_y).max() <= pc.TOL and np.abs(gx - o_gx).max() <= pc.TOL and np.abs(gf - o_gf).max() <= pc.TOL, t
