"""GPU parity of the fused caller-side op chains: ganet_amd.modules.fused vs the reference's own
torch statements on the CPU (oracle/fused_ref.py; LGA/SGA through the C oracle)."""
import numpy as np
import pytest

import parity_cases as pc
from oracle import fused_ref as fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("shape", [(1, 4, 33, 12, 24), (2, 3, 17, 9, 16)])
def test_guided_sga_matches_sgablock_lines(torch_mod, port_oracle, shape):
    """GuidedSGA(x, g_raw) == split/view/normalize x4 + SGA (models/GANet_deep.py:263-269), forward and backward."""
    torch = torch_mod
    from ganet_amd.modules.fused import GuidedSGA
    N, C, D, H, W = shape
    torch.manual_seed(5)
    x = torch.randn(shape, device="cuda", requires_grad=True)
    g = torch.randn(N, 20 * C, H, W, device="cuda", requires_grad=True)
    go = torch.randn(shape, device="cuda")
    out = GuidedSGA()(x, g)
    out.backward(go)
    torch.cuda.synchronize()
    # CPU: the reference's statements for the guidance, the C oracle for SGA, autograd for the normalisation
    gc = g.detach().cpu().requires_grad_()
    ks = fr.sgablock_guidance(gc, C)
    kn = [np.ascontiguousarray(_np(k)) for k in ks]
    o_out, o_tmp, o_mask = port_oracle.sga_forward(_np(x), *kn)
    o_g = port_oracle.sga_backward(_np(x), *kn, o_tmp, o_mask, _np(go))
    torch.autograd.backward(ks, [torch.from_numpy(np.ascontiguousarray(a)) for a in o_g[1:]])
    # the kernel's normalised taps can differ from torch-CPU's in the last bit (order of the 5-term sum),
    # so the forward is compared at the op tolerance, not bit-exactly
    assert np.abs(_np(out) - o_out).max() <= pc.TOL
    assert np.abs(_np(x.grad) - o_g[0]).max() <= pc.TOL
    assert np.abs(_np(g.grad) - gc.grad.numpy()).max() <= pc.TOL


def test_normalize_functions_match_torch(torch_mod):
    torch = torch_mod
    import torch.nn.functional as F
    from ganet_amd.functions.fused import normalize_filters, normalize_guidance
    torch.manual_seed(9)
    g = torch.randn(2, 75, 20, 24, device="cuda", requires_grad=True)
    gy = torch.randn(2, 75, 20, 24, device="cuda")
    y = normalize_filters(g)
    y.backward(gy)
    gc = g.detach().cpu().requires_grad_()
    w = F.normalize(gc, p=1, dim=1)
    w.backward(gy.cpu())
    np.testing.assert_allclose(_np(y), _np(w), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(_np(g.grad), gc.grad.numpy(), rtol=1e-4, atol=1e-6)
    # only two of the four guidance outputs receive a gradient: the others count as zero
    g2 = torch.randn(1, 40, 6, 8, device="cuda", requires_grad=True)
    ks = normalize_guidance(g2, 2)
    (ks[0].sum() + 2 * ks[3].sum()).backward()
    g2c = g2.detach().cpu().requires_grad_()
    kc = fr.sgablock_guidance(g2c, 2)
    (kc[0].sum() + 2 * kc[3].sum()).backward()
    np.testing.assert_allclose(_np(g2.grad), g2c.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("maxdisp,H,W", [(11, 14, 36), (192, 24, 624)])
def test_dispagg_tail_matches_reference_statements(torch_mod, port_oracle, maxdisp, H, W):
    """DispAggTail == models/GANet_deep.py:243-247, forward and all three gradients: a small volume (maxdisp 11) and a
    full-width strip at the models' maxdisp 192 ([1,193,24,624])."""
    torch = torch_mod
    from ganet_amd.modules.fused import DispAggTail
    torch.manual_seed(3)
    x = torch.randn(1, maxdisp + 1, H, W, device="cuda", requires_grad=True)
    lg1 = torch.randn(1, 75, H, W, device="cuda", requires_grad=True)
    lg2 = torch.randn(1, 75, H, W, device="cuda", requires_grad=True)
    go = torch.randn(1, H, W, device="cuda")
    out = DispAggTail(maxdisp)(x, lg1, lg2)
    out.backward(go)
    torch.cuda.synchronize()
    xc, l1c, l2c = (t.detach().cpu().requires_grad_() for t in (x, lg1, lg2))
    parts = {}
    want = fr.dispagg_tail(xc, l1c, l2c, maxdisp, port_oracle, parts=parts)
    want.backward(go.cpu())
    # The tail ends in out = S_dy / S_abs with S_dy = sum_d d * y[d], S_abs = sum_d |y[d]| of the SIGNED second LGA output: where
    # S_abs is tiny the division amplifies fp32 rounding.  So the two sums are held to north_star's 1e-4 on their own (S_dy
    # carries factors d <= maxdisp: its bar scales with the range), and the quotient to the bound those two errors imply
    # PIXEL BY PIXEL: |d out| <= (e_dy + |out| e_abs) / S_abs -- a pixel may only miss the tight bar if its S_abs is small.
    from ganet_amd import _native
    from ganet_amd.functions.GANet import LgaFunction
    from ganet_amd.functions.fused import SoftminFunction, normalize_filters
    from ganet_amd.modules.fused import NormalizedLGA2
    with torch.no_grad():
        x2 = LgaFunction.apply(SoftminFunction.apply(NormalizedLGA2(2)(x, lg1).contiguous()), normalize_filters(lg2), 2)
        s_abs, s_dy = torch.empty(1, H, W, device="cuda"), torch.empty(1, H, W, device="cuda")
        _native.lib().call("ganet_lga_forward_regress", x2.data_ptr(), normalize_filters(lg2).data_ptr(), None, s_abs.data_ptr(),
                           s_dy.data_ptr(), 1, maxdisp + 1, H, W, 2, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    y2 = parts["y2"].numpy().astype(np.float64)
    ref_abs = np.abs(y2).sum(1)
    ref_dy = (y2 * np.arange(maxdisp + 1, dtype=np.float64)[None, :, None, None]).sum(1)
    e_abs, e_dy = 1e-4, 1e-4 * max(1.0, maxdisp / 10.0)
    assert np.abs(_np(s_abs) - ref_abs).max() <= e_abs, np.abs(_np(s_abs) - ref_abs).max()
    assert np.abs(_np(s_dy) - ref_dy).max() <= e_dy, np.abs(_np(s_dy) - ref_dy).max()
    got, ref = _np(out).astype(np.float64), want.detach().numpy().astype(np.float64)
    bound = (e_dy + np.abs(ref) * e_abs) / np.maximum(ref_abs, 1e-12) + 2e-5 * np.abs(ref) + 1e-6
    assert (np.abs(got - ref) <= bound).all(), float((np.abs(got - ref) / bound).max())
    tight = np.abs(got - ref) <= e_dy + 2e-5 * np.abs(ref)
    assert tight.mean() >= 0.999, tight.mean()
    # (the pixels outside the tight bar are small-norm pixels: S_abs below one, i.e. the bound above is what lets them pass)
    assert (ref_abs[~tight] < 1.0).all(), ref_abs[~tight]
    # five chained ops (two of them divisions by small L1 norms) amplify fp32 rounding: the gradients reach
    # |g| ~ 10^1..10^2 here, so the bar is 1e-4 RELATIVE to the largest gradient entry (>= 1e-4 absolute)
    for got, ref in ((x.grad, xc.grad), (lg1.grad, l1c.grad), (lg2.grad, l2c.grad)):
        scale = max(1.0, float(np.abs(ref.numpy()).max()))
        assert np.abs(_np(got) - ref.numpy()).max() <= pc.TOL * scale, (np.abs(_np(got) - ref.numpy()).max(), scale)


def test_norm_regression_full_size(torch_mod):
    """[1,193,240,624] (cfg2): against torch's own ops on the GPU (same statements as the reference)."""
    torch = torch_mod
    import torch.nn.functional as F
    from ganet_amd.modules.fused import NormDisparityRegression
    torch.manual_seed(1)
    x = torch.softmax(-torch.randn(1, 193, 240, 624, device="cuda"), dim=1).requires_grad_()
    go = torch.randn(1, 240, 624, device="cuda")
    out = NormDisparityRegression(192)(x)
    out.backward(go)
    x2 = x.detach().clone().requires_grad_()
    disp = torch.arange(193, device="cuda", dtype=torch.float32).view(1, 193, 1, 1)
    ref = torch.sum(F.normalize(x2, p=1, dim=1) * disp, 1)
    ref.backward(go)
    np.testing.assert_allclose(_np(out), _np(ref), rtol=1e-5, atol=1e-4)
    assert (x.grad - x2.grad).abs().max().item() <= 1e-3 * x2.grad.abs().max().item()


def test_guided_sga_bn_relu_eval_matches_op_chain(torch_mod, port_oracle):
    """GuidedSGABnRelu in eval mode / no_grad (BN affine + ReLU inside the merge kernel) == the op chain
    normalise -> SGA -> BatchNorm3d(eval) -> ReLU of models/GANet_deep.py:263-271, and the training path still
    differentiates."""
    torch = torch_mod
    from ganet_amd.modules.fused import GuidedSGABnRelu
    torch.manual_seed(11)
    N, C, D, H, W = 1, 4, 33, 10, 24
    bn = torch.nn.BatchNorm3d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    m = GuidedSGABnRelu(bn).eval()
    x = torch.randn(N, C, D, H, W, device="cuda")
    g = torch.randn(N, 20 * C, H, W, device="cuda")
    with torch.no_grad():
        got = m(x, g)
    ks = [np.ascontiguousarray(_np(k)) for k in fr.sgablock_guidance(g.cpu(), C)]
    o_out, _, _ = port_oracle.sga_forward(_np(x), *ks)
    bn_cpu = torch.nn.BatchNorm3d(C).eval()
    bn_cpu.load_state_dict({k: v.cpu() for k, v in bn.state_dict().items()})
    with torch.no_grad():
        want = torch.relu(bn_cpu(torch.from_numpy(o_out)))
    assert np.abs(_np(got) - want.numpy()).max() <= pc.TOL
    # with autograd on, the same module runs the differentiable chain
    x.requires_grad_(); g.requires_grad_()
    y = m(x, g)
    y.sum().backward()
    assert x.grad is not None and g.grad is not None
    assert np.abs(_np(y) - want.numpy()).max() <= pc.TOL


def test_softmin_disparity_regression_full_size(torch_mod):
    """SoftminDisparityRegression on [1,193,240,624] == Softmin(dim=1) + DisparityRegression (models/GANet_deep.py:217-219)
    computed with torch's own ops on the GPU, forward and backward."""
    torch = torch_mod
    from ganet_amd.modules.fused import SoftminDisparityRegression
    torch.manual_seed(2)
    x = (3 * torch.randn(1, 193, 240, 624, device="cuda")).requires_grad_()
    go = torch.randn(1, 240, 624, device="cuda")
    out = SoftminDisparityRegression(192)(x)
    out.backward(go)
    x2 = x.detach().clone().requires_grad_()
    disp = torch.arange(193, device="cuda", dtype=torch.float32).view(1, 193, 1, 1)
    ref = torch.sum(torch.nn.functional.softmin(x2, dim=1) * disp, 1)
    ref.backward(go)
    np.testing.assert_allclose(_np(out), _np(ref), rtol=1e-5, atol=1e-4)
    assert (x.grad - x2.grad).abs().max().item() <= 1e-4 * max(1.0, x2.grad.abs().max().item())


@pytest.mark.parametrize("ishape,osize", [((1, 1, 65, 80, 208), (193, 240, 624)), ((2, 3, 9, 16, 32), (25, 48, 96)),
                                          ((1, 2, 7, 5, 6), (10, 13, 17))])
def test_trilinear_upsample_vs_aten(torch_mod, ishape, osize):
    """TrilinearUpsample (gather backward) == F.interpolate(mode='trilinear', align_corners=False) and its autograd
    adjoint, at the model's Disp / DispAgg shape ([1,1,65,80,208] -> [193,240,624], models/GANet_deep.py:212, 240) and two
    small ones; prints both backward times at the full size."""
    torch = torch_mod
    import torch.nn.functional as F
    from ganet_amd.modules.fused import TrilinearUpsample
    torch.manual_seed(sum(ishape))
    x = torch.randn(ishape, device="cuda", requires_grad=True)
    gy = torch.randn(ishape[:2] + osize, device="cuda")
    y = TrilinearUpsample()(x, osize)
    y.backward(gy)
    x2 = x.detach().clone().requires_grad_()
    y2 = F.interpolate(x2, size=list(osize), mode="trilinear", align_corners=False)
    y2.backward(gy)
    torch.cuda.synchronize()
    assert float((y - y2).abs().max()) <= 1e-5
    scale = max(1.0, float(x2.grad.abs().max()))
    assert float((x.grad - x2.grad).abs().max()) <= 1e-5 * scale, float((x.grad - x2.grad).abs().max())
    if ishape[2] == 65:
        def timed(fn, n=5):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) / n
        t_ours = timed(lambda: torch.autograd.grad(TrilinearUpsample()(x, osize), x, gy))
        t_aten = timed(lambda: torch.autograd.grad(F.interpolate(x2, size=list(osize), mode="trilinear", align_corners=False), x2, gy))
        print(f"trilinear upsample [1,1,65,80,208] -> [193,240,624] fwd+bwd: ours {t_ours:.3f} ms, ATen {t_aten:.3f} ms")


def _bn3d(torch, C, seed):
    bn = torch.nn.BatchNorm3d(C)
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5); bn.bias.copy_(torch.randn(C, generator=gen))
        bn.running_mean.copy_(torch.randn(C, generator=gen)); bn.running_var.copy_(torch.rand(C, generator=gen) * 1.5 + 0.5)
    return bn


# the SGABlock volumes of cfg2 / cfg4 ([1,32,65,80,208], [1,48,33,40,104]), cfg3 ([1,32,65,128,416]) and cfg5 per GPU
# ([2,32,65,176,320]); a slice size that is not a multiple of four (scalar kernel); an odd one
_TAIL_SHAPES = [(1, 32, 65, 80, 208), (1, 48, 33, 40, 104), (1, 32, 65, 128, 416), (2, 32, 65, 176, 320), (2, 3, 5, 3, 7),
                (1, 4, 9, 10, 12)]


@pytest.mark.parametrize("shape", _TAIL_SHAPES)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_sgablock_residual_tail_matches_reference_statements(torch_mod, shape, mode):
    """ResidualBnRelu(bn)(t, rem) == models/GANet_deep.py:270-277 behind the convolution (`x = bn(t); x += rem; relu(x)`,
    oracle/fused_ref.sgablock_tail: the reference's statements on the CPU), forward and both gradients.
    eval: the BatchNorm folded into one affine -- one rounding away from torch's (x - mean) * invstd * w + b: 1e-5 relative to
    the pre-activation's size, and the gradients exact wherever the pre-activation is not within that rounding of zero.
    train: bn runs in the framework (batch statistics), add + ReLU fused: <= 1e-5 against the CPU's own batch_norm."""
    torch = torch_mod
    from ganet_amd.modules.fused import ResidualBnRelu
    N, C = shape[:2]
    torch.manual_seed(sum(shape))
    t = torch.randn(shape, device="cuda", requires_grad=True)
    rem = torch.randn(shape, device="cuda", requires_grad=True)
    gy = torch.randn(shape, device="cuda")
    bn = _bn3d(torch, C, 4).cuda().train(mode == "train")
    if mode == "eval":
        for p in bn.parameters():
            p.requires_grad_(False)             # frozen statistics AND parameters: the folded one-pass form
    y = ResidualBnRelu(bn, inplace=False)(t, rem)
    y.backward(gy)
    torch.cuda.synchronize()
    bn_cpu = _bn3d(torch, C, 4).train(mode == "train")
    tc, rc = t.detach().cpu().requires_grad_(), rem.detach().cpu().requires_grad_()
    want = fr.sgablock_tail(tc, rc, bn_cpu)
    want.backward(gy.cpu())
    wn = want.detach().numpy()
    tol = 1e-5 * max(1.0, float(np.abs(wn).max()))
    assert np.abs(_np(y) - wn).max() <= tol, np.abs(_np(y) - wn).max()
    with torch.no_grad():
        pre = (bn_cpu(tc.detach()) if mode == "eval" else None)
    if mode == "eval":
        clear = np.abs(pre.numpy() + rc.detach().numpy()) > tol
        assert clear.mean() > 0.999
        assert np.array_equal(_np(rem.grad)[clear], rc.grad.numpy()[clear])
        np.testing.assert_allclose(_np(t.grad)[clear], tc.grad.numpy()[clear], rtol=1e-5, atol=1e-6)
    else:
        # the GPU's batch statistics differ from the CPU's in the last bits, so a handful of pre-activations change sign
        bad = (_np(rem.grad) != rc.grad.numpy())
        assert bad.mean() < 1e-4, bad.mean()
        scale = max(1.0, float(tc.grad.abs().max()))
        close = np.abs(_np(t.grad) - tc.grad.numpy()) <= 1e-4 * scale
        assert close.mean() > 1 - 1e-4, close.mean()
        # running statistics were updated by the framework exactly as without the fusion
        np.testing.assert_allclose(_np(bn.running_mean), bn_cpu.running_mean.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(_np(bn.running_var), bn_cpu.running_var.numpy(), rtol=1e-4, atol=1e-6)


def test_sgablock_residual_tail_in_place_and_no_grad(torch_mod):
    """in place over t (what the harness uses: t is the convolution's output) == out of place; under no_grad nothing is kept;
    an in-place call on a non-leaf with autograd on differentiates through the producer of t."""
    torch = torch_mod
    from ganet_amd.modules.fused import ResidualBnRelu
    torch.manual_seed(8)
    shape = (1, 4, 9, 10, 12)
    bn = _bn3d(torch, 4, 5).cuda().eval()
    for p in bn.parameters():
        p.requires_grad_(False)
    t, rem = torch.randn(shape, device="cuda"), torch.randn(shape, device="cuda")
    with torch.no_grad():
        want = torch.relu(bn(t) + rem)
        t2 = t.clone()
        got = ResidualBnRelu(bn)(t2, rem)
        assert got.data_ptr() == t2.data_ptr()
    assert float((got - want).abs().max()) <= 1e-5
    w = torch.randn(shape, device="cuda", requires_grad=True)
    rem.requires_grad_()
    y = ResidualBnRelu(bn)(w * 2.0, rem)                      # t = a temporary with a producer
    y.sum().backward()
    w2, rem2 = w.detach().clone().requires_grad_(), rem.detach().clone().requires_grad_()
    torch.relu(bn(w2 * 2.0) + rem2).sum().backward()
    assert float((w.grad - w2.grad).abs().max()) <= 1e-5 and float((rem.grad - rem2.grad).abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ResidualBnRelu(bn)(t.cpu(), rem.detach().cpu())        # no CPU path
