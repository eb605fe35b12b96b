"""The C restatement (oracle/ganet_oracle.c) against the committed golden vectors,
which were produced by the reference's own kernel bodies (tests/golden/make_golden.py).
Bit-exact everywhere: the restatement uses the same fma order as the reference."""
import numpy as np
import pytest

from golden_util import assert_bit_equal, lga_case_names, load, sga_case_names


@pytest.mark.parametrize("name", sga_case_names())
def test_sga_forward_and_backward_match_golden(port_oracle, name):
    z = load("sga_golden.npz")
    x, go = z[f"{name}.x"], z[f"{name}.go"]
    gs = [z[f"{name}.g{d}"] for d in range(4)]
    out, tmp, mask = port_oracle.sga_forward(x, *gs)
    assert_bit_equal(out, z[f"{name}.out"], "out")
    assert_bit_equal(tmp, z[f"{name}.tmp"], "temp_out")
    assert np.array_equal(mask.astype(np.uint8), z[f"{name}.mask"])
    for d in range(4):
        assert_bit_equal(port_oracle.sga_scan(x, gs[d], d), z[f"{name}.A{d}"], f"A{d}")
    grads = port_oracle.sga_backward(x, *gs, tmp, mask, go)
    assert_bit_equal(grads[0], z[f"{name}.gx"], "gradInput")
    for d in range(4):
        assert_bit_equal(grads[1 + d], z[f"{name}.gw{d}"], f"grad{d}")


def test_sga_cfg1_forward_matches_golden(port_oracle):
    """BASELINE.json configs[0]: SGA forward on one 1x48x48x48 volume (C=1), CPU."""
    z = load("sga_cfg1_golden.npz")
    out, _, mask = port_oracle.sga_forward(z["x"], z["g0"], z["g1"], z["g2"], z["g3"])
    assert_bit_equal(out, z["out"], "out")
    assert np.array_equal(mask.astype(np.uint8), z["mask"])


def test_sga_ties_case_really_has_ties():
    z = load("sga_golden.npz")
    A = [z[f"ties.A{d}"] for d in range(4)]
    eq_dirs = sum(int((A[i] == A[j]).sum()) for i in range(4) for j in range(i + 1, 4))
    assert eq_dirs > 20, "direction-merge ties expected in the 'ties' fixture"
    a = A[0]
    mx = a.max(axis=2, keepdims=True)
    assert int(((a == mx).sum(axis=2) > 1).sum()) > 5, "argmax ties expected"


@pytest.mark.parametrize("name", lga_case_names())
def test_lga_chain_matches_golden(port_oracle, name):
    z = load("lga_golden.npz")
    r, passes = (int(v) for v in z[f"{name}.meta"])
    y, ins = port_oracle.lga_chain_forward(z[f"{name}.x"], z[f"{name}.f"], r, passes)
    assert_bit_equal(y, z[f"{name}.y"], "y")
    gx, gf = port_oracle.lga_chain_backward(ins, z[f"{name}.f"], z[f"{name}.gy"], r)
    assert_bit_equal(gx, z[f"{name}.gx"], "gx")
    assert_bit_equal(gf, z[f"{name}.gf"], "gf")


def test_cost_volume_and_regression_match_torch_restatement(port_oracle):
    """GetCostVolume / DisparityRegression are plain torch in the reference
    (libs/GANet/modules/GANet.py:119-148); restate them with torch slicing here."""
    import torch
    torch.manual_seed(0)
    x, y = torch.randn(2, 3, 4, 9), torch.randn(2, 3, 4, 9)
    maxdisp = 5
    cost = torch.zeros(2, 6, maxdisp + 1, 4, 9)
    for i in range(maxdisp + 1):
        if i > 0:
            cost[:, :3, i, :, i:] = x[:, :, :, i:]
            cost[:, 3:, i, :, i:] = y[:, :, :, :-i]
        else:
            cost[:, :3, i] = x
            cost[:, 3:, i] = y
    assert np.array_equal(port_oracle.cost_volume(x.numpy(), y.numpy(), maxdisp), cost.numpy())
    p = torch.softmax(torch.randn(2, maxdisp + 1, 4, 9), 1)
    disp = torch.arange(maxdisp + 1, dtype=torch.float32).view(1, -1, 1, 1)
    ref = torch.sum(p * disp, 1).numpy()
    np.testing.assert_allclose(port_oracle.disparity_regression(p.numpy(), maxdisp), ref, atol=1e-6)


# ---- the FULL model shapes, by digest (tests/golden/digests.json <- make_golden.py --digests <- oracle/_ref) ----------------
import golden_util as gu  # noqa: E402


@pytest.mark.parametrize("name,shape,seed", gu.SGA_DIGEST_CASES)
def test_sga_full_size_restatement_matches_reference_digests(port_oracle, name, shape, seed):
    """BASELINE configs[1]'s SGA volume [1,32,65,80,208], the 1/6-resolution volume and cfg3's: every array of SgaFunction's
    forward and backward as the C restatement computes it has the sha256 the REFERENCE's kernel bodies gave on the same seeded
    inputs -- the full-shape link between restatement and reference that does not need /root/reference at test time."""
    want = gu.load_digests()[name]
    assert tuple(want["shape"]) == shape and want["seed"] == seed
    got = gu.sga_digests(port_oracle, shape, seed)
    assert {k: v for k, v in got.items() if k.startswith("in.")} == {k: v for k, v in want["sha256"].items() if k.startswith("in.")}, \
        "the seeded inputs themselves differ (numpy's generator?)"
    assert got == want["sha256"], [k for k in got if got[k] != want["sha256"][k]]


@pytest.mark.parametrize("name,shape,seed", gu.LGA_DIGEST_CASES)
def test_lga2_full_size_restatement_matches_reference_digests(port_oracle, name, shape, seed):
    """Lga2Function (radius 2) at [1,193,240,624] (configs[1]) and [1,193,384,1248] (cfg3): intermediate, output and both gradients."""
    want = gu.load_digests()[name]
    got = gu.lga_digests(port_oracle, shape, seed)
    assert got == want["sha256"], [k for k in got if got[k] != want["sha256"][k]]
