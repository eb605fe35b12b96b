"""Pins the C restatement against the reference's own kernel bodies (oracle/_ref,
built from /root/reference by oracle/Makefile) on fresh random inputs, including
awkward shapes.  Skipped where neither oracle/_ref nor /root/reference exists."""
import numpy as np
import pytest

from golden_util import assert_bit_equal


def _l1(g, axis):
    return (g / np.abs(g).sum(axis, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("shape", [(1, 2, 5, 4, 3), (2, 2, 16, 7, 5), (1, 3, 34, 9, 17), (1, 1, 3, 2, 33)])
def test_sga_port_equals_reference(port_oracle, ref_oracle, shape):
    rng = np.random.default_rng(sum(shape))
    N, C, D, H, W = shape
    x = rng.standard_normal(shape).astype(np.float32)
    gs = [_l1(rng.standard_normal((N, C, 5, H, W)), 2) for _ in range(4)]
    go = rng.standard_normal(shape).astype(np.float32)
    a, b = port_oracle.sga_forward(x, *gs), ref_oracle.sga_forward(x, *gs)
    for u, v, n in zip(a, b, ("out", "temp_out", "mask")):
        assert_bit_equal(u, v, n)
    ga = port_oracle.sga_backward(x, *gs, a[1], a[2], go)
    gb = ref_oracle.sga_backward(x, *gs, b[1], b[2], go)
    for u, v, n in zip(ga, gb, ("gx", "gw0", "gw1", "gw2", "gw3")):
        assert_bit_equal(u, v, n)


@pytest.mark.parametrize("shape,r", [((1, 6, 5, 7), 1), ((2, 9, 6, 11), 2), ((1, 2, 4, 3, 9), 2), ((1, 4, 3, 3), 3)])
def test_lga_port_equals_reference(port_oracle, ref_oracle, shape, r):
    rng = np.random.default_rng(sum(shape) + r)
    fs = list(shape)
    fs[-3] = 3 * (2 * r + 1) ** 2
    x = rng.standard_normal(shape).astype(np.float32)
    f = _l1(rng.standard_normal(fs), -3)
    gy = rng.standard_normal(shape).astype(np.float32)
    assert_bit_equal(port_oracle.lga_forward(x, f, r), ref_oracle.lga_forward(x, f, r), "y")
    ga, gb = port_oracle.lga_backward(x, f, gy, r), ref_oracle.lga_backward(x, f, gy, r)
    assert_bit_equal(ga[0], gb[0], "gx")
    assert_bit_equal(ga[1], gb[1], "gf")
