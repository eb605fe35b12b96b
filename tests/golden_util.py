"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def sga_case_names():
    z = load("sga_golden.npz")
    return sorted({k.split(".")[0] for k in z.files})


def lga_case_names():
    z = load("lga_golden.npz")
    return sorted({k.split(".")[0] for k in z.files})


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = bits(a) != bits(b)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} words differ, max abs {np.abs(a - b).max()}"
