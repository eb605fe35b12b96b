"""Generates the committed golden vectors from the REFERENCE's own kernel bodies.

Run in the dev container only (needs /root/reference to build oracle/_ref):
    python tests/golden/make_golden.py              the small fixtures (tests/golden/*.npz)
    python tests/golden/make_golden.py --digests    sha256 of every output of the reference at the FULL model shapes
                                                    (BASELINE configs[1] and the SGA-B / cfg3 volumes) -> tests/golden/digests.json
Every output array below comes from oracle/_ref/libganet_ref.so, i.e. from
/root/reference/libs/GANet/src/GANet_kernel.cu compiled through oracle/ref_shim
with the launch order of its host launchers (GANet_kernel.cu:935-1129, 1271-1364)
and the buffer roles of libs/GANet/functions/GANet.py.  Inputs are stored too, so
the fixtures do not depend on any RNG implementation.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import Oracle  # noqa: E402

SGA_CASES = [  # (name, (N,C,D,H,W), kind)
    ("tiny", (1, 2, 4, 3, 5), "randn"),
    ("odd", (2, 3, 7, 6, 9), "randn"),
    ("d33", (1, 2, 33, 8, 12), "randn"),
    ("d65", (1, 1, 65, 4, 6), "randn"),
    ("one", (1, 1, 1, 1, 1), "randn"),
    ("h1", (1, 2, 2, 1, 4), "randn"),
    ("w1", (1, 2, 3, 5, 1), "randn"),
    ("d1", (1, 3, 1, 4, 4), "randn"),
    ("d17w70", (1, 1, 17, 3, 70), "randn"),
    ("ties", (1, 2, 6, 5, 7), "ties"),
]
LGA_CASES = [  # (name, shape, radius, passes)
    ("r1", (1, 5, 4, 6), 1, 1),
    ("r2", (2, 7, 6, 7), 2, 2),
    ("small", (1, 3, 2, 2), 2, 2),
    ("one", (1, 1, 1, 1), 2, 1),
    ("d2", (1, 2, 9, 40), 2, 3),
    ("vol5d", (2, 3, 9, 5, 8), 2, 2),
]


def l1norm(g, axis):
    return (g / np.abs(g).sum(axis, keepdims=True)).astype(np.float32)


def sga_inputs(rng, shape, kind):
    N, C, D, H, W = shape
    if kind == "ties":
        # small integers and dyadic weights: every product and sum is exact, so the
        # in-scan argmax and the direction merge hit many exact ties
        x = rng.integers(-2, 3, shape).astype(np.float32)
        gs = [rng.choice(np.array([0.0, 0.25, 0.5], np.float32), (N, C, 5, H, W)) for _ in range(4)]
        go = rng.integers(-2, 3, shape).astype(np.float32)
    else:
        x = rng.standard_normal(shape).astype(np.float32)
        gs = [l1norm(rng.standard_normal((N, C, 5, H, W)), 2) for _ in range(4)]
        go = rng.standard_normal(shape).astype(np.float32)
    return x, gs, go


def main():
    ref = Oracle("reference")
    rng = np.random.default_rng(20190416)
    out = {}
    for name, shape, kind in SGA_CASES:
        x, gs, go = sga_inputs(rng, shape, kind)
        o, tmp, mask = ref.sga_forward(x, *gs)
        A = [ref.sga_scan(x, gs[d], d) for d in range(4)]
        gx, gw0, gw1, gw2, gw3 = ref.sga_backward(x, *gs, tmp, mask, go)
        out.update({f"{name}.x": x, f"{name}.go": go, f"{name}.out": o, f"{name}.tmp": tmp,
                    f"{name}.mask": mask.astype(np.uint8), f"{name}.gx": gx})
        for d in range(4):
            out[f"{name}.g{d}"] = gs[d]
            out[f"{name}.A{d}"] = A[d]
            out[f"{name}.gw{d}"] = (gw0, gw1, gw2, gw3)[d]
    np.savez_compressed(os.path.join(HERE, "sga_golden.npz"), **out)

    # BASELINE.json configs[0]: SGA forward, 1x(C=1)x48x48x48, forward only
    shape = (1, 1, 48, 48, 48)
    x, gs, _ = sga_inputs(rng, shape, "randn")
    o, tmp, mask = ref.sga_forward(x, *gs)
    np.savez_compressed(os.path.join(HERE, "sga_cfg1_golden.npz"), x=x, g0=gs[0], g1=gs[1], g2=gs[2],
                        g3=gs[3], out=o, mask=mask.astype(np.uint8))

    out = {}
    for name, shape, r, passes in LGA_CASES:
        fs = list(shape)
        fs[-3] = 3 * (2 * r + 1) ** 2
        x = rng.standard_normal(shape).astype(np.float32)
        f = l1norm(rng.standard_normal(fs), -3)
        gy = rng.standard_normal(shape).astype(np.float32)
        y, ins = ref.lga_chain_forward(x, f, r, passes)
        gx, gf = ref.lga_chain_backward(ins, f, gy, r)
        out.update({f"{name}.x": x, f"{name}.f": f, f"{name}.gy": gy, f"{name}.y": y,
                    f"{name}.gx": gx, f"{name}.gf": gf,
                    f"{name}.meta": np.array([r, passes], np.int32)})
    np.savez_compressed(os.path.join(HERE, "lga_golden.npz"), **out)
    for fn in ("sga_golden.npz", "sga_cfg1_golden.npz", "lga_golden.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


# ---- full-size digests (cases, inputs and the digest walk: tests/golden_util.py, shared with the tests that hold the C
# restatement and the GPU to them) ----------------------------------------------------------------------------------
sys.path.insert(0, os.path.dirname(HERE))
import golden_util as gu  # noqa: E402


def digests():
    import json
    ref = Oracle("reference")
    out = {"_made_by": "tests/golden/make_golden.py --digests: sha256 of the float32 / uint8 bytes of each array computed by "
                       "oracle/_ref (the reference's own kernel bodies, GANet_kernel.cu:23-1269, launch order :935-1129, 1271-1322)"}
    for name, shape, seed in gu.SGA_DIGEST_CASES:
        out[name] = {"shape": list(shape), "seed": seed, "sha256": gu.sga_digests(ref, shape, seed)}
        print(name, "done")
    for name, shape, seed in gu.LGA_DIGEST_CASES:
        out[name] = {"shape": list(shape), "seed": seed, "sha256": gu.lga_digests(ref, shape, seed)}
        print(name, "done")
    with open(os.path.join(HERE, "digests.json"), "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    digests() if "--digests" in sys.argv else main()
