"""numpy/ctypes front-end of the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module (see the header of ``ganet_oracle.c``).  Two back-ends share
one interface:

* ``Oracle("port")``      -> ``libganet_oracle.so`` (this repo's C restatement)
* ``Oracle("reference")`` -> ``_ref/libganet_ref.so`` (the reference's own kernel
  bodies host-compiled through ``ref_shim/``; exists only where it was built)

The method layer restates the buffer roles and pass chaining of the reference's
``libs/GANet/functions/GANet.py`` (cited per method).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = ctypes.POINTER(ctypes.c_float)


def build(quiet=True):
    """Compile the checkers (gcc only).  `make ref` is a no-op without /root/reference."""
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True)


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_F)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def have(kind):
    path = {"port": "libganet_oracle.so", "reference": os.path.join("_ref", "libganet_ref.so")}[kind]
    return os.path.exists(os.path.join(_HERE, path))


class Oracle:
    def __init__(self, kind="port"):
        self.kind = kind
        if kind == "port":
            path = os.path.join(_HERE, "libganet_oracle.so")
            if not os.path.exists(path):
                build()
            self.lib = ctypes.CDLL(path)
            self.threads = self.lib.oracle_num_threads()
        elif kind == "reference":
            path = os.path.join(_HERE, "_ref", "libganet_ref.so")
            if not os.path.exists(path):
                build()
            self.lib = ctypes.CDLL(path)
            self.threads = self.lib.ref_num_threads()
        else:
            raise ValueError(kind)

    # -- SGA ---------------------------------------------------------------
    def sga_scan(self, x, g, direction):
        """One directional volume A_dir (GANet_kernel.cu:66-127 and mirrors)."""
        x, g = _f32(x), _f32(g)
        N, C, D, H, W = x.shape
        A = np.empty_like(x)
        if self.kind == "port":
            self.lib.oracle_sga_scan_forward(_p(x), _p(g), _p(A), None, N * C, D, H, W, direction)
        else:
            self.lib.ref_sga_scan(_p(x), _p(g), _p(A), N, C, D, H, W, direction)
        return A

    def sga_forward(self, x, g0, g1, g2, g3):
        """SgaFunction.forward (functions/GANet.py:10-22) -> (output, temp_out, mask)."""
        x, g0, g1, g2, g3 = map(_f32, (x, g0, g1, g2, g3))
        N, C, D, H, W = x.shape
        out, tmp, mask = np.zeros_like(x), np.zeros_like(x), np.zeros_like(x)
        if self.kind == "port":
            self.lib.oracle_sga_forward(_p(x), _p(g0), _p(g1), _p(g2), _p(g3), _p(tmp), _p(out),
                                        _p(mask), None, N, C, D, H, W)
        else:
            self.lib.ref_sga_forward(_p(x), _p(g0), _p(g1), _p(g2), _p(g3), _p(tmp), _p(out),
                                     _p(mask), N, C, D, H, W)
        return out, tmp, mask

    def sga_backward(self, x, g0, g1, g2, g3, temp_out, mask, grad_out):
        """SgaFunction.backward (functions/GANet.py:24-48) -> (gradInput, grad0..grad3)."""
        x, g0, g1, g2, g3, mask, grad_out = map(_f32, (x, g0, g1, g2, g3, mask, grad_out))
        tmp = _f32(temp_out).copy()
        N, C, D, H, W = x.shape
        gx = np.zeros_like(x)
        gw = [np.zeros_like(g0) for _ in range(4)]
        tgrad = np.zeros_like(x)
        idx = np.zeros((N, C, H, W), np.float32)
        fn = self.lib.oracle_sga_backward if self.kind == "port" else self.lib.ref_sga_backward
        fn(_p(x), _p(g0), _p(g1), _p(g2), _p(g3), _p(tmp), _p(mask), _p(idx), _p(grad_out),
           _p(tgrad), _p(gx), _p(gw[0]), _p(gw[1]), _p(gw[2]), _p(gw[3]), N, C, D, H, W)
        return (gx, *gw)

    # -- LGA (4-D [N,D,H,W] or 5-D [N,C,D,H,W]; the 5-D form folds N*C) -------
    def _lga_dims(self, x, f, radius):
        if x.ndim == 5:
            B, D = x.shape[0] * x.shape[1], x.shape[2]
        else:
            B, D = x.shape[0], x.shape[1]
        H, W = x.shape[-2:]
        assert f.shape[-3] == 3 * (2 * radius + 1) ** 2
        return B, D, H, W

    def lga_forward(self, x, f, radius):
        """One pass: lga_cuda_forward / lga3d_cuda_forward (GANet_kernel.cu:1271-1338)."""
        x, f = _f32(x), _f32(f)
        B, D, H, W = self._lga_dims(x, f, radius)
        y = np.zeros_like(x)
        fn = self.lib.oracle_lga_forward if self.kind == "port" else self.lib.ref_lga_forward
        fn(_p(x), _p(f), _p(y), B, D, H, W, radius)
        return y

    def lga_backward(self, x, f, gy, radius, gf=None):
        """One pass: lga_cuda_backward (:1299-1322); gf is accumulated into if given."""
        x, f, gy = _f32(x), _f32(f), _f32(gy)
        B, D, H, W = self._lga_dims(x, f, radius)
        gx = np.empty_like(x)
        gf = np.zeros_like(f) if gf is None else gf
        fn = self.lib.oracle_lga_backward if self.kind == "port" else self.lib.ref_lga_backward
        fn(_p(x), _p(f), _p(gy), _p(gx), _p(gf), B, D, H, W, radius)
        return gx, gf

    def lga_chain_forward(self, x, f, radius, passes):
        """Lga{,2,3}Function / Lga3d{,2,3}Function forward: `passes` chained passes
        (functions/GANet.py:176-187, 54-66).  Returns (output, [inputs of each pass])."""
        ins = [_f32(x)]
        for _ in range(passes):
            ins.append(self.lga_forward(ins[-1], f, radius))
        return ins[-1], ins[:-1]

    def lga_chain_backward(self, pass_inputs, f, gy, radius):
        """Backward of the chain (functions/GANet.py:189-203, 68-83): walk the passes in
        reverse, accumulating gradFilters across passes."""
        gf = np.zeros_like(_f32(f))
        g = _f32(gy)
        for xin in reversed(pass_inputs):
            g, gf = self.lga_backward(xin, f, g, radius, gf)
        return g, gf

    # -- GetCostVolume / DisparityRegression (port only; the reference is plain
    #    torch there, modules/GANet.py:114-148, and is checked against torch) ---
    def cost_volume(self, x, y, maxdisp):
        x, y = _f32(x), _f32(y)
        N, C, H, W = x.shape
        Dn = maxdisp + 1
        cost = np.empty((N, 2 * C, Dn, H, W), np.float32)
        self.lib.oracle_cost_volume(_p(x), _p(y), _p(cost), N, C, Dn, H, W)
        return cost

    def disparity_regression(self, x, maxdisp):
        x = _f32(x)
        N, Dn, H, W = x.shape
        assert Dn == maxdisp + 1
        out = np.empty((N, H, W), np.float32)
        self.lib.oracle_disparity_regression(_p(x), _p(out), N, Dn, H, W)
        return out
