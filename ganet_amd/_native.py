"""ctypes binding of the C ABI in include/ganet_hip.h (libganet_hip.so).

The product path: `lib()` loads ganet_amd/libganet_hip.so -- the gfx950 build made by
`ganet_amd.build.build_hip()` / `__graft_entry__.build()` -- and raises if it is
missing.  There is no CPU fallback.  (`CApi` takes an explicit path so the test-suite
can bind the same ABI to the lockstep kernel emulator under tests/hipsim.)
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libganet_hip.so"
ABI_VERSION = 11

_P = ctypes.c_void_p
_I = ctypes.c_int

# name -> argument types (return type is always int unless listed in _RET)
_PROTOS = {
    "ganet_abi_version": [],
    "ganet_is_simulator": [],
    "ganet_set_option": [ctypes.c_char_p, _I],
    "ganet_get_option": [ctypes.c_char_p],
    "ganet_sga_scan_forward": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ganet_sga_forward": [_P] * 9 + [_I] * 5 + [_P],
    "ganet_sga_forward_infer": [_P] * 9 + [_I] * 5 + [_P],
    "ganet_sga_forward_infer_scratch": [_P] * 6 + [_I] * 5,
    "ganet_sga_backward_scan": [_P] * 5 + [_I] * 6 + [_P],
    "ganet_sga_merge": [_P] * 4 + [_I] * 5 + [_P],
    "ganet_sga_scan_forward_ws": [_P] * 3 + [_I] * 6 + [_P],
    "ganet_sga_backward_scan_ws": [_P] * 5 + [_I] * 6 + [_P],
    "ganet_sga_workspace_layout": [_I] * 5,
    "ganet_sga_backward_point": [_P] * 12 + [_I] * 5 + [_P],
    "ganet_sga_backward_dir": [_P] * 9 + [_I] * 7 + [_P],
    "ganet_sga_backward": [_P] * 15 + [_I] * 5 + [_P],
    "ganet_sga_forward_compat": [_P] * 8 + [_I] * 5 + [_P],
    "ganet_sga_backward_compat": [_P] * 15 + [_I] * 5 + [_P],
    "ganet_lga_forward": [_P] * 3 + [_I] * 5 + [_P],
    "ganet_lga_backward": [_P] * 5 + [_I] * 6 + [_P],
    "ganet_lga_forward_regress": [_P] * 5 + [_I] * 5 + [_P],
    "ganet_lga_apply_paired": [_P] * 3 + [_I] * 8 + [_P],
    "ganet_lga_apply_paired_edges": [_P] * 4 + [_I] * 8 + [_P],
    "ganet_lga_filter_grad_paired": [_P] * 3 + [_I] * 8 + [_P],
    "ganet_cost_volume_forward": [_P] * 3 + [_I] * 5 + [_P],
    "ganet_cost_volume_backward": [_P] * 3 + [_I] * 5 + [_P],
    "ganet_disparity_regression_forward": [_P] * 2 + [_I] * 4 + [_P],
    "ganet_disparity_regression_backward": [_P] * 2 + [_I] * 4 + [_P],
    "ganet_l1_normalize_forward": [_P] * 5 + [_I] * 6 + [_P],
    "ganet_l1_normalize_backward": [_P] * 6 + [_I] * 6 + [_P],
    "ganet_norm_disparity_regression_forward": [_P] * 3 + [_I] * 4 + [_P],
    "ganet_norm_disparity_regression_backward": [_P] * 5 + [_I] * 4 + [_P],
    "ganet_softmin_forward": [_P] * 2 + [_I] * 4 + [_P],
    "ganet_softmin_backward": [_P] * 3 + [_I] * 4 + [_P],
    "ganet_softmin_regression_forward": [_P] * 4 + [_I] * 4 + [_P],
    "ganet_softmin_regression_backward": [_P] * 6 + [_I] * 4 + [_P],
    "ganet_trilinear_upsample_forward": [_P] * 2 + [_I] * 7 + [_P],
    "ganet_trilinear_upsample_backward": [_P] * 2 + [_I] * 7 + [_P],
    "ganet_residual_relu_forward": [_P] * 5 + [_I] * 5 + [_P],
    "ganet_residual_relu_backward": [_P] * 5 + [_I] * 5 + [_P],
    "ganet_selftest_dpp": [_P, _P, _P],
    "ganet_selftest_dpp_wave": [_P, _P, _P],
}
EXPORTS = sorted(list(_PROTOS) + ["ganet_last_error"])


class GanetError(RuntimeError):
    """A C-ABI entry returned a negative code (GANET_E_INVALID -1, GANET_E_UNSUPPORTED -2, GANET_E_RUNTIME -3)."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


E_INVALID, E_UNSUPPORTED, E_RUNTIME = -1, -2, -3


class CApi:
    """Thin, typed view of one loaded libganet_*.so."""

    def __init__(self, path, strict=True):
        """strict=False (development A/B scripts only): bind what an OLDER build of the library exports and skip the ABI
        version check, so that last round's kernels can be timed beside this round's on one box."""
        if not os.path.exists(path):
            raise GanetError(
                f"{path} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        self.path = path
        self._lib = ctypes.CDLL(path)
        self._lib.ganet_last_error.restype = ctypes.c_char_p
        self._lib.ganet_last_error.argtypes = []
        for name, args in _PROTOS.items():
            try:
                fn = getattr(self._lib, name)
            except AttributeError:
                if strict:
                    raise
                continue
            fn.argtypes = args
            fn.restype = _I
        got = self._lib.ganet_abi_version()
        if got != ABI_VERSION and strict:
            raise GanetError(f"{path}: ABI version {got}, expected {ABI_VERSION}")
        self.is_simulator = bool(self._lib.ganet_is_simulator())

    def last_error(self):
        return self._lib.ganet_last_error().decode("utf-8", "replace")

    def has(self, name):
        """does this build export `name`? (strict=False loads of older builds)"""
        return hasattr(self._lib, name)

    def call(self, name, *args):
        rc = getattr(self._lib, name)(*args)
        if rc != 0:
            raise GanetError(f"{name} failed ({rc}): {self.last_error()}", rc)

    def query(self, name, *args):
        """entry points that answer with a non-negative number"""
        rc = getattr(self._lib, name)(*args)
        if rc < 0:
            raise GanetError(f"{name} failed ({rc}): {self.last_error()}", rc)
        return rc

    def set_option(self, name, value):
        self.call("ganet_set_option", name.encode(), int(value))

    def get_option(self, name):
        return self.query("ganet_get_option", name.encode())


_LIB = None
_LOCK = threading.Lock()


def lib_path():
    return os.path.join(_HERE, LIB_NAME)


def lib():
    """The process-wide gfx950 library; raises GanetError if it has not been built."""
    global _LIB
    if _LIB is None:
        with _LOCK:
            if _LIB is None:
                # The library links libamdhip64 by SONAME; PyTorch-ROCm ships its own copy.  Whichever is
                # mapped first serves both, so torch must come first: otherwise the process holds two HIP
                # runtimes and launches on torch-allocated memory fail with hipErrorNoDevice.
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
                _LIB = CApi(lib_path())
    return _LIB
