"""Builds ganet_amd/libganet_hip.so for gfx950 with hipcc (in-tree, no JIT cache)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["ganet_capi.hip"]
HEADERS = ["ga_common.h", "sga_kernels.h", "sga_row_kernels.h", "sga_col_kernels.h", "lga_kernels.h", "misc_kernels.h"]
OUT = os.path.join(_HERE, "libganet_hip.so")
# -fno-slp-vectorize: hipcc otherwise packs neighbouring scalar FMAs of the scan recurrences into
# v_pk_fma_f32 and pays for it in v_mov shuffles (12 per scan position); where packing helps (LGA) the
# kernels use explicit 2-vectors, which this flag does not touch
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
               "-fno-slp-vectorize"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, extra_flags=(), out=OUT, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> libganet_hip.so.  hipcc cross-compiles without a GPU."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "ganet_hip.h"))
    if not force and not _stale(out, deps):
        return out
    if not os.path.exists(hipcc):
        if os.path.exists(out):
            return out      # GPU box without a toolchain: use the prebuilt library as shipped
        raise RuntimeError("hipcc not found and no prebuilt libganet_hip.so")
    cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-I", CSRC] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
