"""Builds ganet_amd/libganet_hip.so for gfx950 with hipcc (in-tree, no JIT cache)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# translation units and the extra flags each is compiled with
SOURCES = {
    "ganet_capi.hip": [],
    # hipcc's SLP vectoriser packs the scalar FMAs of the horizontal forward recurrence into v_pk_fma_f32
    # and pays for it in v_mov shuffles (0.098 -> 0.082 ms per scan without it); everything else is a
    # few per cent faster with it
    "sga_row_fwd_tu.hip": ["-fno-slp-vectorize"],
}
HEADERS = ["ga_common.h", "ga_launch.h", "sga_kernels.h", "sga_row_kernels.h", "sga_col_kernels.h", "lga_kernels.h",
           "misc_kernels.h"]
OUT = os.path.join(_HERE, "libganet_hip.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, extra_flags=(), out=OUT, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> libganet_hip.so.  hipcc cross-compiles without a GPU."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    deps = [os.path.join(CSRC, f) for f in list(SOURCES) + HEADERS]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "ganet_hip.h"))
    if not force and not _stale(out, deps):
        return out
    if not os.path.exists(hipcc):
        if os.path.exists(out):
            return out      # GPU box without a toolchain: use the prebuilt library as shipped
        raise RuntimeError("hipcc not found and no prebuilt libganet_hip.so")
    objs, procs = [], []
    for src, extra in SOURCES.items():
        obj = os.path.join(_HERE, "csrc", src + ".o")
        cmd = [hipcc] + HIPCC_FLAGS + extra + list(extra_flags) + ["-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC"] + objs + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    for o in objs:
        os.remove(o)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
