"""Builds ganet_amd/libganet_hip.so for gfx950 with hipcc (in-tree, no JIT cache)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# translation units and the extra flags each is compiled with
SOURCES = {
    "ganet_capi.hip": [],
    # hipcc's SLP vectoriser packs the scalar FMAs of the horizontal recurrences into v_pk_fma_f32 and pays
    # for it in v_mov shuffles (forward scan 0.098 -> 0.082 ms without it, the adjoint spills with it);
    # everything else is a few per cent faster with it
    "sga_row_tu.hip": ["-fno-slp-vectorize"],
}
LIB_SONAME = "libganet_hip.so"
OUT = os.path.join(_HERE, LIB_SONAME)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, extra_flags=(), out=OUT, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> libganet_hip.so.  hipcc cross-compiles without a GPU."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # every file under csrc/ (sources, headers, textually included kernels): a new include cannot be forgotten here
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    deps += [os.path.join(os.path.dirname(_HERE), "include", "ganet_hip.h"), os.path.abspath(__file__)]
    if not force and not _stale(out, deps):
        return out
    if not os.path.exists(hipcc):
        if os.path.exists(out):
            return out      # GPU box without a toolchain: use the prebuilt library as shipped
        raise RuntimeError("hipcc not found and no prebuilt libganet_hip.so")
    # Objects and the not-yet-complete library live in a directory of this build's own (two processes building at once share
    # nothing); whatever happens -- a failed or interrupted compile included -- the directory goes and the other compiles are
    # stopped (per-PID objects used to stay behind in csrc/ after a failure).
    import tempfile
    procs = []
    with tempfile.TemporaryDirectory(prefix="ganet_build_", dir=os.path.dirname(os.path.abspath(out))) as tmp:
        try:
            objs = []
            for src, extra in SOURCES.items():
                obj = os.path.join(tmp, src + ".o")
                cmd = [hipcc] + HIPCC_FLAGS + extra + list(extra_flags) + ["-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                procs.append((cmd, subprocess.Popen(cmd)))
                objs.append(obj)
            for cmd, pr in procs:
                if pr.wait() != 0:
                    raise subprocess.CalledProcessError(pr.returncode, cmd)
            lib_tmp = os.path.join(tmp, os.path.basename(out))
            cmd = [hipcc, "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC", "-Wl,-soname," + LIB_SONAME] + objs + ["-o", lib_tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            os.replace(lib_tmp, out)          # same directory tree as `out`: an atomic rename
        finally:
            for _, pr in procs:
                if pr.poll() is None:
                    pr.kill()
                    pr.wait()
    return out


EXT_SRC = os.path.join(CSRC, "ganet_torch_ext.cpp")
# where the reference's `from ..build.lib import GANet` (libs/GANet/functions/GANet.py:3) looks
EXT_DIR = os.path.join(os.path.dirname(_HERE), "libs", "GANet", "build", "lib")


def ext_path():
    import sysconfig
    return os.path.join(EXT_DIR, "GANet" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_torch_ext(force=False, verbose=False):
    """The pybind module `GANet` (csrc/ganet_torch_ext.cpp: the reference's six-function native surface on top of the
    C ABI) -> libs/GANet/build/lib/GANet.<abi>.so.  Host-only C++ (no device code): g++ against the torch headers,
    linked to libganet_hip.so by rpath: next to the module ($ORIGIN, for a copy into a GANet checkout's
    libs/GANet/build/lib/ together with libganet_hip.so) or in this tree's ganet_amd/."""
    out = ext_path()
    deps = [EXT_SRC, os.path.join(os.path.dirname(_HERE), "include", "ganet_hip.h"), os.path.abspath(__file__)]
    if not force and not _stale(out, deps):
        return out
    cxx = shutil.which("g++")
    if cxx is None:
        if os.path.exists(out):
            return out
        raise RuntimeError("g++ not found and no prebuilt GANet extension module")
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(EXT_DIR, exist_ok=True)
    for d in (os.path.dirname(EXT_DIR), EXT_DIR):             # `..build.lib` must be a package chain
        init = os.path.join(d, "__init__.py")
        if not os.path.exists(init):
            open(init, "w").close()
    inc = ce.include_paths("cuda") if "device_type" in ce.include_paths.__code__.co_varnames else ce.include_paths(True)
    inc = [i for i in inc if os.path.isdir(i)] + [sysconfig.get_paths()["include"]]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-deprecated-declarations",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=GANet", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    for i in inc:
        cmd += ["-isystem", i]
    cmd += [EXT_SRC, "-o", out + ".tmp", "-L", tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
            "-ltorch_python", "-L", _HERE, "-lganet_hip",
            "-Wl,-rpath,$ORIGIN:$ORIGIN/../../../../ganet_amd", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_torch_ext(force=True, verbose=True))
