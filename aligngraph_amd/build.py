"""Builds aligngraph_amd/libagx.so for gfx950 (hipcc for the kernels, g++ for the host side)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libagx.so")
CLI = os.path.join(HERE, "AlignGraph_amd")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HOST_SRC = ["agx_engine.cpp", "agx_host.cpp", "agx_walk.cpp", "agx_load.cpp"]
DEV_SRC = ["agx_kernels.hip"]
HEADERS = ["agx_core.h", "agx_host.h", "agx_parse.h", "agx_mem.h", "agx_kargs.h", os.path.join("..", "..", "include", "agx.h")]


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False, defines=(), out=None, tag=""):
    """defines/out/tag: build an experimental variant (e.g. defines=["AGX_MAXV_LDS=2u"], out="/tmp/libagx_v2.so", tag="v2")."""
    global LIB
    lib = out or LIB
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    objdir = os.path.join(HERE, "_obj" + tag)
    dflags = ["-D" + d for d in defines]
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for s in DEV_SRC:
        o = os.path.join(objdir, s + ".o")
        if force or _stale(o, [os.path.join(CSRC, s)] + hdrs):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + dflags + ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    for s in HOST_SRC:
        o = os.path.join(objdir, s + ".o")
        if force or _stale(o, [os.path.join(CSRC, s)] + hdrs):
            cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include")] + dflags + \
                  ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L" + os.path.join(ROCM, "lib"), "-lhsa-runtime64"]      # (HSA: the downloads go through its asynchronous copy)
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # the AlignGraph-compatible command line (rows f1/f3/f4): plain C++ against the C-ABI
    if out is None:
        cli_src = os.path.join(CSRC, "agx_cli.cpp")
        if force or _stale(CLI, [cli_src, lib, os.path.join(HERE, "..", "include", "agx.h")]):
            cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", cli_src, "-o", CLI, "-L" + HERE, "-lagx", "-Wl,-rpath,$ORIGIN"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
