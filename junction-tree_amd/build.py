"""Build libjtprop.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python junction-tree_amd/build.py [--force]

The shared library lands in junctiontree_amd/lib/ (git-ignored; it travels to the GPU box
with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "junctiontree_amd", "lib")
LIB = os.path.join(LIBDIR, "libjtprop.so")
SOURCES = ["jtp_plan.cpp", "jtp_engine.hip"]
DEPS = SOURCES + ["jtp_internal.h", "jtp_plan.h", "jtp_kernels.hip.h", os.path.join("..", "..", "include", "jtprop.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=True, extra=(), out=None):
    out = out or LIB
    if out == LIB and not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-Wno-unused-variable"] + list(extra)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", out, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    # python build.py [--force] [--out path.so] [-DNAME ...]   (the last two: experimental builds)
    args = sys.argv[1:]
    out = args[args.index("--out") + 1] if "--out" in args else None
    print("built", build(force="--force" in args or out is not None, out=out, extra=[a for a in args if a.startswith("-D")]))
