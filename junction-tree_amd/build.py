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
INST = ["jtp_inst_%s_%s.hip" % (fam, t) for fam in ("multi", "mix", "mixc", "both", "flow", "level", "shape") for t in ("f32", "f64")]
SOURCES = ["jtp_plan.cpp", "jtp_engine.hip"] + INST       # (compiled in parallel, one object each, then linked)
DEPS = SOURCES + ["jtp_internal.h", "jtp_plan.h", "jtp_kernels.hip.h", os.path.join("..", "..", "include", "jtprop.h")]


ID_FILE = os.path.join(LIBDIR, "BUILD_ID")


def source_id():
    """Digest of everything libjtprop.so is compiled from: identifies the CODE a profile was measured on
    (jtp_version() ends in it; tools/collect_profiles.sh stamps it into the files under profiles/)."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        with open(os.path.join(CSRC, d), "rb") as fh:
            h.update(d.encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def built_id():
    """Source digest of the library in place (without the "+flags" suffix an experimental build carries)."""
    try:
        with open(ID_FILE) as fh:
            return fh.read().split()[0].split("+")[0]
    except (OSError, IndexError):
        return None


def needs_build():
    if not os.path.exists(LIB):
        return True
    try:                       # (the sources do not travel with every copy of the tree: then the built library stands)
        return built_id() != source_id()
    except OSError:
        return False


def build(force=False, verbose=True, extra=(), out=None, jobs=None):
    out = out or LIB
    if out == LIB and extra:
        raise SystemExit("experimental flags (%s) build to --out, not onto the product library" % " ".join(extra))
    if out == LIB and not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    import shutil
    import tempfile
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    sid = source_id() + ("+" + "".join(sorted(extra)).replace(" ", "") if extra else "")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
             '-DJTP_SOURCE_ID="%s"' % sid] + list(extra)
    objdir = tempfile.mkdtemp(prefix="jtprop_build_")
    # objects of translation units whose sources did not change are taken from a cache beside the library (git-ignored):
    # a planner-only change recompiles jtp_plan.cpp and the engine, not the twelve kernel instantiation units
    cache = os.path.join(LIBDIR, "objcache")
    os.makedirs(cache, exist_ok=True)
    import hashlib
    headers = {"jtp_plan.cpp": ["jtp_plan.h", "jtp_internal.h", os.path.join("..", "..", "include", "jtprop.h")],
               "jtp_engine.hip": ["jtp_plan.h", "jtp_internal.h", "jtp_kernels.hip.h", os.path.join("..", "..", "include", "jtprop.h")]}
    try:                       # (the compiler is part of what an object is made of: a toolchain upgrade must not link old objects)
        toolchain = subprocess.check_output([hipcc, "--version"], stderr=subprocess.STDOUT)
    except (OSError, subprocess.CalledProcessError):
        toolchain = hipcc.encode()
    try:
        def compile_one(src):
            obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
            h = hashlib.sha256()
            h.update(toolchain)
            for d in [src] + headers.get(src, ["jtp_internal.h", "jtp_kernels.hip.h"]):
                with open(os.path.join(CSRC, d), "rb") as fh:
                    h.update(d.encode() + b"\0" + fh.read())
            # (the source id is compiled into the engine only: the other units do not change with it)
            tu_flags = [f for f in flags if not f.startswith("-DJTP_SOURCE_ID") or src == "jtp_engine.hip"]
            h.update(" ".join(tu_flags).encode())
            kept = os.path.join(cache, os.path.splitext(src)[0] + "." + h.hexdigest()[:16] + ".o")
            if os.path.exists(kept):
                shutil.copy(kept, obj)
                return obj
            cmd = [hipcc] + tu_flags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            # (the four newest objects of a unit are kept: the product and a -D variant of it can be rebuilt in turn without either
            #  pushing the other's objects out)
            mine = sorted((f for f in os.listdir(cache) if f.startswith(os.path.splitext(src)[0] + ".") and f.endswith(".o")),
                          key=lambda f: os.path.getmtime(os.path.join(cache, f)))
            for old in mine[:-3]:
                os.remove(os.path.join(cache, old))
            tmp_kept = kept + ".tmp%d" % os.getpid()      # (another build may be reading the cache: entries appear whole or not at all)
            shutil.copy(obj, tmp_kept)
            os.replace(tmp_kept, kept)
            return obj
        # (the template translation units take 1-2 GiB of compiler each: at most eight at once, JTP_BUILD_JOBS overrides)
        jobs = jobs or int(os.environ.get("JTP_BUILD_JOBS", 0)) or max(1, min(len(SOURCES), os.cpu_count() or 2, 8))
        with ThreadPoolExecutor(jobs) as pool:          # (the heaviest translation units are listed first)
            objs = list(pool.map(compile_one, INST + ["jtp_engine.hip", "jtp_plan.cpp"]))
        # link beside the target and move the result into place: a process that has the old library mapped keeps it,
        # nobody ever sees a half-written one
        tmp_out = out + ".tmp%d" % os.getpid()
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp_out, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp_out, out)
        finally:
            if os.path.exists(tmp_out):
                os.remove(tmp_out)
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    if out == LIB:
        head = "unknown"
        try:
            head = subprocess.check_output(["git", "-C", HERE, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
            if subprocess.check_output(["git", "-C", HERE, "status", "--porcelain", "--", CSRC, os.path.join(HERE, "..", "include")],
                                       stderr=subprocess.DEVNULL).strip():
                head += "+uncommitted"
        except (OSError, subprocess.CalledProcessError):
            pass
        with open(ID_FILE, "w") as fh:
            fh.write("%s\ngit %s\n" % (sid, head))
    return out


if __name__ == "__main__":
    # python build.py [--force] [--out path.so] [-DNAME ...]   (the last two: experimental builds)
    args = sys.argv[1:]
    out = args[args.index("--out") + 1] if "--out" in args else None
    print("built", build(force="--force" in args or out is not None, out=out, extra=[a for a in args if a.startswith("-D")]))
