"""Builds libblsw.so (HIP, gfx950) in-tree. hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libblsw.so")
SOURCES = ["kernels.hip"]
HOST_SOURCES = ["r1cs.cpp"]  # host-only C++ (the constraint-matrix emitter): compiled by g++, linked into the same library
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".cuh", ".h")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HOST_SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "blsw.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None, defines=()):
    """out / defines: an alternative build next to the shipped one (A/B runs: BLSW_LIB=<out> selects it at import)"""
    if out is None and not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in HOST_SOURCES:
        obj = os.path.join(HERE, "build_" + os.path.splitext(src)[0] + ".o")
        subprocess.check_call([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-c", os.path.join(CSRC, src), "-o", obj], cwd=CSRC)
        objs.append(obj)
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-o", out or OUT] + list(defines) + [os.path.join(CSRC, s) for s in SOURCES] + ["-Wl," + o for o in objs]  # objects go straight to the linker (hipcc would read them as HIP source)
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    # python build.py [--force] [--out <file.so> -DNAME=VALUE ...]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    build(force="--force" in sys.argv, verbose=out is None, out=out, defines=[a for a in sys.argv[1:] if a.startswith("-D")])
