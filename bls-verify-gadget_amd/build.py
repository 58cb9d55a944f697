"""Builds libblsw.so (HIP, gfx950) in-tree. hipcc cross-compiles without a GPU.

One translation unit per kernel family (csrc/k_*.hip) plus the engine / C ABI (csrc/engine.hip) and the host-only constraint-matrix
emitter (csrc/r1cs.cpp, g++); the units are compiled in parallel and linked into one shared library."""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libblsw.so")
OBJ = os.path.join(HERE, "build_obj")
# the shipped translation units, listed: a stray or experimental .hip file in csrc/ is an error, not silently linked
SOURCES = ["engine.hip", "k_bench.hip", "k_cofactor.hip", "k_cofv.hip", "k_g1.hip", "k_g2.hip", "k_map.hip", "k_miller_par.hip", "k_pairing_lane.hip", "k_prepare.hip", "k_sha.hip",
           "k_sign.hip", "k_stream.hip", "k_team.hip", "k_values.hip"]
HOST_SOURCES = ["r1cs.cpp"]  # host-only C++: compiled by g++, linked into the same library
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".h")))
# the one-instance-per-lane chain units are compiled a second time with their programs inlined (kernels *_inl: kcommon.hpp)
DUAL = ["k_sha.hip", "k_g1.hip", "k_g2.hip", "k_map.hip", "k_cofactor.hip", "k_prepare.hip"]
# the units whose chains have a latency compilation: one chain on the four lanes of a quad (kernels *_q: kcommon.hpp, fp.hpp)
QUAD = ["k_map.hip", "k_cofv.hip", "k_prepare.hip", "k_g2.hip"]
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]


_REASON = {"text": ""}


def needs_build():
    """True when libblsw.so is missing, older than a source / header, or tagged by other compilers or flags than the present ones. A library that
    arrives WITHOUT its tag file (the tag is git-ignored like the library; a copy may carry only the .so) is accepted when it is newer than every
    source: it is not silently rebuilt — a full -O3 build of 40 units — on the strength of a missing side file. The reason for a rebuild is kept
    in build_reason() and logged once by build()."""
    if not os.path.exists(OUT):
        _REASON["text"] = "no library"
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HOST_SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "blsw.h")]
    newer = [d for d in deps if os.path.getmtime(d) > t]
    if newer:
        _REASON["text"] = "%s is newer than the library" % os.path.basename(newer[0])
        return True
    try:  # built by other compilers / flags than the present ones: rebuild (the tag is written next to the library)
        have = open(OUT + ".tag").read().strip()
    except OSError:
        return False
    if have != build_tag():
        _REASON["text"] = "the library's tag %s is not this toolchain's %s (other compiler release or flags)" % (have, build_tag())
        return True
    return False


def build_reason():
    return _REASON["text"]


def _compile(job):
    cmd, log = job
    p = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if log:
        open(log, "w").write(p.stdout)
    if p.returncode:
        sys.stderr.write(p.stdout)
        raise subprocess.CalledProcessError(p.returncode, cmd)
    return p.stdout


_TOOL_VERSION = {}
_TAGS = {}


def _tool_version(tool):
    if tool not in _TOOL_VERSION:
        try:
            _TOOL_VERSION[tool] = subprocess.run([tool, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        except OSError:
            _TOOL_VERSION[tool] = "missing"
    return _TOOL_VERSION[tool]


def build_tag(defines=()):
    """Key of the object cache: everything that shapes an object besides its source and headers — the compilers (path and --version: the
    ROCm release), the flags, the defines and the list of doubly compiled units. "std" names the shipped configuration's logs
    (resource_table); the objects carry the hash."""
    hipcc, cxx = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), os.environ.get("CXX", "g++")
    memo = (hipcc, cxx, tuple(defines))
    if memo in _TAGS:  # one pair of compiler processes per Python process, not per call
        return _TAGS[memo]
    key = "\n".join([hipcc, _tool_version(hipcc), cxx, _tool_version(cxx), " ".join(HIP_FLAGS), " ".join(HOST_FLAGS), " ".join(defines), " ".join(DUAL), " ".join(QUAD)])
    _TAGS[memo] = hashlib.sha1(key.encode()).hexdigest()[:8]
    return _TAGS[memo]


def build(force=False, verbose=False, out=None, defines=()):
    """out / defines: an alternative build next to the shipped one (A/B runs: BLSW_LIB=<out> selects it at import).
    Objects are cached per (source, build_tag): a unit is recompiled when it, a header, a compiler or a flag changed."""
    stray = sorted(set(f for f in os.listdir(CSRC) if f.endswith(".hip")) - set(SOURCES))
    if stray or any(not os.path.exists(os.path.join(CSRC, f)) for f in SOURCES + HOST_SOURCES):
        raise RuntimeError("csrc/ does not hold exactly the listed translation units (unlisted: %s)" % ", ".join(stray))
    if out is None and not force and not needs_build():
        return OUT
    if out is None and not force and os.path.exists(OUT):
        print("[blsw build] rebuilding libblsw.so: %s" % build_reason(), file=sys.stderr)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.abspath(out) if out else None
    os.makedirs(OBJ, exist_ok=True)
    tag = build_tag(defines)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_time = max(hdr_time, os.path.getmtime(os.path.join(HERE, "..", "include", "blsw.h")))
    jobs, objs = [], []
    units = [(src, ()) for src in HOST_SOURCES + SOURCES] + [(src, ("-DBLSW_KVARIANT_INL",)) for src in DUAL] + [(src, ("-DBLSW_KVARIANT_QUAD",)) for src in QUAD]
    for src, extra in units:
        suffix = "_inl" if "-DBLSW_KVARIANT_INL" in extra else ("_q" if extra else "")
        obj = os.path.join(OBJ, "%s%s.%s.o" % (os.path.splitext(src)[0], suffix, tag))
        objs.append(obj)
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            continue
        if src.endswith(".cpp"):
            cmd = [os.environ.get("CXX", "g++")] + HOST_FLAGS + ["-c", path, "-o", obj]
            jobs.append((cmd, None))
        else:
            cmd = [hipcc] + HIP_FLAGS + list(defines) + list(extra) + ["-c", path, "-o", obj, "-Rpass-analysis=kernel-resource-usage"]
            jobs.append((cmd, obj + ".log"))
    if verbose:
        print("compiling %d units" % len(jobs), file=sys.stderr)
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(_compile, jobs))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or OUT] + objs, cwd=CSRC)
    open((out or OUT) + ".tag", "w").write(tag + "\n")
    return out or OUT


def resource_table(tag=None):
    """kernel -> registers / scratch / occupancy, from the compile logs (-Rpass-analysis=kernel-resource-usage)"""
    import re

    rows = []
    tag = tag or build_tag()
    for f in sorted(os.listdir(OBJ)):
        if not f.endswith(".%s.o.log" % tag):
            continue
        cur = None
        for ln in open(os.path.join(OBJ, f)):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark: .*?    (VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", ln)
            if m and cur is not None:
                cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return rows


if __name__ == "__main__":
    # python build.py [--force] [--table] [--out <file.so> -DNAME=VALUE ...]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    build(force="--force" in sys.argv, verbose=True, out=out, defines=defs)
    if "--table" in sys.argv:
        for r in resource_table(build_tag(defs)):
            name = r["name"]
            if "k_sha_expand" in name and "ILi384ELi8ELi16ELi0" not in name:
                continue
            print("%-60s VGPR %3d AGPR %3d scratch %5d occupancy %d spilled %4d LDS %5d" % (name[:60], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("ScratchSize", 0), r.get("Occupancy", 0), r.get("VGPRs Spill", 0), r.get("LDS Size", 0)))
