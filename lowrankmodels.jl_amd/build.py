"""In-tree build of the gfx950 shared libraries (explicit hipcc; hipcc cross-compiles without a GPU).

    libglrm_hip.so          the engine + C ABI (include/glrm_hip.h) -- the PRODUCT library: no test hook, no link emulator in it
    libglrm_hip_testing.so  the same objects, with csrc/glrm_testhooks.hip and csrc/glrm_multigpu.hip rebuilt under -DGLRM_HIP_TESTING: the
                            environment-driven test hooks (injected set-up failure, RCCL stand-in, link emulator) live only here; loaded by name
                            by the tests that need them (_capi.hip_testing_api) and by `bench.py --emulate-link-gbps`
    libglrm_synth.so        device-side synthetic workload generator (bench / test tooling)
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]

TARGETS = {
    "libglrm_hip.so": (["glrm_hip.hip", "glrm_tiled.hip", "glrm_dense.hip", "glrm_multi.hip", "glrm_subset.hip", "glrm_svd.hip", "glrm_impute.hip", "glrm_tilesort.hip", "glrm_multigpu.hip", "glrm_blocked.hip", "glrm_cached.hip", "glrm_reforder.hip", "glrm_transpose.hip", "glrm_testhooks.hip", "glrm_lane.hip"],
                       ["glrm_device.hpp", "glrm_fastmath.hpp", "glrm_tiled.hpp", "glrm_dense.hpp", "glrm_multi.hpp", "glrm_impute.hpp", "glrm_engine.hpp", "glrm_lane.hpp", "../../include/glrm_hip.h"]),
    "libglrm_synth.so": (["glrm_synth.hip"], ["../../include/glrm_synth.h"]),
}


# translation units that differ between the product library and the test build (everything else is shared object for object)
TESTING_UNITS = ("glrm_testhooks.hip", "glrm_multigpu.hip")
TESTING_LIB = "libglrm_hip_testing.so"
# -Bsymbolic: a library's calls to its own extern "C" entry points bind inside the library, so the product library and the test build can
# live in one process (the suite loads both) without one interposing on the other
LINK = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_all(force=False, verbose=True):
    built = []
    for name, (srcs, hdrs) in TARGETS.items():
        out = os.path.join(PKG, name)
        src_paths = [os.path.join(CSRC, s) for s in srcs]
        deps = src_paths + [os.path.normpath(os.path.join(CSRC, h)) for h in hdrs]
        if force or _stale(out, deps):
            # one hipcc -c per source, in parallel, then link
            objs, procs = [], []
            hdr_paths = deps[len(src_paths):]
            for sp in src_paths:
                obj = os.path.join(PKG, "build", os.path.basename(sp) + ".o")
                os.makedirs(os.path.dirname(obj), exist_ok=True)
                objs.append(obj)
                if not force and not _stale(obj, [sp] + hdr_paths):  # this object is newer than its source and every header
                    continue
                cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", sp, "-o", obj]
                if verbose:
                    print("[build]", " ".join(cmd), flush=True)
                procs.append((cmd, subprocess.Popen(cmd)))
            for cmd, pr in procs:
                if pr.wait() != 0:
                    raise subprocess.CalledProcessError(pr.returncode, cmd)
            link = [HIPCC] + LINK + objs + ["-ldl", "-o", out]
            if verbose:
                print("[build]", " ".join(link), flush=True)
            subprocess.run(link, check=True)
            built.append(name)
    built += build_testing(force=force, verbose=verbose)
    return built


def build_testing(force=False, verbose=True):
    """libglrm_hip_testing.so: the product library's objects with TESTING_UNITS recompiled under -DGLRM_HIP_TESTING."""
    srcs, hdrs = TARGETS["libglrm_hip.so"]
    out = os.path.join(PKG, TESTING_LIB)
    hdr_paths = [os.path.normpath(os.path.join(CSRC, h)) for h in hdrs]
    objs, procs = [], []
    for sname in srcs:
        sp = os.path.join(CSRC, sname)
        if sname not in TESTING_UNITS:
            objs.append(os.path.join(PKG, "build", sname + ".o"))
            continue
        obj = os.path.join(PKG, "build", "testing", sname + ".o")
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        objs.append(obj)
        if not force and not _stale(obj, [sp] + hdr_paths):
            continue
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-DGLRM_HIP_TESTING", "-c", sp, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if not force and not _stale(out, objs):
        return []
    link = [HIPCC] + LINK + objs + ["-ldl", "-o", out]
    if verbose:
        print("[build]", " ".join(link), flush=True)
    subprocess.run(link, check=True)
    return [TESTING_LIB]


def build_variant(tag, defines, verbose=True):
    """libglrm_hip_<tag>.so: the engine compiled with extra -D switches, next to the product library (same-box A/B of kernel variants with
    tests/perf/ab_lib.py; e.g. tag "libm", defines GLRM_LOGISTIC_LIBM GLRM_ORDINAL_BRANCHY GLRM_POISSON_LIBM = round 2's loss formulas)."""
    srcs, _ = TARGETS["libglrm_hip.so"]
    out = os.path.join(PKG, f"libglrm_hip_{tag}.so")
    objs, procs = [], []
    for sname in srcs:
        sp = os.path.join(CSRC, sname)
        obj = os.path.join(PKG, "build", tag, sname + ".o")
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-D" + d for d in defines] + ["-c", sp, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    subprocess.run([HIPCC] + LINK + objs + ["-ldl", "-o", out], check=True)
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":  # python -m lowrankmodels.jl_amd.build --variant libm GLRM_LOGISTIC_LIBM ...
        build_variant(sys.argv[2], sys.argv[3:])
    else:
        build_all(force=True)
