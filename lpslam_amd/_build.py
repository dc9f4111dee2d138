"""Builds the HIP/C-ABI shared library in-tree for gfx950 (no JIT cache: the .so travels with the repo snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblpslam_hip.so")
HIP_SOURCES = ["api.hip", "frontend.hip", "match.hip", "ba.hip", "bow.hip", "share.hip"]
DEPS = ["internal.h", "orb_pattern.inc", os.path.join("..", "..", "include", "lpslam_hip.h"), "sim3.inl"]
# -ffp-contract=off: parity with the CPU definition forbids FMA contraction ("Numerics").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


OBJ_DIR = os.path.join(CSRC, "_obj")
# what each translation unit includes (beyond itself): a change there recompiles only that unit
UNIT_DEPS = {"ba.hip": ["sim3.inl", "ba_build.inl", "ba_solve.inl", "ba_band.inl", "ba_update.inl"], "frontend.hip": ["orb_pattern.inc"], "match.hip": [], "api.hip": [], "bow.hip": [], "share.hip": []}
COMMON_DEPS = ["internal.h", os.path.join("..", "..", "include", "lpslam_hip.h")]


def hip_library(force=False, verbose=False):
    """One object per translation unit (compiled in parallel, rebuilt only when its sources changed), then the link."""
    srcs = [s for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"] + os.environ.get("LPSLAM_HIP_EXTRA_FLAGS", "").split()
    # objects compiled with other flags (a development build with -DLPSLAM_..._STAMPS) are stale whatever their age
    flags_file = os.path.join(OBJ_DIR, "flags.txt")
    flags_now = " ".join(compile_flags)
    if not os.path.exists(flags_file) or open(flags_file).read() != flags_now:
        force = True
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        deps = [os.path.join(CSRC, d) for d in [src] + UNIT_DEPS.get(src, []) + COMMON_DEPS]
        deps = [d for d in deps if os.path.exists(d)] + [os.path.abspath(__file__)]
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
            cmd = [hipcc] + compile_flags + ["-c", "-o", obj, os.path.join(CSRC, src)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, p in jobs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    with open(flags_file, "w") as f:
        f.write(flags_now)
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


HOST_LIB = os.path.join(HERE, "liblpslam.so")
HOST_SOURCES = ["slam_manager.cpp", "hip_tracker.cpp", "interface.cpp", "rectify.cpp", "replay.cpp", "two_view.cpp", "bow.cpp", "jpeg.cpp"]


def host_library(force=False, verbose=False):
    """C++ host mirror of the reference interface (g++), linked against the HIP C-ABI library next to it."""
    hdir = os.path.join(HERE, "host")
    srcs = [os.path.join(hdir, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(hdir, h) for h in ("core.h", "json_min.h", "hip_tracker.h", "slam_manager.h", "rectify.h", "replay.h", "two_view.h", "bow.h", "jpeg.h")] + \
        [os.path.join(HERE, "..", "include", h) for h in ("lpslam_types.h", "lpslam_manager.h", "lpslam_hip.h")] + [LIB]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps if os.path.exists(d)):
        return HOST_LIB
    hip_library(force=False, verbose=verbose)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wextra",
           "-Wno-unused-parameter", "-pthread", "-o", HOST_LIB] + srcs + ["-L" + HERE, "-llpslam_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == "__main__":
    print(hip_library(force="-f" in sys.argv, verbose=True))
    print(host_library(force="-f" in sys.argv, verbose=True))
