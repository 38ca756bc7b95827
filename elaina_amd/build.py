"""Build the HIP shared library (libwost_hip.so) for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container as the
"does it build" check and the resulting .so travels to the GPU box with the tree.
"""
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libwost_hip.so")
HOST_EXE = os.path.join(LIB_DIR, "elaina-exec")
OBJ_DIR = os.path.join(LIB_DIR, "obj")

SOURCES = ["wost_hip.hip", "wost_order.hip", "wost_hip3d.hip", "wost_build2.hip", "wost_build3.hip", "wost_vmm3.hip", "wost_guided3.hip", "wost_vmm.hip", "wost_net.hip", "wost_guided.hip", "lbvh_build.cpp"]

# -ffp-contract=off is part of the arithmetic contract (DESIGN.md "deterministic math")
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
    "-Wall", "-Wno-unused-function",
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    """One object per translation unit (compiled in parallel, rebuilt when the unit or ANY header
    of csrc/ or include/wost.h is newer, or when the compiler flags differ from those the objects
    were built with), then one link."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    headers.append(os.path.join(_HERE, "..", "include", "wost.h"))
    # developer knob: extra -D definitions for kernel-tuning experiments (WOST_HIPCC_DEFS="-DX=1 -DY=2")
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + os.environ.get("WOST_HIPCC_DEFS", "").split()
    # objects are only reused under the flags they were compiled with: a library carrying experimental -D variants must
    # never be mistaken for the default build by the next plain build (and the other way round)
    stamp = os.path.join(OBJ_DIR, "flags.stamp")
    flag_id = hashlib.sha256(" ".join(flags).encode()).hexdigest()
    try:
        same_flags = open(stamp).read().strip() == flag_id
    except OSError:
        same_flags = False
    if not same_flags:
        force = True
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    # wait for every compiler before reporting a failure: none is left running behind an exception
    failed = [(cmd, p.returncode) for cmd, p in jobs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
    with open(stamp, "w") as f:
        f.write(flag_id + "\n")
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


def source_id():
    """identity of the kernel sources (csrc/*.h, *.hip, *.cpp and include/wost.h): committed counter files carry the id of
    the tree they were measured on, and bench.py marks them stale when the running tree differs"""
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp")))
    for f in files + [os.path.join(_HERE, "..", "include", "wost.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


HOST_DIR = os.path.join(_HERE, "host")
HOST_SOURCES = ["main.cpp", "exec.cpp", os.path.join("core", "problem.cpp"),
                os.path.join("integrator", "common.cpp"), os.path.join("integrator", "uniform", "integrator.cpp"),
                os.path.join("integrator", "guided", "integrator.cpp"), os.path.join("util", "json.cpp"),
                os.path.join("util", "image_io.cpp")]


def build_host(force=False, verbose=False):
    """elaina-exec: the C++ host mirror of the reference's entry point, linked against
    libwost_hip.so (rpath = $ORIGIN, both live in elaina_amd/lib)."""
    build_library(force=force, verbose=verbose)
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_SOURCES]
    deps = list(srcs)
    for root, _, files in os.walk(HOST_DIR):
        deps += [os.path.join(root, f) for f in files if f.endswith(".h")]
    deps.append(LIB_PATH)
    if force or _stale(HOST_EXE, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", HOST_DIR] + srcs + [
            "-L", LIB_DIR, "-lwost_hip", "-Wl,-rpath,$ORIGIN", "-o", HOST_EXE]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_EXE


if __name__ == "__main__":
    print(build_host(force=True, verbose=True))
    print(build_library(force=True, verbose=True))
