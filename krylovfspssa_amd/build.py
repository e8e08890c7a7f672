"""Build libkfsp_hip.so (gfx950 only) in-tree with hipcc.

The shared object is git-ignored but travels to the GPU box with the
repository snapshot; `python -m krylovfspssa_amd.build` (or
__graft_entry__.build()) recreates it.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libkfsp_hip.so")
SOURCES = ["kfsp_api.cpp", "kfsp_group.cpp", "kfsp_padm.cpp", "kfsp_stepper.cpp", "kfsp_kernels.hip", "kfsp_build.hip", "kfsp_drop.hip", "kfsp_expand.hip", "kfsp_prop.hip", "kfsp_ssa.hip"]
HEADERS = ["kfsp_internal.h", "kfsp_ctx.h", "kfsp_prop_dev.h", "kfsp_hash_dev.h", os.path.join("..", "..", "include", "kfsp.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required; there is no CPU build of this library)")


def _objects():
    return [(os.path.join(CSRC, s), os.path.join(LIBDIR, "obj", s.rsplit(".", 1)[0] + ".o")) for s in SOURCES]


def _obj_stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + [os.path.join(CSRC, h) for h in HEADERS])


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """One object per source (only the stale ones are recompiled, in parallel), then one link."""
    if not force and not stale():
        return LIB
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    cc = _hipcc()
    jobs = []
    for src, obj in _objects():
        if force or _obj_stale(src, obj):
            cmd = [cc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", src, "-o", obj, "-I/opt/rocm/include",
                   "-Wno-ignored-attributes"]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [cc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [o for _, o in _objects()] + \
          ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
