"""Builds libppca_hip.so (HIP kernels + C-ABI) for gfx950, in-tree.

    python -m ppca_rs_amd.build [--force]        (PPCA_FORCE_BUILD=1 in the environment: the same from any caller, e.g. __graft_entry__.build())

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels
with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libppca_hip.so")
SOURCES = ["ppca_kernels.hip", "ppca_em8.hip", "ppca_em9.hip", "ppca_em16.hip", "ppca_llk.hip", "ppca_generic.hip", "ppca_solve4.hip", "ppca_capi.hip", "ppca_comm.hip"]
DEPS = SOURCES + ["ppca_small.hpp", "ppca_internal.hpp", "ppca_solve.hpp", "ppca_handles.hpp", "ppca_device.hpp", os.path.join("..", "..", "include", "ppca_hip.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in DEPS)


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib_path: str = LIB) -> str:
    """extra_flags / lib_path: diagnostic builds (e.g. -DPPCA_PHASE_TIMING into another .so)."""
    force = force or os.environ.get("PPCA_FORCE_BUILD") == "1"  # (a clean-tree compile on demand: the mtime check re-uses prebuilt objects)
    if not force and not extra_flags and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    headers = [os.path.join(CSRC, f) for f in DEPS if f not in SOURCES]
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".diag.o" if extra_flags else ".o"))
        objs.append(obj)
        newest = max(os.path.getmtime(f) for f in [os.path.join(CSRC, src)] + headers)
        if not force and not extra_flags and os.path.exists(obj) and os.path.getmtime(obj) > newest:
            continue  # this object is current: only what changed is recompiled (the kernels take ~2 minutes)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra_flags, "-c", os.path.join(CSRC, src),
               "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), cmd))
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib_path


def build_example() -> str:
    """examples/em_train: a compiled host (plain g++) that links libppca_hip.so through include/ppca_hip.h only."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", "em_train.cpp")
    out = os.path.join(root, "examples", "em_train")
    if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return out
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), src, "-L" + HERE, "-lppca_hip",
                           "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


if __name__ == "__main__":
    if "--timing" in sys.argv:
        print(build(force=True, verbose=True, extra_flags=("-DPPCA_PHASE_TIMING",),
                    lib_path=os.path.join(HERE, "libppca_hip_timing.so")))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
        print(build_example())
