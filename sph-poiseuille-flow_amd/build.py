"""Build libsphx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  The shared object lands next to the sources
(sph-poiseuille-flow_amd/csrc/libsphx.so) so it travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["sphx_common.hip", "sphx_pairlist.hip", "sphx_resident.hip"]
HEADERS = ["sphx_common.hpp", "sphx_device.hpp", "sphx_kernels.hpp", os.path.join("..", "..", "include", "sphx.h")]
LIB = os.path.join(CSRC, "libsphx.so")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libsphx.so)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in srcs:
        obj = src[:-4] + ".o"
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
