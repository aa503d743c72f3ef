"""Builds libdspeed_hip.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m dspeed_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdspeed_hip.so")
SOURCES = ["dsp_vm.hip", "dsp_energy.hip", "dsp_fit.hip", "dsp_host.cpp"]
DEPS = SOURCES + ["dsp_program.h", "dsp_wave.h", os.path.join("..", "..", "include", "dspeed_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), *FLAGS, "-x", "hip", *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
