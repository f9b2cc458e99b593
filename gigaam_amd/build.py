"""Build libgigaam_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libgigaam_hip.so")


def _newest_src() -> float:
    inc = os.path.join(os.path.dirname(_HERE), "include", "gigaam_hip.h")
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [inc]
    return max(os.path.getmtime(f) for f in files)


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest_src():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           os.path.join(CSRC, "gam_api.hip"), "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT
