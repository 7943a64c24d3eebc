"""Compile the HIP sources into ring_zk_amd/librzk_hip.so with hipcc for gfx950 (in-tree build)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "librzk_hip.so")
SOURCES = ["rzk_kernels.hip", "rzk_api.cpp", "rzk_wire.cpp"]
HEADERS = ["rzk_core.h", "rzk_dev.h", "rzk_tables.h"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the ring-zk MI355X backend needs the ROCm toolchain to build")


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(HERE, "..", "include", "rzk.h"))
    return os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps)


def build_library(force: bool = False, verbose: bool = False, out: str = SO, defines=()) -> str:
    """`out` / `defines` build tuning variants next to the default library (used by tools/ only)."""
    if out == SO and not force and not needs_build():
        return SO
    cmd = [_hipcc(), "-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-shared", "-o", out]
    cmd += [f"-D{d}" for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
