"""Compile the HIP sources into ring_zk_amd/librzk_hip.so with hipcc for gfx950 (in-tree build).

The library is git-ignored (it travels to the GPU box with the working tree), so "is it up to date" cannot rely on
file times: a stamp file next to the .so records a hash of every file under csrc/ plus include/rzk.h and the
compiler command; build_library() recompiles whenever that hash differs."""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "librzk_hip.so")
SOURCES = ["rzk_kernels.hip", "rzk_api.cpp", "rzk_wire.cpp"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the ring-zk MI355X backend needs the ROCm toolchain to build")


def _flags(defines=()):
    return ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-shared"] + [f"-D{d}" for d in defines]


def source_hash(defines=()) -> str:
    """Hash of everything the library is built from: csrc/* (sources and every header), include/rzk.h, flags."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*"))) + [os.path.join(HERE, "..", "include", "rzk.h")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(_flags(defines)).encode())
    return h.hexdigest()


def needs_build(out: str = SO, defines=()) -> bool:
    stamp = out + ".stamp"
    if not os.path.exists(out) or not os.path.exists(stamp):
        return True
    with open(stamp) as fh:
        return fh.read().strip() != source_hash(defines)


def build_library(force: bool = False, verbose: bool = False, out: str = SO, defines=()) -> str:
    """`out` / `defines` build tuning variants next to the default library (used by tools/ only)."""
    if not force and not needs_build(out, defines):
        return out
    cmd = [_hipcc()] + _flags(defines) + ["-o", out]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(out + ".stamp", "w") as fh:
        fh.write(source_hash(defines) + "\n")
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
