#!/usr/bin/env python3
"""Compile source variants in parallel and print the register / scratch usage of selected kernels (no GPU).
usage: tools/variant_resources.py <dir-with-variant-subdirs> [kernel-name-prefix ...]"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

base = sys.argv[1]
prefixes = sys.argv[2:] or ["row_kernel<1", "unit_kernel<10"]


def run(v):
    src = os.path.join(base, v, "csrc", "rzk_kernels.hip")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c", src, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    return v, p.stderr


with cf.ThreadPoolExecutor(6) as ex:
    for v, txt in ex.map(run, sorted(os.listdir(base))):
        print("==", v)
        errs = [l for l in txt.splitlines() if "error" in l]
        if errs:
            print("\n".join(errs[:5]))
            continue
        cur, rows = None, []
        for line in txt.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
        names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
        for r, n in zip(rows, names):
            n = re.sub(r"\(.*", "", n).replace("void rzk::", "")
            if any(n.startswith(p) for p in prefixes):
                print("%-52s vgpr %3d scratch %4d occ %d" % (n, r.get("vgpr", -1), r.get("scratch", -1), r.get("occ", -1)))
