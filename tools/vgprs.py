#!/usr/bin/env python3
"""Register / scratch / LDS use per kernel from an assembly dump:  hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only ...
usage: tools/vgprs.py k.s [name-filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
    name, scr, sg, vg = m.groups()
    if flt in name:
        short = re.sub(r"^_ZN3rzk\d+", "", name)
        short = re.sub(r"EvP.*$", "", short)
        print(f"{short:40s} vgpr {vg:>4s} sgpr {sg:>4s} scratch {scr}")
