#!/usr/bin/env python3
"""Build library variants next to the default one: tools/ab_build.py name:DEF1=V,DEF2=V ...  -> ring_zk_amd/variants/lib_<name>.so (git-ignored, travels to the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ring_zk_amd import build  # noqa: E402

os.makedirs(os.path.join(ROOT, "ring_zk_amd", "variants"), exist_ok=True)
for spec in sys.argv[1:]:
    name, _, defs = spec.partition(":")
    out = os.path.join(ROOT, "ring_zk_amd", "variants", f"lib_{name}.so")
    build.build_library(out=out, defines=[d for d in defs.split(",") if d])
    print(name, out)
