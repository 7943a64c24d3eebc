#!/usr/bin/env python3
"""Run ONE entry point a few times (for rocprofv3 counter passes): tools/run_one.py <polymul|matvec1|open_commit|open_verify|cmul> [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ring_zk_amd import Context  # noqa: E402

what = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N, n, k, l = 1024, 1, 3, 1
ctx = Context(N, n, k, l)
ctx.generate_key(1)
half = (ctx.q - 1) // 2
uni = lambda s, *lead: ctx.sample_uniform(1, s, half, lead)
a, b = uni(0, B), uni(1, B)
x = uni(2, B, l)
r = ctx.sample_uniform(1, 3, 1, (B, k))
y = ctx.sample_gauss(1, 4, float(ctx.sigma), (B, k))
d = ctx.sample_challenge(1, 5, (B,))
c, t, ok = ctx.open_commit(x, r, y)
z = ctx.open_response(y, r, d)
fn = {"polymul": lambda: ctx.polymul(a, b), "matvec1": lambda: ctx.matvec(0, y), "cmul": lambda: ctx.cmul(x, a),
      "open_commit": lambda: ctx.open_commit(x, r, y), "open_verify": lambda: ctx.open_verify(z, t, c, d)}[what]
for _ in range(4):
    fn()
torch.cuda.synchronize()
