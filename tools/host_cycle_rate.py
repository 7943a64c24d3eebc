#!/usr/bin/env python3
"""PCIe-inclusive rate of the OpenProof cycle: the host-pointer entry points (rzk_open_{commit,response,verify}_batch) on
numpy buffers, every call staging its inputs to the device and its outputs back (never `value`).
usage: tools/host_cycle_rate.py [--N 1024] [--batch 4096] [--reps 5]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ring_zk_amd import Context, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=1024)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
n, k, l = 1, 3, 1
ctx = Context(a.N, n, k, l)
rng = np.random.default_rng(1)
ctx.load_key(synth.key(rng, a.N, n, k, l))
B = a.batch
x = synth.uniform(rng, (B, l, a.N))
r = synth.small(rng, (B, k, a.N))
y = synth.gauss(rng, (B, k, a.N), ctx.sigma)
d = synth.challenge(rng, (B,), a.N, ctx.kappa)


def cycle():
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    acc = ctx.open_verify(z, t, c, d)
    return int(ok.sum()), int(acc.sum())


cycle()
t0 = time.perf_counter()
for _ in range(a.reps):
    ok, acc = cycle()
dt = (time.perf_counter() - t0) / a.reps
assert ok == B and acc == B
moved = 8 * a.N * B * (7 + 3 + 7 + 3 + 6)   # polys staged in and out by the three calls: commit 7 in / 3 out, response 7 / 3, verify 6
print(json.dumps({"workload": f"OpenProof cycle through the host-pointer entry points, N={a.N}, batch={B}",
                  "proofs_per_s": B / dt, "ms_per_cycle": dt * 1e3, "bytes_staged_per_cycle": moved,
                  "staging_GBps": moved / dt / 1e9}))
