#!/usr/bin/env python3
"""BASELINE config 1 on the CPU restatement: per-phase single-thread latency of the Open / Linear / Sum proofs at
N = 512, (1,3,1), VL = 4 — the workload of the reference's criterion benches (benches/bench.rs:31,200), timed
like them (1 s warm-up, >= 10 samples, mean).  Labelled "CPU restatement (oracle)", never "reference Rust"."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from ring_zk_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = O.Params(N=N)
rng = np.random.default_rng(1)
A = synth.key(rng, N, 1, 3, 1)
V = 4
x = synth.uniform(rng, (1, N)); g = synth.uniform(rng, N)
r, rp = synth.small(rng, (3, N)), synth.small(rng, (3, N))
y, yp = synth.gauss(rng, (3, N), P.sigma), synth.gauss(rng, (3, N), P.sigma)
d = synth.challenge(rng, (), N, 36)
gs = synth.uniform(rng, (V, N)); xs = synth.uniform(rng, (V, 1, N)); rs = synth.small(rng, (V, 3, N)); ys = synth.gauss(rng, (V, 3, N), P.sigma)

def bench(f, measure=1.0):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0: f()          # warm-up 1 s
    n, t0 = 0, time.perf_counter()
    while n < 10 or time.perf_counter() - t0 < measure:
        f(); n += 1
    return (time.perf_counter() - t0) / n * 1e9        # mean ns

res = {}
c, t, _ = O.open_commit(P, A, x, r, y); z = O.open_response(P, y, r, d)
res["open_proof_commit"] = bench(lambda: O.open_commit(P, A, x, r, y))
res["open_proof_create_response"] = bench(lambda: O.open_response(P, y, r, d))
res["open_proof_verify"] = bench(lambda: O.open_verify(P, A, z, t, c, d))
lc = O.linear_commit(P, A, g, x, r, rp, y, yp); lz = O.linear_response(P, y, yp, r, rp, d)
res["linear_proof_commit"] = bench(lambda: O.linear_commit(P, A, g, x, r, rp, y, yp), 2.0)
res["linear_proof_create_response"] = bench(lambda: O.linear_response(P, y, yp, r, rp, d), 2.0)
res["linear_proof_verify"] = bench(lambda: O.linear_verify(P, A, lz[0], lz[1], lc[0], lc[1], g, lc[2], lc[3], lc[4], d), 2.0)
sc = O.sum_commit(P, A, gs, xs, rs, rp, ys, yp); sz = O.sum_response(P, ys, yp, rs, rp, d)
res["sum_proof_commit"] = bench(lambda: O.sum_commit(P, A, gs, xs, rs, rp, ys, yp), 4.0)
res["sum_proof_create_response"] = bench(lambda: O.sum_response(P, ys, yp, rs, rp, d), 4.0)
res["sum_proof_verify"] = bench(lambda: O.sum_verify(P, A, sz[0], sz[1], sc[0], sc[1], gs, sc[2], sc[3], sc[4], d), 4.0)
print(json.dumps({"what": "CPU restatement (oracle, schoolbook), single thread, mean ns per call", "N": N, "ns": res}))
