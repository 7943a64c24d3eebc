#!/usr/bin/env python3
"""The reference's twelve criterion benches (benches/bench.rs: {open,linear,sum}_proof_{commit,generate_challenge,
create_response,verify}, N = 512, Params::default(), VL = 4 summands) on the GPU path: mean time per call of each phase
for a single proof (B = 1, what criterion times) and per proof inside a batch (default 4096).  Inputs are resident in
HBM; generate_challenge is the device-side challenge sampler (it does no ring arithmetic in the reference either).
Prints one JSON object: {"<bench name>": {"single_us": ..., "batched_ns_per_proof": ...}, ...}.

usage: tools/bench_phases.py [--N 512] [--batch 4096] [--reps 200]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ring_zk_amd import Context  # noqa: E402


def timed(fn, reps):
    for _ in range(max(3, reps // 10)):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=512)          # benches/bench.rs:31
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--summands", type=int, default=4)     # VL, benches/bench.rs:200
    args = ap.parse_args()
    N, V = args.N, args.summands
    ctx = Context(N)                                       # Params::default(): (n,k,l) = (1,3,1), kappa 36, b 1
    ctx.generate_key(7)
    half, sig = (ctx.q - 1) // 2, float(ctx.sigma)
    out = {}
    for B, tag in ((1, "single_us"), (args.batch, "batched_ns_per_proof")):
        sid = iter(range(1, 64))

        def uni(*lead):
            return ctx.sample_uniform(11, next(sid), half, lead)

        def small(*lead):
            return ctx.sample_uniform(11, next(sid), ctx.b, lead)

        def gauss(*lead):
            return ctx.sample_gauss(11, next(sid), sig, lead)

        reps = args.reps if B == 1 else max(20, args.reps // 4)
        scale = 1e6 if B == 1 else 1e9 / B
        d = ctx.sample_challenge(11, 0, (B,))
        res = {}
        # ---- open
        x, r, y = uni(B, 1), small(B, 3), gauss(B, 3)
        res["open_proof_commit"] = timed(lambda: ctx.open_commit(x, r, y), reps)
        c, t, _ = ctx.open_commit(x, r, y)
        res["open_proof_generate_challenge"] = timed(lambda: ctx.sample_challenge(12, 0, (B,)), reps)
        res["open_proof_create_response"] = timed(lambda: ctx.open_response(y, r, d), reps)
        z = ctx.open_response(y, r, d)
        res["open_proof_verify"] = timed(lambda: ctx.open_verify(z, t, c, d), reps)
        # ---- linear
        g, rp, yp = uni(B), small(B, 3), gauss(B, 3)
        res["linear_proof_commit"] = timed(lambda: ctx.linear_commit(g, x, r, rp, y, yp), reps)
        lc = ctx.linear_commit(g, x, r, rp, y, yp)
        res["linear_proof_generate_challenge"] = res["open_proof_generate_challenge"]
        res["linear_proof_create_response"] = timed(lambda: ctx.linear_response(y, yp, r, rp, d), reps)
        lz, lzp = ctx.linear_response(y, yp, r, rp, d)
        res["linear_proof_verify"] = timed(lambda: ctx.linear_verify(lz, lzp, lc[0], lc[1], g, lc[2], lc[3], lc[4], d), reps)
        # ---- sum
        gs, xs, rs, ys = uni(B, V), uni(B, V, 1), small(B, V, 3), gauss(B, V, 3)
        res["sum_proof_commit"] = timed(lambda: ctx.sum_commit(gs, xs, rs, rp, ys, yp), max(10, reps // 4))
        sc = ctx.sum_commit(gs, xs, rs, rp, ys, yp)
        res["sum_proof_generate_challenge"] = res["open_proof_generate_challenge"]
        res["sum_proof_create_response"] = timed(lambda: ctx.sum_response(ys, yp, rs, rp, d), max(10, reps // 4))
        szs, szp = ctx.sum_response(ys, yp, rs, rp, d)
        res["sum_proof_verify"] = timed(lambda: ctx.sum_verify(szs, szp, sc[0], sc[1], gs, sc[2], sc[3], sc[4], d),
                                        max(10, reps // 4))
        acc = ctx.sum_verify(szs, szp, sc[0], sc[1], gs, sc[2], sc[3], sc[4], d)
        assert int(acc.sum()) == B and int(ctx.open_verify(z, t, c, d).sum()) == B
        for name, sec in res.items():
            out.setdefault(name, {})[tag] = round(sec * scale, 3)
    print(json.dumps({"N": N, "summands": V, "batch": args.batch, "benches": out}, indent=1))


if __name__ == "__main__":
    main()
