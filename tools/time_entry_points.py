#!/usr/bin/env python3
"""Device time of individual entry points (HIP events around back-to-back calls), for A/B runs of library variants:
RZK_LIB=ring_zk_amd/variants/lib_x.so python tools/time_entry_points.py [N] [n,k,l] [B] [V]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ring_zk_amd import Context  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n, k, l = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,3,1").split(","))
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
V = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ctx = Context(N, n, k, l)
ctx.generate_key(1)
half = (ctx.q - 1) // 2
sid = iter(range(100))
uni = lambda *lead: ctx.sample_uniform(1, next(sid), half, lead)
small = lambda *lead: ctx.sample_uniform(1, next(sid), 1, lead)
gauss = lambda *lead: ctx.sample_gauss(1, next(sid), float(ctx.sigma), lead)
a, b = uni(B), uni(B)
m = uni(B, l)
v, r, y, rp, yp = uni(B, k), small(B, k), gauss(B, k), small(B, k), gauss(B, k)
x = uni(B, l)
d = ctx.sample_challenge(1, next(sid), (B,))
c, t, ok = ctx.open_commit(x, r, y)
z = ctx.open_response(y, r, d)
lc = ctx.linear_commit(a, x, r, rp, y, yp)
lz = ctx.linear_response(y, yp, r, rp, d)
Bs = max(B // (4 * V), 1)
gs, xs = uni(Bs, V), uni(Bs, V, l)
rs, rps, ys, yps = small(Bs, V, k), small(Bs, k), gauss(Bs, V, k), gauss(Bs, k)
ds = ctx.sample_challenge(1, next(sid), (Bs,))
sc = ctx.sum_commit(gs, xs, rs, rps, ys, yps)
sz = ctx.sum_response(ys, yps, rs, rps, ds)


def timeit(name, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-28s %9.1f us" % (name, e0.elapsed_time(e1) * 1e3 / reps))


print(f"N={N} (n,k,l)=({n},{k},{l}) B={B} (sum: B={Bs}, V={V}) lib={os.environ.get('RZK_LIB', 'default')}")
timeit("polymul full x full", lambda: ctx.polymul(a, b))
timeit("polymul full x ternary", lambda: ctx.polymul(a, r[:, 0].contiguous()))
timeit("cmul rows=l", lambda: ctx.cmul(m, a))
timeit("matvec A (full v)", lambda: ctx.matvec(2, v))
timeit("matvec A (ternary v)", lambda: ctx.matvec(2, r))
timeit("matvec A1 (gauss v)", lambda: ctx.matvec(0, y))
timeit("commit", lambda: ctx.commit(x, r))
timeit("open_commit", lambda: ctx.open_commit(x, r, y))
timeit("open_response", lambda: ctx.open_response(y, r, d))
timeit("open_verify", lambda: ctx.open_verify(z, t, c, d))
timeit("linear_commit", lambda: ctx.linear_commit(a, x, r, rp, y, yp))
timeit("linear_response", lambda: ctx.linear_response(y, yp, r, rp, d))
timeit("linear_verify", lambda: ctx.linear_verify(lz[0], lz[1], lc[0], lc[1], a, lc[2], lc[3], lc[4], d))
timeit("sum_commit", lambda: ctx.sum_commit(gs, xs, rs, rps, ys, yps), 5)
timeit("sum_response", lambda: ctx.sum_response(ys, yps, rs, rps, ds), 5)
timeit("sum_verify", lambda: ctx.sum_verify(sz[0], sz[1], sc[0], sc[1], gs, sc[2], sc[3], sc[4], ds), 5)
