#!/usr/bin/env python3
"""Generate tests/golden/golden.json — golden vectors for the ring-zk hot path.

The reference (Rust; /root/reference) cannot be built or imported in this image (no cargo/rustc,
its dependencies are not vendored, and the ring multiply lives in the absent third-party crate
poly-ring-xnp1 ^0.3 — Cargo.toml:18), and none of its tests pins a numeric product mod q
(SURVEY.md §8c).  These vectors are therefore produced by an INDEPENDENT arbitrary-precision
restatement in pure Python integers (no numpy, no C), seeded from the inputs that do appear in
the reference's own tests, and are used to pin oracle/rzk_oracle.c and the HIP path:

  * Mat tests, N=4 (src/mat.rs:243-268, 389-406): products of the literal test polynomials;
  * norm KATs (src/polynomial.rs:107-120), sigma KAT (src/params.rs:145-150), bound table
    (SURVEY §8a row a8);
  * x^(N-1) * x = -1 and all-(Q-1)/2 operands at N = 512/1024/2048 (max-magnitude CRT check);
  * protocol tuples (c, t, z, accept) for Open / Linear / Sum at N=16 (the size tests/test.rs:8
    uses) with seeded inputs, plus a tampered-z reject case per proof type.

Run:  python tests/golden/make_golden.py      (writes tests/golden/golden.json, ~1 minute)
"""
import json
import math
import os
import random

Q = 3515337053
HALF = (Q - 1) // 2


def center(v, q=Q):
    r = v % q
    return r - q if r > (q - 1) // 2 else r


def pmul(a, b, q=Q):
    n = len(a)
    acc = [0] * n
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            t = i + j
            if t < n:
                acc[t] += ai * bj
            else:
                acc[t - n] -= ai * bj
    return [center(v, q) for v in acc]


def padd(a, b, q=Q):
    return [center(x + y, q) for x, y in zip(a, b)]


def psub(a, b, q=Q):
    return [center(x - y, q) for x, y in zip(a, b)]


def mat_dot(A, B, q=Q):  # A: m x n, B: n x p (lists of lists of polys)
    m, n, p = len(A), len(B), len(B[0])
    N = len(B[0][0])
    out = [[[0] * N for _ in range(p)] for _ in range(m)]
    for i in range(m):
        for j in range(p):
            for k in range(n):
                out[i][j] = padd(out[i][j], pmul(A[i][k], B[k][j], q), q)
    return out


def col(v):  # vector of polys -> m x 1 matrix
    return [[p] for p in v]


def uncol(M):
    return [row[0] for row in M]


def isqrt(x):
    return math.isqrt(x)


def norm2(p):
    return isqrt(sum(c * c for c in p))


def sigma(b, kappa, k, N):
    return b * (11 * kappa) * isqrt(k * N)


def commit_bound(b, kappa, k, N):
    return 4 * sigma(b, kappa, k, N) * isqrt(N)


def verify_bound(b, kappa, k, N):
    return 2 * sigma(b, kappa, k, N) * isqrt(N)


class Prm:
    def __init__(self, N, n=1, k=3, l=1, kappa=36, b=1):
        self.N, self.n, self.k, self.l, self.kappa, self.b = N, n, k, l, kappa, b

    def d(self):
        return dict(N=self.N, n=self.n, k=self.k, l=self.l, kappa=self.kappa, b=self.b, q=Q)


def rand_full(rng, N):
    return [rng.randint(-HALF, HALF) for _ in range(N)]


def rand_small(rng, N, b):
    return [rng.randint(-b, b) for _ in range(N)]


def rand_gauss(rng, N, sig):
    return [int(rng.gauss(0.0, sig)) for _ in range(N)]  # truncation toward zero, like `as i64`


def rand_challenge(rng, N, kappa):
    kap = min(kappa, N)
    c = [(1 if rng.random() < 0.5 else -1) for _ in range(kap)] + [0] * (N - kap)
    rng.shuffle(c)
    return c


def key_build(P, rng):
    N, n, k, l = P.N, P.n, P.k, P.l
    zero, one = [0] * N, [1] + [0] * (N - 1)
    a1 = [[(one if j == i else zero) for j in range(n)] + [rand_full(rng, N) for _ in range(k - n)]
          for i in range(n)]
    a2 = [[zero] * n + [(one if j == i else zero) for j in range(l)] +
          [rand_full(rng, N) for _ in range(k - n - l)] for i in range(l)]
    return a1, a2


def commit(P, a1, a2, x, r):
    A = a1 + a2
    ar = uncol(mat_dot(A, col(r)))
    z = [[0] * P.N for _ in range(P.n)] + x
    c = [padd(u, v) for u, v in zip(ar, z)]
    ok = all(norm2(p) <= commit_bound(P.b, P.kappa, P.k, P.N) for p in r)
    return c, ok


def response(y, r, d):
    return [padd(yi, pmul(ri, d)) for yi, ri in zip(y, r)]


def c1c2(P, c):  # Commitment::c1_c2 -> split_rows(n): (first m-n rows, last n rows)
    m = len(c)
    return c[:m - P.n], c[m - P.n:]


def check_norm(P, polys):
    vb = verify_bound(P.b, P.kappa, P.k, P.N)
    return all(norm2(p) <= vb for p in polys)


def a1_relation(a1, z, t, c1, d):
    lhs = uncol(mat_dot(a1, col(z)))
    rhs = [padd(ti, pmul(ci, d)) for ti, ci in zip(t, c1)]
    return lhs == rhs


def open_case(P, seed, tamper=False):
    rng = random.Random(seed)
    a1, a2 = key_build(P, rng)
    x = [rand_full(rng, P.N) for _ in range(P.l)]
    r = [rand_small(rng, P.N, P.b) for _ in range(P.k)]
    sg = sigma(P.b, P.kappa, P.k, P.N)
    y = [rand_gauss(rng, P.N, sg) for _ in range(P.k)]
    d = rand_challenge(rng, P.N, P.kappa)
    c, ok = commit(P, a1, a2, x, r)
    t = uncol(mat_dot(a1, col(y)))
    z = response(y, r, d)
    if tamper:
        z[1][3] = center(z[1][3] + 1)
    c1, _ = c1c2(P, c)
    acc = check_norm(P, z) and a1_relation(a1, z, t, c1, d)
    return dict(params=P.d(), seed=seed, A=a1 + a2, x=x, r=r, y=y, d=d, c=c, t=t, z=z,
                commit_ok=ok, accept=bool(acc), tampered=tamper)


def linear_case(P, seed, tamper=False):
    rng = random.Random(seed)
    a1, a2 = key_build(P, rng)
    x = [rand_full(rng, P.N) for _ in range(P.l)]
    g = rand_full(rng, P.N)
    r = [rand_small(rng, P.N, P.b) for _ in range(P.k)]
    rp = [rand_small(rng, P.N, P.b) for _ in range(P.k)]
    sg = sigma(P.b, P.kappa, P.k, P.N)
    y = [rand_gauss(rng, P.N, sg) for _ in range(P.k)]
    yp = [rand_gauss(rng, P.N, sg) for _ in range(P.k)]
    d = rand_challenge(rng, P.N, P.kappa)
    gx = [pmul(xi, g) for xi in x]
    cp, okp = commit(P, a1, a2, gx, rp)
    c, ok = commit(P, a1, a2, x, r)
    t = uncol(mat_dot(a1, col(y)))
    tp = uncol(mat_dot(a1, col(yp)))
    a2y = uncol(mat_dot(a2, col(y)))
    a2yp = uncol(mat_dot(a2, col(yp)))
    u = [psub(pmul(p, g), q_) for p, q_ in zip(a2y, a2yp)]
    z, zp = response(y, r, d), response(yp, rp, d)
    if tamper:
        zp[0][0] = center(zp[0][0] - 1)
    c1, c2 = c1c2(P, c)
    c1p, c2p = c1c2(P, cp)
    acc = check_norm(P, z) and check_norm(P, zp)
    acc = acc and a1_relation(a1, z, t, c1, d) and a1_relation(a1, zp, tp, c1p, d)
    if acc:
        lhs = [psub(pmul(p, g), q_) for p, q_ in
               zip(uncol(mat_dot(a2, col(z))), uncol(mat_dot(a2, col(zp))))]
        rhs = [padd(pmul(psub(pmul(a, g), b_), d), ui) for a, b_, ui in zip(c2, c2p, u)]
        acc = lhs == rhs
    return dict(params=P.d(), seed=seed, A=a1 + a2, x=x, g=g, r=r, rp=rp, y=y, yp=yp, d=d, c=c, cp=cp,
                t=t, tp=tp, u=u, z=z, zp=zp, commit_ok=(1 if ok else 0) | (2 if okp else 0),
                accept=bool(acc), tampered=tamper)


def sum_case(P, V, seed, tamper=False):
    rng = random.Random(seed)
    a1, a2 = key_build(P, rng)
    xs = [[rand_full(rng, P.N) for _ in range(P.l)] for _ in range(V)]
    gs = [rand_full(rng, P.N) for _ in range(V)]
    rs = [[rand_small(rng, P.N, P.b) for _ in range(P.k)] for _ in range(V)]
    rp = [rand_small(rng, P.N, P.b) for _ in range(P.k)]
    sg = sigma(P.b, P.kappa, P.k, P.N)
    ys = [[rand_gauss(rng, P.N, sg) for _ in range(P.k)] for _ in range(V)]
    yp = [rand_gauss(rng, P.N, sg) for _ in range(P.k)]
    d = rand_challenge(rng, P.N, P.kappa)
    xp = None
    for x, g in zip(xs, gs):
        term = [pmul(xi, g) for xi in x]
        xp = term if xp is None else [padd(a, b_) for a, b_ in zip(xp, term)]
    cp, okp = commit(P, a1, a2, xp, rp)
    cs, ok = [], okp
    for x, r in zip(xs, rs):
        c, o = commit(P, a1, a2, x, r)
        cs.append(c)
        ok = ok and o
    ts = [uncol(mat_dot(a1, col(y))) for y in ys]
    tp = uncol(mat_dot(a1, col(yp)))
    u = None
    for y, g in zip(ys, gs):
        term = [pmul(p, g) for p in uncol(mat_dot(a2, col(y)))]
        u = term if u is None else [padd(a, b_) for a, b_ in zip(u, term)]
    u = [psub(a, b_) for a, b_ in zip(u, uncol(mat_dot(a2, col(yp))))]
    zs = [response(y, r, d) for y, r in zip(ys, rs)]
    zp = response(yp, rp, d)
    if tamper:
        zs[V - 1][2][5] = center(zs[V - 1][2][5] + 2)
    acc = all(check_norm(P, z) for z in zs) and check_norm(P, zp)
    if acc:
        for z, t, c in zip(zs, ts, cs):
            acc = acc and a1_relation(a1, z, t, c1c2(P, c)[0], d)
        acc = acc and a1_relation(a1, zp, tp, c1c2(P, cp)[0], d)
    if acc:
        lhs = rhs = None
        for z, g, c in zip(zs, gs, cs):
            term = [pmul(p, g) for p in uncol(mat_dot(a2, col(z)))]
            lhs = term if lhs is None else [padd(a, b_) for a, b_ in zip(lhs, term)]
            term = [pmul(p, g) for p in c1c2(P, c)[1]]
            rhs = term if rhs is None else [padd(a, b_) for a, b_ in zip(rhs, term)]
        lhs = [psub(a, b_) for a, b_ in zip(lhs, uncol(mat_dot(a2, col(zp))))]
        rhs = [padd(pmul(psub(a, b_), d), ui) for a, b_, ui in zip(rhs, c1c2(P, cp)[1], u)]
        acc = lhs == rhs
    return dict(params=P.d(), V=V, seed=seed, A=a1 + a2, xs=xs, gs=gs, rs=rs, rp=rp, ys=ys, yp=yp, d=d,
                cs=cs, cp=cp, ts=ts, tp=tp, u=u, zs=zs, zp=zp, commit_ok=bool(ok), accept=bool(acc),
                tampered=tamper)


def main():
    G = {"q": Q}
    # --- Mat-test inputs (src/mat.rs:244-253, 390-398), N = 4, small integers
    pad = lambda v: v + [0] * (4 - len(v))
    a00, a01, b00, b10 = pad([1, 2, 3]), pad([4, 5, 6]), pad([1, 2]), pad([3, 4])
    G["mat_dot_n4"] = dict(a=[a00, a01], b=[b00, b10],
                           out=padd(pmul(a00, b00), pmul(a01, b10)))
    b = pad([1, 2, 3])
    G["mat_cmul_n4"] = dict(a=[a00, a01], elem=b, out=[pmul(a00, b), pmul(a01, b)])
    # --- norm KATs (src/polynomial.rs:107-120)
    G["norm_kat"] = dict(p=[1, -2, 3, -4], norm1=10, norm2=5, norm_inf=4)
    # --- sigma KAT (src/params.rs:145-150) and the bound table
    G["sigma_kat"] = dict(b=1, kappa=36, k=3, N=1024, sigma=sigma(1, 36, 3, 1024))
    G["bounds"] = [dict(N=N, k=k, sigma=sigma(1, 36, k, N), commit=commit_bound(1, 36, k, N),
                        verify=verify_bound(1, 36, k, N))
                   for (N, k) in [(512, 3), (1024, 3), (1024, 9), (2048, 17), (16, 3)]]
    # --- wrap-around and max-magnitude products
    ext = []
    for N in (512, 1024, 2048):
        xn1 = [0] * (N - 1) + [1]
        x1 = [0, 1] + [0] * (N - 2)
        ext.append(dict(N=N, name="x^(N-1)*x", a=xn1, b=x1, out=pmul(xn1, x1)))
        hp = [HALF] * N
        hm = [-HALF] * N
        ext.append(dict(N=N, name="all(+half)*all(+half)", a=hp, b=hp, out=pmul(hp, hp)))
        alt = [HALF if i % 2 == 0 else -HALF for i in range(N)]
        ext.append(dict(N=N, name="all(+half)*alternating", a=hp, b=alt, out=pmul(hp, alt)))
        ext.append(dict(N=N, name="all(-half)*all(+half)", a=hm, b=hp, out=pmul(hm, hp)))
    G["extreme_products"] = ext
    # --- one random full-range product per BASELINE N
    rnd = []
    for N in (512, 1024, 2048):
        rng = random.Random(1000 + N)
        a, b_ = rand_full(rng, N), rand_full(rng, N)
        rnd.append(dict(N=N, a=a, b=b_, out=pmul(a, b_)))
    G["random_products"] = rnd
    # --- protocol tuples, N = 16 (tests/test.rs:8), default (n,k,l) = (1,3,1) and a (2,5,2) shape
    P = Prm(16)
    P2 = Prm(16, n=2, k=5, l=2)
    G["open"] = [open_case(P, 11), open_case(P, 12, tamper=True), open_case(P2, 13)]
    G["linear"] = [linear_case(P, 21), linear_case(P, 22, tamper=True), linear_case(P2, 23)]
    G["sum"] = [sum_case(P, 4, 31), sum_case(P, 4, 32, tamper=True), sum_case(P2, 3, 33)]
    # --- one protocol tuple per BASELINE N for Open (kept small: default shape)
    G["open_big"] = [open_case(Prm(512), 41), open_case(Prm(1024), 42)]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(out, "w") as f:
        json.dump(G, f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
