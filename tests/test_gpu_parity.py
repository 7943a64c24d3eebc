"""GPU parity tests: the HIP path (through the C ABI, include/rzk.h) against the CPU oracle.

Bar: bit-exact (integer arithmetic).  Inputs are seeded; sizes are chosen so the schoolbook oracle
finishes in seconds.  Full BASELINE sizes are covered by size-independent properties in
tests/test_gpu_properties.py.  Shapes follow the reference's tests (tests/test.rs: fresh key,
commit -> Commitment::verify -> challenge -> response -> verify) at the ring degrees the kernels
support (512 / 1024 / 2048; the reference's N=16 cases are pinned on the oracle instead).
"""
import numpy as np
import pytest

from oracle import oracle as O
from ring_zk_amd import synth

pytestmark = pytest.mark.gpu

Q = O.Q_DEFAULT
HALF = (Q - 1) // 2


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


_ctx_cache = {}


def ctx_for(N, n=1, k=3, l=1):
    from ring_zk_amd import Context

    key = (N, n, k, l)
    if key not in _ctx_cache:
        _ctx_cache[key] = Context(N, n, k, l)
    return _ctx_cache[key]


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---- transforms -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_ntt_forward_inverse_vs_oracle(torch_mod, N):
    ctx = ctx_for(N)
    perm = ctx.ntt_layout()
    assert sorted(perm.tolist()) == list(range(N))
    rng = np.random.default_rng(N)
    for prime in range(3):
        p = ctx.ntt_prime(prime)
        psi = ctx.ntt_psi(prime)
        assert O.powmod(psi, N, p) == p - 1
        x = rng.integers(0, p, (5, N), dtype=np.uint32)   # 5: not a multiple of the 4 waves per block
        x[0, :3] = [0, p - 1, 1]
        f = ctx.ntt_forward(prime, x)                      # host-pointer path
        for i in range(x.shape[0]):
            ref = O.ntt_forward(x[i], p, psi)
            assert np.array_equal(f[i][perm], ref)
        assert np.array_equal(ctx.ntt_inverse(prime, f), x)
        # device-pointer path
        xd = dev(torch_mod, x.view(np.int32))
        fd = ctx.ntt_forward(prime, xd)
        assert np.array_equal(fd.cpu().numpy().view(np.uint32), f)
        assert np.array_equal(ctx.ntt_inverse(prime, fd).cpu().numpy().view(np.uint32), x)


# ---- Polynomial::mul ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_polymul_full_range_and_golden(torch_mod, N, golden):
    ctx = ctx_for(N)
    rng = np.random.default_rng(7 * N)
    cases = [c for c in golden["extreme_products"] + golden["random_products"] if c["N"] == N]
    a = np.array([c["a"] for c in cases] + [synth.uniform(rng, N).tolist() for _ in range(3)], dtype=np.int64)
    b = np.array([c["b"] for c in cases] + [synth.uniform(rng, N).tolist() for _ in range(3)], dtype=np.int64)
    out = ctx.polymul(a, b)
    for i, c in enumerate(cases):
        assert out[i].tolist() == c["out"], c.get("name", "random")
    for i in range(len(cases), a.shape[0]):
        assert np.array_equal(out[i], O.poly_mul(a[i], b[i]))
    outd = ctx.polymul(dev(torch_mod, a), dev(torch_mod, b)).cpu().numpy()
    assert np.array_equal(outd, out)


def test_polymul_small_operands_pick_fewer_primes(torch_mod):
    """Operand-norm dependent prime count: ternary x sparse (1 prime), key x ternary / gaussian (2)."""
    N = 1024
    ctx = ctx_for(N)
    rng = np.random.default_rng(3)
    r = synth.small(rng, (4, N))
    d = synth.challenge(rng, (4,), N, 36)
    kfull = synth.uniform(rng, (4, N))
    y = synth.gauss(rng, (4, N), 21780)
    for a, b in ((r, d), (d, r), (kfull, r), (kfull, y), (y, kfull), (kfull, d)):
        out = ctx.polymul(a, b)
        for i in range(4):
            assert np.array_equal(out[i], O.poly_mul(a[i], b[i]))
    z = np.zeros((1, N), dtype=np.int64)
    assert not ctx.polymul(z, kfull[:1]).any()


def test_empty_batch(torch_mod):
    ctx = ctx_for(512)
    e = np.empty((0, 512), dtype=np.int64)
    assert ctx.polymul(e, e).shape == (0, 512)
    assert ctx.add(e, e).shape == (0, 512)
    # phase-level and commitment entry points with no proofs: nothing launched, empty results
    rng = np.random.default_rng(0)
    ctx.load_key(synth.key(rng, 512, 1, 3, 1))
    x0 = np.empty((0, 1, 512), dtype=np.int64)
    v0 = np.empty((0, 3, 512), dtype=np.int64)
    c0, t0, ok0 = ctx.open_commit(x0, v0, v0)
    assert c0.shape == (0, 2, 512) and t0.shape == (0, 1, 512) and ok0.shape == (0,)
    assert ctx.open_response(v0, v0, e).shape == (0, 3, 512)
    assert ctx.open_verify(v0, t0, c0, e).shape == (0,)
    cm0, okc0 = ctx.commit(x0, v0)
    assert cm0.shape == (0, 2, 512) and ctx.commitment_verify(cm0, x0, v0).shape == (0,)
    assert ctx.sample_challenge(1, 0, (0,)).shape == (0, 512)


# ---- Mat seam ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(512, 1, 3, 1), (1024, 1, 3, 1), (512, 2, 5, 2), (2048, 1, 3, 1)])
def test_matvec_cmul_add_sub_norm_eq(torch_mod, shape):
    N, n, k, l = shape
    ctx = ctx_for(N, n, k, l)
    rng = np.random.default_rng(sum(shape))
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 3
    v = synth.uniform(rng, (B, k, N))
    v[1] = synth.small(rng, (k, N))
    for which, rows, sl in ((0, n, slice(0, n)), (1, l, slice(n, n + l)), (2, n + l, slice(0, n + l))):
        addend = synth.uniform(rng, (B, rows, N))
        out = ctx.matvec(which, v)
        out_add = ctx.matvec(which, v, addend)
        for b in range(B):
            ref = O.mat_dot(A[sl], v[b][:, None, :])[:, 0, :]
            assert np.array_equal(out[b], ref)
            assert np.array_equal(out_add[b], O.mat_add(ref[:, None, :], addend[b][:, None, :])[:, 0, :])
    m = synth.uniform(rng, (B, k, N))
    p = synth.uniform(rng, (B, N))
    cm = ctx.cmul(m, p)
    for b in range(B):
        assert np.array_equal(cm[b], O.mat_cmul(m[b][:, None, :], p[b])[:, 0, :])
    a1, a2 = synth.uniform(rng, (B, k, N)), synth.uniform(rng, (B, k, N))
    a1[0, 0, :4] = [HALF, -HALF, HALF, 0]
    a2[0, 0, :4] = [HALF, -HALF, 1, 0]
    s, df = ctx.add(a1, a2), ctx.sub(a1, a2)
    for b in range(B):
        assert np.array_equal(s[b], O.mat_add(a1[b][:, None], a2[b][:, None])[:, 0])
        assert np.array_equal(df[b], O.mat_sub(a1[b][:, None], a2[b][:, None])[:, 0])
    # norm predicate: floor(sqrt(sum c^2)) <= bound, exact at the boundary
    y = synth.gauss(rng, (B, k, N), 1000)
    nb = max(O.norm2(y[0, j]) for j in range(k))
    assert ctx.norm2_le(y[:1], nb).tolist() == [1]
    assert ctx.norm2_le(y[:1], nb - 1).tolist() == [0]
    big = synth.uniform(rng, (2, k, N))
    exp = [int(O.check_norm(big[i], 2 ** 32 - 1)) for i in range(2)]
    assert ctx.norm2_le(big, 2 ** 32 - 1).tolist() == exp
    kat = np.zeros((1, 1, N), dtype=np.int64)
    kat[0, 0, :4] = [1, -2, 3, -4]                       # polynomial.rs:111-115: norm_2 = 5
    assert ctx.norm2_le(kat, 5).tolist() == [1] and ctx.norm2_le(kat, 4).tolist() == [0]
    # equality
    e2 = a1.copy()
    e2[1, k - 1, N - 1] ^= 1
    assert ctx.eq(a1, e2).tolist() == [1, 0, 1]
    # device-pointer path gives the same bytes
    assert np.array_equal(ctx.matvec(2, dev(torch_mod, v)).cpu().numpy(), ctx.matvec(2, v))


def test_shape_mismatch_is_an_error(torch_mod):
    ctx = ctx_for(512)
    a = np.zeros((2, 3, 512), dtype=np.int64)
    b = np.zeros((2, 2, 512), dtype=np.int64)
    with pytest.raises(ValueError):
        ctx.add(a, b)                      # Mat::add panics (mat.rs:129-130)
    with pytest.raises(ValueError):
        ctx.matvec(0, b)                   # Mat::dot panics (mat.rs:103)
    with pytest.raises(ValueError):
        ctx.load_key(np.zeros((1, 3, 512), dtype=np.int64))


# ---- OpenProof ----------------------------------------------------------------------------------------------------------
def _P(ctx):
    return O.Params(N=ctx.N, n=ctx.n, k=ctx.k, l=ctx.l, kappa=ctx.kappa, b=ctx.b)


def test_open_golden_fixtures(torch_mod, golden):
    for g in golden["open_big"]:
        pr = g["params"]
        ctx = ctx_for(pr["N"], pr["n"], pr["k"], pr["l"])
        arr = lambda name: np.array(g[name], dtype=np.int64)
        ctx.load_key(arr("A"))
        c, t, ok = ctx.open_commit(arr("x")[None], arr("r")[None], arr("y")[None])
        assert c[0].tolist() == g["c"] and t[0].tolist() == g["t"] and bool(ok[0]) == g["commit_ok"]
        z = ctx.open_response(arr("y")[None], arr("r")[None], arr("d")[None])
        assert z[0].tolist() == g["z"]
        acc = ctx.open_verify(z, t, c, arr("d")[None])
        assert bool(acc[0]) == g["accept"]


@pytest.mark.parametrize("shape", [(512, 1, 3, 1), (1024, 1, 3, 1), (512, 2, 5, 2), (2048, 1, 3, 1)])
def test_open_cycle_vs_oracle(torch_mod, shape):
    N, n, k, l = shape
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    assert (ctx.sigma, ctx.commit_bound, ctx.verify_bound) == (P.sigma, P.commit_bound, P.verify_bound)
    rng = np.random.default_rng(11 + sum(shape))
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 5 if N <= 1024 else 3
    x = synth.uniform(rng, (B, l, N))
    r = synth.small(rng, (B, k, N))
    y = synth.gauss(rng, (B, k, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)
    r[1] = synth.uniform(rng, (k, N))          # breaks the commit constraint -> ok = 0
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    zt = z.copy()
    zt[2, k - 1, 5] = O.center(int(zt[2, k - 1, 5]) + 1)      # tampered response -> reject
    y_big = y.copy()
    acc = ctx.open_verify(zt, t, c, d)
    for b in range(B):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[b], r[b], y[b])
        assert bool(ok[b]) == ok_ref
        assert np.array_equal(t[b], t_ref)
        assert np.array_equal(c[b], c_ref)
        assert np.array_equal(z[b], O.open_response(P, y[b], r[b], d[b]))
        assert int(acc[b]) == int(O.open_verify(P, A, zt[b], t[b], c[b], d[b]) == 1)
    assert acc.tolist()[0] == 1 and acc.tolist()[2] == 0
    # same call with device pointers (torch tensors): identical bytes
    cd, td, okd = ctx.open_commit(dev(torch_mod, x), dev(torch_mod, r), dev(torch_mod, y))
    assert np.array_equal(cd.cpu().numpy(), c) and np.array_equal(td.cpu().numpy(), t)
    assert np.array_equal(okd.cpu().numpy(), ok)
    accd = ctx.open_verify(dev(torch_mod, zt), td, cd, dev(torch_mod, d))
    assert np.array_equal(accd.cpu().numpy(), acc)
    del y_big


# ---- LinearProof ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(512, 1, 3, 1), (1024, 1, 3, 1), (512, 2, 5, 2)])
def test_linear_cycle_vs_oracle(torch_mod, shape):
    N, n, k, l = shape
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    rng = np.random.default_rng(21 + sum(shape))
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 3
    g = synth.uniform(rng, (B, N))
    x = synth.uniform(rng, (B, l, N))
    r, rp = synth.small(rng, (B, k, N)), synth.small(rng, (B, k, N))
    y, yp = synth.gauss(rng, (B, k, N), P.sigma), synth.gauss(rng, (B, k, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)
    rp[2] = synth.uniform(rng, (k, N))
    c, cp, t, tp, u, ok = ctx.linear_commit(g, x, r, rp, y, yp)
    z, zp = ctx.linear_response(y, yp, r, rp, d)
    zpt = zp.copy()
    zpt[1, 0, 0] = O.center(int(zpt[1, 0, 0]) - 1)
    acc = ctx.linear_verify(z, zpt, c, cp, g, t, tp, u, d)
    for b in range(B):
        ref = O.linear_commit(P, A, g[b], x[b], r[b], rp[b], y[b], yp[b])
        for got, want, name in zip((c, cp, t, tp, u), ref[:5], ("c", "cp", "t", "tp", "u")):
            assert np.array_equal(got[b], want), name
        assert int(ok[b]) == ref[5]
        zr, zpr = O.linear_response(P, y[b], yp[b], r[b], rp[b], d[b])
        assert np.array_equal(z[b], zr) and np.array_equal(zp[b], zpr)
        want = O.linear_verify(P, A, z[b], zpt[b], c[b], cp[b], g[b], t[b], tp[b], u[b], d[b])
        assert int(acc[b]) == int(want == 1)
    assert acc.tolist()[0] == 1 and acc.tolist()[1] == 0


# ---- SumProof ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(512, 1, 3, 1, 4), (1024, 1, 3, 1, 2), (512, 2, 5, 2, 3)])
def test_sum_cycle_vs_oracle(torch_mod, shape):
    N, n, k, l, V = shape
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    rng = np.random.default_rng(31 + sum(shape))
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 2
    gs = synth.uniform(rng, (B, V, N))
    xs = synth.uniform(rng, (B, V, l, N))
    rs, rp = synth.small(rng, (B, V, k, N)), synth.small(rng, (B, k, N))
    ys, yp = synth.gauss(rng, (B, V, k, N), P.sigma), synth.gauss(rng, (B, k, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)
    cs, cp, ts, tp, u, ok = ctx.sum_commit(gs, xs, rs, rp, ys, yp)
    zs, zp = ctx.sum_response(ys, yp, rs, rp, d)
    zst = zs.copy()
    zst[1, V - 1, k - 1, 7] = O.center(int(zst[1, V - 1, k - 1, 7]) + 2)
    acc = ctx.sum_verify(zst, zp, cs, cp, gs, ts, tp, u, d)
    for b in range(B):
        ref = O.sum_commit(P, A, gs[b], xs[b], rs[b], rp[b], ys[b], yp[b])
        for got, want, name in zip((cs, cp, ts, tp, u), ref[:5], ("cs", "cp", "ts", "tp", "u")):
            assert np.array_equal(got[b], want), name
        assert bool(ok[b]) == ref[5]
        zr, zpr = O.sum_response(P, ys[b], yp[b], rs[b], rp[b], d[b])
        assert np.array_equal(zs[b], zr) and np.array_equal(zp[b], zpr)
        want = O.sum_verify(P, A, zst[b], zp[b], cs[b], cp[b], gs[b], ts[b], tp[b], u[b], d[b])
        assert int(acc[b]) == int(want == 1)
    assert acc.tolist() == [1, 0]


def _four_squares(m):
    """a^2+b^2+c^2+d^2 = m by search (Lagrange)."""
    import math

    a = math.isqrt(m)
    while a >= 0:
        r1 = m - a * a
        b = math.isqrt(r1)
        while b >= 0:
            r2 = r1 - b * b
            c = math.isqrt(r2)
            while c >= 0 and c * c * 2 >= r2:
                r3 = r2 - c * c
                d = math.isqrt(r3)
                if d * d == r3:
                    return a, b, c, d
                c -= 1
            b -= 1
            if (r1 - b * b) > 2 * (math.isqrt(r1) + 1) ** 2:
                break
        a -= 1
    raise AssertionError("no decomposition")


@pytest.mark.parametrize("N", [512, 1024])
def test_fused_norm_predicate_exact_at_the_boundary(torch_mod, N):
    """check_commit_constraint is fused into the commit rows: floor(sqrt(sum r^2)) <= bound must flip exactly
    between sum = (bound+1)^2 - 1 and (bound+1)^2, for whichever polynomial of r carries the weight."""
    ctx = ctx_for(N)
    rng = np.random.default_rng(N + 1)
    A = synth.key(rng, N, 1, 3, 1)
    ctx.load_key(A)
    Bd = ctx.commit_bound
    a, b, c, d = _four_squares(2 * Bd)          # (Bd+1)^2 - 1 = Bd^2 + 2*Bd
    cases = []
    for col in range(3):                        # weight in r0 (identity column: an addition), r1, r2 (products)
        r_ok = np.zeros((3, N), dtype=np.int64)
        r_ok[col, :5] = [Bd, a, b, c, d]
        assert O.norm2(r_ok[col]) == Bd and int((r_ok[col].astype(object) ** 2).sum()) == (Bd + 1) ** 2 - 1
        r_bad = r_ok.copy()
        r_bad[col, 7] = 1                       # sum = (Bd+1)^2 -> norm_2 = Bd + 1
        assert O.norm2(r_bad[col]) == Bd + 1
        r_huge = np.zeros((3, N), dtype=np.int64)
        r_huge[col, 3] = (1 << 24) + 5          # above the clamp of the fused sum
        cases += [(r_ok, 1), (r_bad, 0), (r_huge, 0)]
    r = np.stack([cse[0] for cse in cases])
    Bn = r.shape[0]
    x = synth.uniform(rng, (Bn, 1, N))
    y = synth.gauss(rng, (Bn, 3, N), ctx.sigma)
    c_, t_, ok = ctx.open_commit(x, r, y)
    assert ok.tolist() == [cse[1] for cse in cases]
    P = _P(ctx)
    for i in range(Bn):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[i], r[i], y[i])
        assert bool(ok[i]) == ok_ref
        assert np.array_equal(c_[i], c_ref) and np.array_equal(t_[i], t_ref)   # products stay exact either way
    # the same predicate with the verify bound, fused into the verifier rows
    Vb = ctx.verify_bound
    a, b, c, d = _four_squares(2 * Vb)
    z = np.zeros((2, 3, N), dtype=np.int64)
    z[:, 2, :5] = [Vb, a, b, c, d]
    z[1, 2, 9] = -1
    tt = np.stack([O.mat_dot(A[:1], z[i][:, None, :])[:, 0, :] for i in range(2)])   # t := a1.z, c1 = 0 -> relation holds
    cz = np.zeros((2, 2, N), dtype=np.int64)
    dz = synth.challenge(rng, (2,), N, 36)
    assert ctx.open_verify(z, tt, cz, dz).tolist() == [1, 0]
    assert [int(O.open_verify(P, A, z[i], tt[i], cz[i], dz[i]) == 1) for i in range(2)] == [1, 0]


@pytest.mark.parametrize("kind", ["dense", "zero_column", "ones_everywhere"])
def test_unstructured_keys(torch_mod, kind):
    """The key loader classifies entries (0 / 1 / general) instead of assuming [I | A']: a fully dense key, a key
    whose random part has a zero column (that column of r is then never loaded by a product, so the fused norm
    predicate must fall back to the norm kernel) and a key made only of 0/1 entries must all match the oracle."""
    N, n, k, l = 512, 2, 5, 2
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    rng = np.random.default_rng({"dense": 1, "zero_column": 2, "ones_everywhere": 3}[kind])
    if kind == "dense":
        A = synth.uniform(rng, (n + l, k, N))
    elif kind == "zero_column":
        A = synth.key(rng, N, n, k, l)
        A[:, k - 1, :] = 0                      # last column multiplies nothing
        A[0, 0, :] = 0                          # and row 0 loses its identity entry
    else:
        A = np.zeros((n + l, k, N), dtype=np.int64)
        A[:, :, 0] = rng.integers(0, 2, (n + l, k))
    ctx.load_key(A)
    B = 4
    x = synth.uniform(rng, (B, l, N))
    r = synth.small(rng, (B, k, N))
    r[1, k - 1] = synth.uniform(rng, N)         # breaks the constraint through the never-multiplied column
    y = synth.gauss(rng, (B, k, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    z[2, k - 1, :] = P.verify_bound             # norm violation in the last column
    acc = ctx.open_verify(z, t, c, d)
    for b in range(B):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[b], r[b], y[b])
        assert np.array_equal(c[b], c_ref) and np.array_equal(t[b], t_ref) and bool(ok[b]) == ok_ref
        assert int(acc[b]) == int(O.open_verify(P, A, z[b], t[b], c[b], d[b]) == 1)
    assert ok.tolist()[1] == 0 and acc.tolist()[2] == 0
    v = synth.uniform(rng, (B, k, N))
    mv = ctx.matvec(2, v)
    for b in range(B):
        assert np.array_equal(mv[b], O.mat_dot(A, v[b][:, None, :])[:, 0, :])
    ctx.load_key(synth.key(rng, N, n, k, l))    # restore a regular key for the cached context


def test_cmul_with_more_rows_than_a_program_holds(torch_mod):
    N = 512
    ctx = ctx_for(N)
    rng = np.random.default_rng(50)
    rows = 50                                    # > kMaxRows (48): rows become batch entries sharing p
    m = synth.uniform(rng, (2, rows, N))
    p = synth.uniform(rng, (2, N))
    out = ctx.cmul(m, p)
    for b in range(2):
        for i in (0, 17, 49):
            assert np.array_equal(out[b, i], O.poly_mul(m[b, i], p[b]))


# ---- challenge products (shift-add kernel): every accumulation width, any multiplier ------------------------
@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_challenge_products_any_multiplier(torch_mod, N):
    """z = y + r (.) d and a1.z == t + c1 (.) d with multipliers that are NOT kappa-sparse +-1: the
    shift-add path must stay exact (32-bit sums, 64-bit sums, two 16-bit passes), as the reference is
    for any polynomial d (src/prove/open.rs:113-115, 172)."""
    n, k, l = 1, 3, 1
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    rng = np.random.default_rng(900 + N)
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 6
    d = synth.challenge(rng, (B,), N, P.kappa)
    d[0, :] = 0                                                 # zero multiplier
    d[1] = synth.challenge(rng, (1,), N, P.kappa)[0] * 3        # sparse, coefficients +-3 -> 64-bit sums
    d[2] = synth.uniform(rng, (N,))                             # dense full range -> 16-bit passes
    d[3, :] = 0
    d[3, [0, 1, N - 2, N - 1]] = [1, -1, 1, -1]                 # rotations by 0, 1, N-2, N-1
    d[4] = rng.integers(-1, 2, N)                               # dense ternary
    y = synth.gauss(rng, (B, k, N), P.sigma)
    r = synth.small(rng, (B, k, N))
    r[2] = synth.uniform(rng, (k, N))                           # full-range r times full-range d
    r[5] = synth.uniform(rng, (k, N))                           # full-range r times a real challenge
    r[2, 0, :2] = [HALF, -HALF]
    d[2, :2] = [-HALF, HALF]
    z = ctx.open_response(y, r, d)
    for b in range(B):
        assert np.array_equal(z[b], O.open_response(P, y[b], r[b], d[b])), b
    # verification equation with the same multipliers; t is chosen so that every proof satisfies it
    zs = synth.gauss(rng, (B, k, N), P.sigma // 4)
    c = synth.uniform(rng, (B, n + l, N))
    t = np.empty((B, n, N), dtype=np.int64)
    for b in range(B):
        lhs = O.mat_dot(A[:n], zs[b][:, None, :])
        t[b] = O.mat_sub(lhs, O.mat_cmul(c[b, :n][:, None, :], d[b]))[:, 0, :]
    acc = ctx.open_verify(zs, t, c, d)
    assert acc.tolist() == [1] * B
    t[4, 0, 7] = O.center(int(t[4, 0, 7]) + 1)
    t[2, 0, N - 1] = O.center(int(t[2, 0, N - 1]) - 1)
    acc = ctx.open_verify(zs, t, c, d)
    assert acc.tolist() == [1, 1, 0, 1, 0, 1]
    for b in range(B):
        assert int(acc[b]) == int(O.open_verify(P, A, zs[b], t[b], c[b], d[b]) == 1)


# ---- commitment scheme (src/commit.rs) ------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(512, 1, 3, 1), (1024, 1, 3, 1), (512, 2, 5, 2)])
def test_commit_and_commitment_verify(torch_mod, shape):
    """CommitmentKey::commit and Commitment::verify incl. the f = Some(_) branch (commit.rs:88-128, 173-210);
    mirrors the reference's doctest (commit.rs:152-171): a fresh commitment verifies, a wrong message or
    wrong randomness does not."""
    N, n, k, l = shape
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    rng = np.random.default_rng(4242 + sum(shape))
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 6
    x = synth.uniform(rng, (B, l, N))
    r = synth.small(rng, (B, k, N))
    r[4] = synth.uniform(rng, (k, N))                       # violates the commit constraint
    c, ok = ctx.commit(x, r)
    for b in range(B):
        c_ref, ok_ref = O.commit(P, A, x[b], r[b])
        assert np.array_equal(c[b], c_ref) and bool(ok[b]) == ok_ref
    assert ok.tolist() == [1, 1, 1, 1, 0, 1]
    # f = None
    x_bad = x.copy()
    x_bad[1, 0, 3] = O.center(int(x_bad[1, 0, 3]) + 1)      # wrong message
    r_bad = r.copy()
    r_bad[2, k - 1, 0] = O.center(int(r_bad[2, k - 1, 0]) + 1)   # wrong randomness (still small)
    for xs, rs in ((x, r), (x_bad, r), (x, r_bad)):
        got = ctx.commitment_verify(c, xs, rs)
        want = [int(O.commitment_verify(P, A, c[b], xs[b], rs[b])) for b in range(B)]
        assert got.tolist() == want
    assert ctx.commitment_verify(c, x, r).tolist() == [1, 1, 1, 1, 0, 1]
    assert ctx.commitment_verify(c, x_bad, r).tolist() == [1, 0, 1, 1, 0, 1]
    # f = Some(f): f * c == a.r' + f * [0;x] holds for r' = f * r; use a small f so that r' stays short
    f = np.zeros((B, N), dtype=np.int64)
    f[:, 0] = 2
    f[3, 5] = -1
    rf = np.stack([O.mat_cmul(r[b][:, None, :], f[b])[:, 0, :] for b in range(B)])
    got = ctx.commitment_verify(c, x, rf, f)
    want = [int(O.commitment_verify(P, A, c[b], x[b], rf[b], f[b])) for b in range(B)]
    assert got.tolist() == want and got.tolist()[:4] == [1, 1, 1, 1] and got.tolist()[4] == 0
    got = ctx.commitment_verify(c, x_bad, rf, f)
    want = [int(O.commitment_verify(P, A, c[b], x_bad[b], rf[b], f[b])) for b in range(B)]
    assert got.tolist() == want and got.tolist()[1] == 0
    ff = synth.uniform(rng, (B, N))                          # full-range f with the unscaled r: must reject
    got = ctx.commitment_verify(c, x, r, ff)
    want = [int(O.commitment_verify(P, A, c[b], x[b], r[b], ff[b])) for b in range(B)]
    assert got.tolist() == want
    # device pointers: identical results
    gd = ctx.commitment_verify(dev(torch_mod, c), dev(torch_mod, x), dev(torch_mod, rf), dev(torch_mod, f))
    assert gd.cpu().numpy().tolist() == ctx.commitment_verify(c, x, rf, f).tolist()
    cd, okd = ctx.commit(dev(torch_mod, x), dev(torch_mod, r))
    assert np.array_equal(cd.cpu().numpy(), c) and np.array_equal(okd.cpu().numpy(), ok)


@pytest.mark.parametrize("N", [512, 1024])
def test_linear_commit_two_bit_verdicts(torch_mod, N):
    """ok[b] of the Linear commit: bit 0 = constraint(r), bit 1 = constraint(r') (the two ck.commit calls of
    linear.rs:96-97).  With a word-aligned flag array of a multiple of 4 proofs both predicates ride on the commit
    rows (atomic clears of single bits); all four bit patterns must come out, next to each other in one word."""
    n, k, l = 1, 3, 1
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    rng = np.random.default_rng(77 + N)
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 8
    g = synth.uniform(rng, (B, N))
    x = synth.uniform(rng, (B, l, N))
    r, rp = synth.small(rng, (B, k, N)), synth.small(rng, (B, k, N))
    y, yp = synth.gauss(rng, (B, k, N), P.sigma), synth.gauss(rng, (B, k, N), P.sigma)
    big = synth.uniform(rng, (k, N))
    r[1], rp[2], r[3], rp[3] = big, big, big, big            # patterns 2, 1, 0 in one flag word
    r[6, k - 1] = big[0]                                     # only the LAST polynomial of r violates the bound
    rp[7, 0] = big[1]                                        # only the FIRST polynomial of r'
    want_ok = [3, 2, 1, 0, 3, 3, 2, 1]
    for conv in (lambda a: a, lambda a: dev(torch_mod, a)):
        out = ctx.linear_commit(*(conv(a) for a in (g, x, r, rp, y, yp)))
        c, cp, t, tp, u, ok = (o if isinstance(o, np.ndarray) else o.cpu().numpy() for o in out)
        assert ok.tolist() == want_ok
        for b in range(B):
            ref = O.linear_commit(P, A, g[b], x[b], r[b], rp[b], y[b], yp[b])
            for got, want in zip((c, cp, t, tp, u), ref[:5]):
                assert np.array_equal(got[b], want)
            assert int(ok[b]) == ref[5]
    # a batch that is not a multiple of 4 takes the separate norm kernels: same verdicts
    out = ctx.linear_commit(g[:7], x[:7], r[:7], rp[:7], y[:7], yp[:7])
    assert out[5].tolist() == want_ok[:7]


# ---- row blocks (operands of a block staged once in LDS, 8-wave workgroups) ----------------------------------
@pytest.mark.parametrize("N", [1024, 2048])
def test_row_blocks_vs_oracle(torch_mod, N, monkeypatch):
    """Key-only programs on the row-block kernel (default from N = 2048 on; forced at N = 1024 here): commit,
    matrix-vector products, commitment verification and the two-step A1 relation, bit-exact against the oracle,
    including a failing norm predicate and a batch larger than the resident grid would need."""
    from ring_zk_amd import Context

    monkeypatch.setenv("RZK_BLOCK_MIN_LOGN", "10")
    n, k, l = 2, 5, 2
    ctx = Context(N, n, k, l)                       # fresh context: the knob is read at creation
    P = _P(ctx)
    rng = np.random.default_rng(5150 + N)
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 4
    x = synth.uniform(rng, (B, l, N))
    r = synth.small(rng, (B, k, N))
    y = synth.gauss(rng, (B, k, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)
    r[2, k - 1] = synth.uniform(rng, (N,))          # commit constraint fails for proof 2
    c, t, ok = ctx.open_commit(x, r, y)
    for b in range(B):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[b], r[b], y[b])
        assert np.array_equal(c[b], c_ref) and np.array_equal(t[b], t_ref) and bool(ok[b]) == ok_ref
    assert ok.tolist() == [1, 1, 0, 1]
    for which, rows in ((0, slice(0, n)), (1, slice(n, n + l)), (2, slice(0, n + l))):
        got = ctx.matvec(which, y)
        for b in range(B):
            assert np.array_equal(got[b], O.mat_dot(A[rows], y[b][:, None, :])[:, 0, :])
    cm, okc = ctx.commit(x, r)
    assert np.array_equal(cm, c) and okc.tolist() == ok.tolist()
    assert ctx.commitment_verify(c, x, r).tolist() == [1, 1, 0, 1]
    z = ctx.open_response(y, r, d)
    zt = z.copy()
    zt[1, 0, 3] = O.center(int(zt[1, 0, 3]) + 1)
    acc = ctx.open_verify(zt, t, c, d)
    for b in range(B):
        assert np.array_equal(z[b], O.open_response(P, y[b], r[b], d[b]))
        assert int(acc[b]) == int(O.open_verify(P, A, zt[b], t[b], c[b], d[b]) == 1)
    assert acc.tolist()[0] == 1 and acc.tolist()[1] == 0
    # many proofs: more tasks than resident workgroups, every proof checked through a size-independent property
    Bb = 700
    yb = synth.gauss(rng, (Bb, k, N), P.sigma)
    t1 = ctx.matvec(0, yb)
    t2 = ctx.matvec(0, 2 * yb)                      # linearity: a1.(2y) = 2 (a1.y) mod q
    twice = (2 * t1.astype(object)) % Q
    twice = np.where(twice > HALF, twice - Q, twice).astype(np.int64)
    assert np.array_equal(t2, twice)
    assert np.array_equal(t1[-1], O.mat_dot(A[:n], yb[-1][:, None, :])[:, 0, :])


def test_device_buffers_at_8_byte_alignment(torch_mod):
    """The C ABI takes int64*: device buffers need no more than 8-byte alignment, although the rotation kernel
    and the stores use 16-byte accesses (global memory on this target tolerates them unaligned)."""
    N, n, k, l = 1024, 1, 3, 1
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    rng = np.random.default_rng(1234)
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 3
    x = synth.uniform(rng, (B, l, N))
    r = synth.small(rng, (B, k, N))
    y = synth.gauss(rng, (B, k, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)

    def odd(a):
        """copy of `a` on the device that starts one int64 past a 256-byte aligned allocation"""
        buf = torch_mod.empty(a.size + 1, dtype=torch_mod.int64, device="cuda")
        view = buf[1:].view(a.shape)
        view.copy_(torch_mod.from_numpy(a))
        assert view.data_ptr() % 16 == 8 and view.is_contiguous()
        return view

    c, t, ok = ctx.open_commit(odd(x), odd(r), odd(y))
    z = ctx.open_response(odd(y), odd(r), odd(d))
    zo = odd(z.cpu().numpy())
    acc = ctx.open_verify(zo, odd(t.cpu().numpy()), odd(c.cpu().numpy()), odd(d))
    for b in range(B):
        c_ref, t_ref, _ = O.open_commit(P, A, x[b], r[b], y[b])
        assert np.array_equal(c[b].cpu().numpy(), c_ref) and np.array_equal(t[b].cpu().numpy(), t_ref)
        assert np.array_equal(z[b].cpu().numpy(), O.open_response(P, y[b], r[b], d[b]))
    assert acc.cpu().numpy().tolist() == [1] * B


# ---- other parameter sets: the reference is generic in Q, kappa and b (src/params.rs:18-36) -------------------------
@pytest.mark.parametrize("q", [1073741827, 4294606851])
@pytest.mark.parametrize("N", [512, 1024])
def test_other_moduli_and_parameters(torch_mod, N, q):
    """Smallest and largest supported ring moduli (just above 2^30; just below 4 p2, include/rzk.h) with kappa = 60
    (the reference's own challenge-set test, challenge_space.rs:60) and b = 3: every mod-q path of the kernels
    (32-bit add/sub with wrap-around, 64-bit reductions of the rotation sums, CRT constants) against the oracle."""
    from ring_zk_amd import Context

    n, k, l, kappa, b = 1, 3, 1, 60, 3
    ctx = Context(N, n, k, l, kappa=kappa, b=b, q=q)
    P = O.Params(N, n, k, l, kappa, b, q)
    assert (ctx.sigma, ctx.commit_bound, ctx.verify_bound) == (P.sigma, P.commit_bound, P.verify_bound)
    half = (q - 1) // 2
    rng = np.random.default_rng(q % 1000 + N)
    A = O.key_build(P, rng.integers(-half, half + 1, (n, k - n, N)), rng.integers(-half, half + 1, (l, k - n - l, N)))
    ctx.load_key(A)
    B = 4
    x = rng.integers(-half, half + 1, (B, l, N))
    r = rng.integers(-b, b + 1, (B, k, N))
    y = np.trunc(rng.normal(0, P.sigma, (B, k, N))).astype(np.int64)
    d = np.zeros((B, N), dtype=np.int64)
    for i in range(B):
        pos = rng.choice(N, kappa, replace=False)
        d[i, pos] = rng.choice([-1, 1], kappa)
    x[0, 0, :4] = [half, -half, half, -half]              # extreme coefficients
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    zt = z.copy()
    zt[3, 1, 7] = O.center(int(zt[3, 1, 7]) + 1, q)
    acc = ctx.open_verify(zt, t, c, d)
    for i in range(B):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[i], r[i], y[i])
        assert np.array_equal(c[i], c_ref) and np.array_equal(t[i], t_ref) and bool(ok[i]) == ok_ref
        assert np.array_equal(z[i], O.open_response(P, y[i], r[i], d[i]))
        assert int(acc[i]) == int(O.open_verify(P, A, zt[i], t[i], c[i], d[i]) == 1)
    assert acc.tolist() == [1, 1, 1, 0]
    # full-range products (three primes) and a full-range "challenge" (16-bit passes of the rotation kernel)
    a = rng.integers(-half, half + 1, (B, N))
    bb = rng.integers(-half, half + 1, (B, N))
    prod = ctx.polymul(a, bb)
    for i in range(B):
        assert np.array_equal(prod[i], O.poly_mul(a[i], bb[i], q))
    zz = ctx.open_response(y, x.repeat(k, axis=1), a)      # r := full-range, d := full-range
    for i in range(B):
        assert np.array_equal(zz[i], O.open_response(P, y[i], x.repeat(k, axis=1)[i], a[i]))
    # device-side samplers respect the modulus of the context
    u = ctx.sample_uniform(1, 0, half, (8,)).cpu().numpy()
    assert u.min() >= -half and u.max() <= half and np.abs(u).max() > 0.9 * half
    dd = ctx.sample_challenge(1, 1, (8,)).cpu().numpy()
    assert (np.abs(dd).sum(axis=1) == kappa).all()


def test_norm_bounds_beyond_32_bits(torch_mod):
    """b = 2^22 makes sigma = 11*kappa*b*floor(sqrt(kN)) and both norm bounds exceed 2^32 (params.rs:94-118 computes
    them in usize); every canonical polynomial then satisfies them and the predicates must say so."""
    from ring_zk_amd import Context

    N, n, k, l, kappa, b = 512, 1, 3, 1, 36, 1 << 22
    ctx = Context(N, n, k, l, kappa=kappa, b=b)
    P = O.Params(N, n, k, l, kappa, b)
    assert ctx.commit_bound == P.commit_bound > 1 << 32 and ctx.verify_bound == P.verify_bound > 1 << 32
    rng = np.random.default_rng(8)
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 3
    x = synth.uniform(rng, (B, l, N))
    r = rng.integers(-b, b + 1, (B, k, N))
    y = synth.uniform(rng, (B, k, N))                      # (i64) N(0, sigma) reduced mod q: spread over the whole range
    d = synth.challenge(rng, (B,), N, kappa)
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    acc = ctx.open_verify(z, t, c, d)
    for i in range(B):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[i], r[i], y[i])
        assert np.array_equal(c[i], c_ref) and np.array_equal(t[i], t_ref) and bool(ok[i]) == ok_ref == True
        assert np.array_equal(z[i], O.open_response(P, y[i], r[i], d[i]))
        assert int(acc[i]) == int(O.open_verify(P, A, z[i], t[i], c[i], d[i]) == 1) == 1
    assert ctx.norm2_le(y, P.verify_bound).tolist() == [1] * B
    assert ctx.norm2_le(y, 5).tolist() == [0] * B


def test_canonicalize(torch_mod):
    ctx = ctx_for(512)
    rng = np.random.default_rng(3)
    a = rng.integers(-2 ** 62, 2 ** 62, (5, 512), dtype=np.int64)
    a[0, :6] = [0, HALF, HALF + 1, -HALF, -HALF - 1, Q]
    a[1, :3] = [np.iinfo(np.int64).max, np.iinfo(np.int64).min, -Q]
    want = np.array([[O.center(int(v)) for v in row] for row in a], dtype=np.int64)
    assert np.array_equal(ctx.canonicalize(a), want)
    assert np.array_equal(ctx.canonicalize(dev(torch_mod, a)).cpu().numpy(), want)
    assert np.array_equal(ctx.canonicalize(want), want)      # idempotent


@pytest.mark.parametrize("nnz", [1, 255, 256, 257, 512, 2047])
def test_rotation_list_round_boundaries_n2048(torch_mod, nnz):
    """N = 2048: a two-wavefront team walks the multiplier's non-zeros from a list of 256 entries per round
    (rzk_core.h kShiftListCap); counts at and around the round boundary, split unevenly between the two wavefronts
    (positions drawn from the low half of the ring only, from the high half only, and from both), must all give the
    exact product — response rows (shift_row_kernel) and the verifier's rotation term (unit_kernel)."""
    N, n, k, l = 2048, 1, 3, 1
    ctx = ctx_for(N, n, k, l)
    P = _P(ctx)
    rng = np.random.default_rng(7000 + nnz)
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 3
    d = np.zeros((B, N), dtype=np.int64)
    pools = [np.arange(N), np.arange(0, N // 2) if nnz <= N // 2 else np.arange(N), np.arange(N // 2, N) if nnz <= N // 2 else np.arange(N)]
    for b in range(B):
        pos = rng.choice(pools[b], nnz, replace=False)
        d[b, pos] = rng.choice([-1, 1], nnz)
    y = synth.gauss(rng, (B, k, N), P.sigma)
    r = synth.small(rng, (B, k, N))
    z = ctx.open_response(y, r, d)
    for b in range(B):
        assert np.array_equal(z[b], O.open_response(P, y[b], r[b], d[b])), b
    zs = synth.gauss(rng, (B, k, N), P.sigma // 4)
    c = synth.uniform(rng, (B, n + l, N))
    t = np.empty((B, n, N), dtype=np.int64)
    for b in range(B):
        lhs = O.mat_dot(A[:n], zs[b][:, None, :])
        t[b] = O.mat_sub(lhs, O.mat_cmul(c[b, :n][:, None, :], d[b]))[:, 0, :]
    assert ctx.open_verify(zs, t, c, d).tolist() == [1] * B
    t[1, 0, N - 1] = O.center(int(t[1, 0, N - 1]) + 1)
    assert ctx.open_verify(zs, t, c, d).tolist() == [1, 0, 1]
