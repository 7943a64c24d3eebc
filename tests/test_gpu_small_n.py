"""Small ring degrees on the GPU path (N = 4 .. 256): the sizes the reference's own tests use
(src/mat.rs:241 N = 4, src/polynomial.rs:91 N = 4, tests/test.rs:8 N = 16).  The N = 16 golden protocol
tuples (tests/golden/golden.json) and the reference's literal Mat-test polynomials are checked directly
through the C ABI; random cases against the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from ring_zk_amd import synth

pytestmark = pytest.mark.gpu
Q = O.Q_DEFAULT
HALF = (Q - 1) // 2


@pytest.fixture(scope="module")
def T():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    return torch


def ctx_for(N, n=1, k=3, l=1):
    from ring_zk_amd import Context

    return Context(N, n, k, l)


def arr(x):
    return np.array(x, dtype=np.int64)


def test_reference_mat_and_norm_kats_n4(T, golden):
    ctx = ctx_for(4)
    g = golden["mat_dot_n4"]          # mat.rs:243-268: a00*b00 + a01*b10
    prod = ctx.polymul(arr(g["a"]), arr(g["b"]))
    assert ctx.add(prod[0:1], prod[1:2])[0].tolist() == g["out"] == [13, 35, 45, 30]
    g = golden["mat_cmul_n4"]         # mat.rs:389-406
    out = ctx.cmul(arr(g["a"])[None], arr(g["elem"])[None])
    assert out[0].tolist() == g["out"]
    p = arr(golden["norm_kat"]["p"])[None, None]    # polynomial.rs:111-115: norm_2([1,-2,3,-4]) = 5
    assert ctx.norm2_le(p, 5).tolist() == [1] and ctx.norm2_le(p, 4).tolist() == [0]
    with pytest.raises(Exception):
        ctx.ntt_forward(0, np.zeros((1, 4), dtype=np.uint32))   # transforms need N >= 512


@pytest.mark.parametrize("N", [4, 8, 16, 32, 64, 128, 256])
def test_polymul_matvec_vs_oracle(T, N):
    n, k, l = (1, 3, 1) if N < 64 else (2, 5, 2)
    ctx = ctx_for(N, n, k, l)
    rng = np.random.default_rng(N)
    a, b = synth.uniform(rng, (6, N)), synth.uniform(rng, (6, N))
    a[0, :] = HALF
    b[0, :] = -HALF
    out = ctx.polymul(a, b)
    for i in range(6):
        assert np.array_equal(out[i], O.poly_mul(a[i], b[i]))
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    v = synth.uniform(rng, (3, k, N))
    add = synth.uniform(rng, (3, n + l, N))
    mv = ctx.matvec(2, v, add)
    for i in range(3):
        ref = O.mat_add(O.mat_dot(A, v[i][:, None, :]), add[i][:, None, :])[:, 0, :]
        assert np.array_equal(mv[i], ref)
    e2 = v.copy()
    e2[1, 0, N - 1] ^= 1
    assert ctx.eq(v, e2).tolist() == [1, 0, 1]
    assert np.array_equal(ctx.matvec(2, T.from_numpy(v).cuda(), T.from_numpy(add).cuda()).cpu().numpy(), mv)


def _ctx_from(g):
    p = g["params"]
    return ctx_for(p["N"], p["n"], p["k"], p["l"])


def test_open_golden_n16(T, golden):
    for g in golden["open"]:
        ctx = _ctx_from(g)
        ctx.load_key(arr(g["A"]))
        c, t, ok = ctx.open_commit(arr(g["x"])[None], arr(g["r"])[None], arr(g["y"])[None])
        assert c[0].tolist() == g["c"] and t[0].tolist() == g["t"] and bool(ok[0]) == g["commit_ok"]
        if not g["tampered"]:
            assert ctx.open_response(arr(g["y"])[None], arr(g["r"])[None], arr(g["d"])[None])[0].tolist() == g["z"]
        acc = ctx.open_verify(arr(g["z"])[None], t, c, arr(g["d"])[None])
        assert bool(acc[0]) == g["accept"]


def test_linear_golden_n16(T, golden):
    for g in golden["linear"]:
        ctx = _ctx_from(g)
        ctx.load_key(arr(g["A"]))
        one = lambda name: arr(g[name])[None]
        c, cp, t, tp, u, ok = ctx.linear_commit(one("g"), one("x"), one("r"), one("rp"), one("y"), one("yp"))
        for name, val in (("c", c), ("cp", cp), ("t", t), ("tp", tp), ("u", u)):
            assert val[0].tolist() == g[name], name
        assert int(ok[0]) == g["commit_ok"]
        if not g["tampered"]:
            z, zp = ctx.linear_response(one("y"), one("yp"), one("r"), one("rp"), one("d"))
            assert z[0].tolist() == g["z"] and zp[0].tolist() == g["zp"]
        acc = ctx.linear_verify(one("z"), one("zp"), c, cp, one("g"), t, tp, u, one("d"))
        assert bool(acc[0]) == g["accept"]


def test_sum_golden_n16(T, golden):
    for g in golden["sum"]:
        ctx = _ctx_from(g)
        ctx.load_key(arr(g["A"]))
        one = lambda name: arr(g[name])[None]
        cs, cp, ts, tp, u, ok = ctx.sum_commit(one("gs"), one("xs"), one("rs"), one("rp"), one("ys"), one("yp"))
        for name, val in (("cs", cs), ("cp", cp), ("ts", ts), ("tp", tp), ("u", u)):
            assert val[0].tolist() == g[name], name
        assert bool(ok[0]) == g["commit_ok"]
        if not g["tampered"]:
            zs, zp = ctx.sum_response(one("ys"), one("yp"), one("rs"), one("rp"), one("d"))
            assert zs[0].tolist() == g["zs"] and zp[0].tolist() == g["zp"]
        acc = ctx.sum_verify(one("zs"), one("zp"), cs, cp, one("gs"), ts, tp, u, one("d"))
        assert bool(acc[0]) == g["accept"]


def test_reference_style_iterations_n16(T):
    """tests/test.rs: fresh key per iteration, N = 16, complete cycle must verify (batched here: 100 proofs)."""
    N, B = 16, 100
    ctx = ctx_for(N)
    P = O.Params(N=N)
    rng = np.random.default_rng(16)
    A = synth.key(rng, N, 1, 3, 1)
    ctx.load_key(A)
    x = synth.uniform(rng, (B, 1, N))
    r = synth.small(rng, (B, 3, N))
    y = synth.gauss(rng, (B, 3, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    acc = ctx.open_verify(z, t, c, d)
    for b in range(0, B, 9):
        c_ref, t_ref, _ = O.open_commit(P, A, x[b], r[b], y[b])
        assert np.array_equal(c[b], c_ref) and np.array_equal(t[b], t_ref)
        assert int(acc[b]) == int(O.open_verify(P, A, z[b], t[b], c[b], d[b]) == 1)
    assert ok.all()
