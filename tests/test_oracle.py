"""Pin the CPU oracle (oracle/rzk_oracle.c) against the golden vectors and the reference's KATs.

Golden vectors come from tests/golden/make_golden.py (independent pure-Python big-int restatement,
seeded from the inputs of the reference's own tests).  Reference KATs pinned here:
  norm_1/2/inf([1,-2,3,-4]) = 10/5/4      src/polynomial.rs:107-120
  standard_deviation(1024) = 21780        src/params.rs:145-150
  Mat::dot / componentwise_mul structure  src/mat.rs:243-268, 389-406
"""
import numpy as np
import pytest

from oracle import oracle as O

Q = O.Q_DEFAULT
HALF = (Q - 1) // 2


def test_norm_kats(golden):
    k = golden["norm_kat"]
    assert O.norm1(k["p"]) == k["norm1"] == 10
    assert O.norm2(k["p"]) == k["norm2"] == 5
    assert O.norm_inf(k["p"]) == k["norm_inf"] == 4


def test_sigma_kat_and_bounds(golden):
    s = golden["sigma_kat"]
    assert O.Params(N=1024).sigma == s["sigma"] == 21780
    for row in golden["bounds"]:
        P = O.Params(N=row["N"], k=row["k"])
        assert P.sigma == row["sigma"]
        assert P.commit_bound == row["commit"]
        assert P.verify_bound == row["verify"]
    # SURVEY §8a row a8 values
    assert O.Params(N=512).commit_bound == 1359072 and O.Params(N=512).verify_bound == 679536
    assert O.Params(N=1024).commit_bound == 2787840 and O.Params(N=1024).verify_bound == 1393920


def test_center():
    assert O.center(HALF) == HALF
    assert O.center(HALF + 1) == -HALF
    assert O.center(-HALF - 1) == HALF
    assert O.center(Q) == 0
    assert O.center(-Q - 5) == -5


def test_mat_structure_n4(golden):
    g = golden["mat_dot_n4"]
    A = np.array([g["a"]], dtype=np.int64)  # 1 x 2
    B = np.array([[g["b"][0]], [g["b"][1]]], dtype=np.int64)  # 2 x 1
    out = O.mat_dot(A, B)
    assert out.shape == (1, 1, 4)
    assert out[0, 0].tolist() == g["out"] == [13, 35, 45, 30]
    g = golden["mat_cmul_n4"]
    out = O.mat_cmul(np.array([g["a"]], dtype=np.int64), g["elem"])
    assert out[0].tolist() == g["out"]
    assert g["out"][0] == [-8, 4, 10, 12] and g["out"][1] == [-14, 13, 28, 27]


def test_mat_add_sub_dims():
    A = np.zeros((2, 1, 4), dtype=np.int64)
    B = np.zeros((1, 2, 4), dtype=np.int64)
    with pytest.raises(AssertionError):
        O.mat_add(A, B)
    with pytest.raises(AssertionError):
        O.mat_sub(A, B)
    with pytest.raises(AssertionError):
        O.mat_dot(A, A)


def test_extreme_and_random_products(golden):
    for case in golden["extreme_products"] + golden["random_products"]:
        out = O.poly_mul(case["a"], case["b"])
        assert out.tolist() == case["out"], case.get("name", "random")


def test_wraparound():
    for N in (4, 16, 512):
        a = np.zeros(N, dtype=np.int64)
        b = np.zeros(N, dtype=np.int64)
        a[N - 1] = 1
        b[1] = 1
        out = O.poly_mul(a, b)
        assert out[0] == -1 and not out[1:].any()


def _arr(x):
    return np.array(x, dtype=np.int64)


def test_open_golden(golden):
    for g in golden["open"] + golden["open_big"]:
        P = O.Params(**{k: g["params"][k] for k in ("N", "n", "k", "l", "kappa", "b")})
        A = _arr(g["A"])
        c, t, ok = O.open_commit(P, A, _arr(g["x"]), _arr(g["r"]), _arr(g["y"]))
        assert ok == g["commit_ok"]
        assert c.tolist() == g["c"] and t.tolist() == g["t"]
        z = O.open_response(P, _arr(g["y"]), _arr(g["r"]), _arr(g["d"]))
        if not g["tampered"]:
            assert z.tolist() == g["z"]
            assert O.commitment_verify(P, A, c, _arr(g["x"]), _arr(g["r"]))
        acc = O.open_verify(P, A, _arr(g["z"]), t, c, _arr(g["d"]))
        assert bool(acc == 1) == g["accept"]
        assert g["accept"] == (not g["tampered"])


def test_linear_golden(golden):
    for g in golden["linear"]:
        P = O.Params(**{k: g["params"][k] for k in ("N", "n", "k", "l", "kappa", "b")})
        A = _arr(g["A"])
        c, cp, t, tp, u, ok = O.linear_commit(P, A, _arr(g["g"]), _arr(g["x"]), _arr(g["r"]),
                                              _arr(g["rp"]), _arr(g["y"]), _arr(g["yp"]))
        assert ok == g["commit_ok"]
        for name, val in (("c", c), ("cp", cp), ("t", t), ("tp", tp), ("u", u)):
            assert val.tolist() == g[name], name
        z, zp = O.linear_response(P, _arr(g["y"]), _arr(g["yp"]), _arr(g["r"]), _arr(g["rp"]), _arr(g["d"]))
        if not g["tampered"]:
            assert z.tolist() == g["z"] and zp.tolist() == g["zp"]
        acc = O.linear_verify(P, A, _arr(g["z"]), _arr(g["zp"]), c, cp, _arr(g["g"]), t, tp, u, _arr(g["d"]))
        assert bool(acc == 1) == g["accept"] == (not g["tampered"])


def test_sum_golden(golden):
    for g in golden["sum"]:
        P = O.Params(**{k: g["params"][k] for k in ("N", "n", "k", "l", "kappa", "b")})
        A = _arr(g["A"])
        cs, cp, ts, tp, u, ok = O.sum_commit(P, A, _arr(g["gs"]), _arr(g["xs"]), _arr(g["rs"]),
                                             _arr(g["rp"]), _arr(g["ys"]), _arr(g["yp"]))
        assert ok == g["commit_ok"]
        for name, val in (("cs", cs), ("cp", cp), ("ts", ts), ("tp", tp), ("u", u)):
            assert val.tolist() == g[name], name
        zs, zp = O.sum_response(P, _arr(g["ys"]), _arr(g["yp"]), _arr(g["rs"]), _arr(g["rp"]), _arr(g["d"]))
        if not g["tampered"]:
            assert zs.tolist() == g["zs"] and zp.tolist() == g["zp"]
        acc = O.sum_verify(P, A, _arr(g["zs"]), _arr(g["zp"]), cs, cp, _arr(g["gs"]), ts, tp, u, _arr(g["d"]))
        assert bool(acc == 1) == g["accept"] == (not g["tampered"])


def test_oracle_ntt_roundtrip_and_convolution():
    # auxiliary-prime NTT used as the checker of the device NTT kernels
    p = 1073692673  # prime, = 1 mod 8192
    rng = np.random.default_rng(5)
    for N in (8, 512, 1024):
        # find a primitive 2N-th root of unity
        for gcand in range(2, 100):
            psi = O.powmod(gcand, (p - 1) // (2 * N), p)
            if O.powmod(psi, N, p) == p - 1:
                break
        a = rng.integers(0, p, N, dtype=np.uint32)
        b = rng.integers(0, p, N, dtype=np.uint32)
        fa, fb = O.ntt_forward(a, p, psi), O.ntt_forward(b, p, psi)
        assert np.array_equal(O.ntt_inverse(fa, p, psi), a)
        prod = (fa.astype(np.uint64) * fb.astype(np.uint64) % p).astype(np.uint32)
        c = O.ntt_inverse(prod, p, psi).astype(object)
        # schoolbook negacyclic mod p
        ref = [0] * N
        for i in range(N if N <= 8 else 0):
            for j in range(N):
                t = i + j
                v = int(a[i]) * int(b[j])
                if t < N:
                    ref[t] = (ref[t] + v) % p
                else:
                    ref[t - N] = (ref[t - N] - v) % p
        if N <= 8:
            assert [int(v) for v in c] == ref


def test_commitment_verify_with_scalar_f():
    """Commitment::verify, f = Some(_) branch (commit.rs:199-206): f*c == a.(f*r) + f*[0;x] for every f,
    and the f = None form is the special case f = 1."""
    N, n, k, l = 16, 1, 3, 1
    P = O.Params(N, n, k, l)
    rng = np.random.default_rng(31)
    half = (O.Q_DEFAULT - 1) // 2
    A = O.key_build(P, rng.integers(-half, half + 1, (n, k - n, N)), rng.integers(-half, half + 1, (l, k - n - l, N)))
    x = rng.integers(-half, half + 1, (l, N))
    r = rng.integers(-1, 2, (k, N))
    c, ok = O.commit(P, A, x, r)
    assert ok and O.commitment_verify(P, A, c, x, r)
    one = np.zeros(N, dtype=np.int64)
    one[0] = 1
    assert O.commitment_verify(P, A, c, x, r, one)
    f = np.zeros(N, dtype=np.int64)
    f[0], f[3] = 2, -1
    rf = O.mat_cmul(r[:, None, :], f)[:, 0, :]
    assert O.commitment_verify(P, A, c, x, rf, f)
    assert not O.commitment_verify(P, A, c, x, r, f)         # r not scaled by f
    x2 = x.copy()
    x2[0, 1] = O.center(int(x2[0, 1]) + 1)
    assert not O.commitment_verify(P, A, c, x2, rf, f)
    big = rng.integers(-half, half + 1, (k, N))              # norm constraint on r comes first (commit.rs:183-185)
    cb, okb = O.commit(P, A, x, big)
    assert not okb and not O.commitment_verify(P, A, cb, x, big) and not O.commitment_verify(P, A, cb, x, big, one)


def test_oracle_batched_ntt_matches_the_single_transform():
    # the table-driven batch transform bench.py times on the host cores ("same-algorithm CPU") = rzko_ntt_forward per row
    p = 1073668097
    rng = np.random.default_rng(6)
    for N in (512, 2048):
        for gcand in range(2, 100):
            psi = O.powmod(gcand, (p - 1) // (2 * N), p)
            if O.powmod(psi, N, p) == p - 1:
                break
        a = rng.integers(0, p, (5, N), dtype=np.uint32)
        a[0, :] = p - 1
        got = O.ntt_forward_batch(a, p, psi, threads=2)
        for i in range(a.shape[0]):
            assert np.array_equal(got[i], O.ntt_forward(a[i], p, psi)), (N, i)

