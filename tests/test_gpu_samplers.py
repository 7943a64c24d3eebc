"""Device-side samplers (include/rzk.h "device-side samplers"): the distributions of the reference's RNG helpers
(src/polynomial.rs:14-44, src/challenge_space.rs:12-33).  Parity is statistical: every check is a distribution
property with a tolerance of several standard errors (stated where used), plus determinism in the seed and the
reference's own structural test of the challenge set (challenge_space.rs:56-82: |c|_1 = kappa, |c|_inf = 1)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
Q = O.Q_DEFAULT
HALF = (Q - 1) // 2


@pytest.fixture(scope="module")
def ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from ring_zk_amd import Context

    return Context(1024, 1, 3, 1)


def test_uniform_small_and_full_range(ctx):
    r = ctx.sample_uniform(7, 0, 1, (512, 3)).cpu().numpy()          # commitment randomness, b = 1 (commit.rs:101)
    assert r.shape == (512, 3, 1024) and r.min() == -1 and r.max() == 1
    n = r.size
    for v in (-1, 0, 1):                                              # each value 1/3; 6 standard errors
        assert abs((r == v).mean() - 1 / 3) < 6 * np.sqrt((1 / 3) * (2 / 3) / n)
    assert ctx.commit_bound > 0 and all(O.check_norm(r[b], ctx.commit_bound) for b in range(8))
    u = ctx.sample_uniform(7, 1, HALF, (256,)).cpu().numpy().astype(np.float64)   # key / message range
    assert u.min() >= -HALF and u.max() <= HALF
    assert abs(u.mean()) < 6 * (HALF / np.sqrt(3)) / np.sqrt(u.size)
    assert abs(u.std() / (Q / np.sqrt(12)) - 1) < 0.01
    assert u.max() > 0.999 * HALF and u.min() < -0.999 * HALF        # the ends of the range are reached
    hist = np.histogram(u, bins=64, range=(-HALF - 1, HALF + 1))[0]
    expect = u.size / 64
    assert ((hist - expect) ** 2 / expect).sum() < 63 + 6 * np.sqrt(2 * 63)   # chi-square, 63 dof


def test_gauss_matches_truncated_normal(ctx):
    sigma = float(ctx.sigma)                                          # 21780 at N = 1024, k = 3 (params.rs:149)
    y = ctx.sample_gauss(11, 0, sigma, (256, 3)).cpu().numpy()
    assert y.dtype == np.int64
    f = y.astype(np.float64)
    n = f.size
    assert abs(f.mean()) < 6 * sigma / np.sqrt(n)
    assert abs(f.std() / sigma - 1) < 0.01                            # truncation toward zero changes sigma by ~1e-9
    assert abs(((f / sigma) ** 4).mean() - 3.0) < 0.1                 # kurtosis of a normal
    assert abs((np.abs(f) < sigma).mean() - 0.6827) < 0.005
    assert np.abs(f).max() < 8 * sigma
    # truncation toward zero (I::from_f64): |trunc(x)| has P(0) = P(|x| < 1) = 2 * pdf(0) = 0.7979 / sigma
    small = ctx.sample_gauss(12, 0, 3.0, (512,)).cpu().numpy()
    p0 = (small == 0).mean()
    from math import erf, sqrt
    want = erf(1 / (3.0 * sqrt(2)))
    assert abs(p0 - want) < 6 * np.sqrt(want * (1 - want) / small.size)
    # honest responses built from sampled y pass the verifier's norm predicate (open.rs:167-169)
    assert all(O.check_norm(y[b], ctx.verify_bound) for b in range(16))


def test_challenge_set_structure_and_uniformity(ctx):
    B = 4096
    d = ctx.sample_challenge(3, 0, (B,)).cpu().numpy()
    assert d.shape == (B, 1024)
    assert (np.abs(d).sum(axis=1) == ctx.kappa).all() and np.abs(d).max() == 1     # challenge_space.rs:56-82
    nz = d != 0
    per_pos = nz.sum(axis=0).astype(np.float64)                        # each position hit with prob kappa / N
    expect = B * ctx.kappa / 1024
    assert ((per_pos - expect) ** 2 / (expect * (1 - ctx.kappa / 1024))).sum() < 1023 + 6 * np.sqrt(2 * 1023)
    signs = d[nz]
    assert abs((signs == 1).mean() - 0.5) < 6 * 0.5 / np.sqrt(signs.size)
    assert len({d[b].tobytes() for b in range(B)}) == B                # all distinct


def test_determinism_and_stream_separation(ctx):
    a = ctx.sample_gauss(5, 2, 100.0, (64, 3)).cpu().numpy()
    b = ctx.sample_gauss(5, 2, 100.0, (64, 3)).cpu().numpy()
    assert np.array_equal(a, b)                                         # same (seed, stream) -> same bytes
    c = ctx.sample_gauss(5, 3, 100.0, (64, 3)).cpu().numpy()
    e = ctx.sample_gauss(6, 2, 100.0, (64, 3)).cpu().numpy()
    assert not np.array_equal(a, c) and not np.array_equal(a, e)
    head = ctx.sample_gauss(5, 2, 100.0, (16, 3)).cpu().numpy()         # prefix property: polynomial i does not
    assert np.array_equal(head, a[:16])                                 # depend on how many are drawn
    assert abs(np.corrcoef(a.ravel(), c.ravel())[0, 1]) < 0.02


def test_sampled_inputs_drive_a_full_open_cycle(ctx):
    """Prover and verifier fed entirely by the device-side samplers: every honest proof is accepted."""
    import torch

    from ring_zk_amd import synth

    rng = np.random.default_rng(1)
    A = synth.key(rng, 1024, 1, 3, 1)
    ctx.load_key(A)
    B = 256
    x = ctx.sample_uniform(21, 0, HALF, (B, 1))
    r = ctx.sample_uniform(21, 1, 1, (B, 3))
    y = ctx.sample_gauss(21, 2, float(ctx.sigma), (B, 3))
    d = ctx.sample_challenge(21, 3, (B,))
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    acc = ctx.open_verify(z, t, c, d)
    torch.cuda.synchronize()
    assert int(ok.sum()) == B and int(acc.sum()) == B
    P = O.Params(1024, 1, 3, 1)
    for b in (0, B - 1):
        assert O.open_verify(P, A, z[b].cpu().numpy(), t[b].cpu().numpy(), c[b].cpu().numpy(), d[b].cpu().numpy()) == 1


def test_bad_arguments(ctx):
    from ring_zk_amd.backend import RzkError

    with pytest.raises(RzkError):
        ctx.sample_uniform(1, 0, 0, (4,))                 # bound must be positive (polynomial.rs:12-13)
    with pytest.raises(RzkError):
        ctx.sample_uniform(1, 0, HALF + 1, (4,))          # beyond the centred range
    with pytest.raises(RzkError):
        ctx.sample_gauss(1, 0, 0.0, (4,))


@pytest.mark.parametrize("shape", [(1024, 1, 3, 1), (512, 2, 5, 2), (16, 1, 3, 1)])
def test_generated_key_has_the_reference_structure(shape):
    """CommitmentKey::new (commit.rs:33-60): a1 = [I_n | U], a2 = [0 | I_l | U]; a commitment made with the
    generated key verifies against the oracle, which takes the key as plain data."""
    from ring_zk_amd import Context

    N, n, k, l = shape
    c = Context(N, n, k, l)
    A = c.generate_key(77)
    assert A.shape == (n + l, k, N)
    one = np.zeros(N, dtype=np.int64)
    one[0] = 1
    for i in range(n):
        for j in range(n):
            assert np.array_equal(A[i, j], one if i == j else 0 * one)
    for i in range(l):
        for j in range(n):
            assert not A[n + i, j].any()
        for j in range(l):
            assert np.array_equal(A[n + i, n + j], one if i == j else 0 * one)
    rnd = np.concatenate([A[:n, n:].ravel(), A[n:, n + l:].ravel()]).astype(np.float64)
    assert np.abs(rnd).max() <= HALF and rnd.size == (n * (k - n) + l * (k - n - l)) * N
    if rnd.size >= 4096:
        assert abs(rnd.std() / (Q / np.sqrt(12)) - 1) < 0.05
    assert not np.array_equal(c.generate_key(78), A)
    assert np.array_equal(c.generate_key(77), A)          # deterministic in the seed; key 77 is loaded again
    rng = np.random.default_rng(2)
    P = O.Params(N, n, k, l)
    x = rng.integers(-HALF, HALF + 1, (1, l, N))
    r = rng.integers(-1, 2, (1, k, N))
    cm, ok = c.commit(x, r)
    c_ref, ok_ref = O.commit(P, A, x[0], r[0])
    assert np.array_equal(cm[0], c_ref) and bool(ok[0]) == ok_ref
    assert O.commitment_verify(P, A, cm[0], x[0], r[0])


def test_uniform_and_challenge_kernels_match_their_cpu_statement(ctx):
    """The counter-based samplers are a function of (seed, stream, polynomial): tests/emul/emul.cpp states that function
    on the CPU (Philox block per coefficient pair; Floyd's kappa-subset for the challenge, challenge_space.rs:12-33) and
    the kernels — which draw in parallel, two coefficients per thread / one Floyd step per lane — must reproduce it
    bit for bit, whatever the launch shape and the alignment of the output buffer."""
    import ctypes as C

    import torch

    from test_emul_core import load_emul

    L = load_emul()
    N = 1024
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    want = np.empty(N, dtype=np.int64)
    for bound, stream in ((1, 3), (HALF, 4), (12345, 5)):
        got = ctx.sample_uniform(77, stream, bound, (9,)).cpu().numpy()
        for poly in (0, 5, 8):
            L.emul_sample_uniform(C.c_uint64(77), C.c_uint32(stream), C.c_uint64(poly), C.c_uint32(N), C.c_uint32(bound), p64(want))
            assert np.array_equal(got[poly], want), (bound, poly)
    got = ctx.sample_challenge(78, 2, (70,)).cpu().numpy()
    for poly in (0, 1, 63, 69):
        L.emul_sample_challenge(C.c_uint64(78), C.c_uint32(2), C.c_uint64(poly), C.c_uint32(N), C.c_uint32(ctx.kappa), p64(want))
        assert np.array_equal(got[poly], want), poly
    # an output buffer that is only 8-byte aligned takes the 8-byte store path: same values
    buf = torch.empty(3 * N + 1, dtype=torch.int64, device="cuda")
    view = buf[1:]
    assert view.data_ptr() % 16 == 8
    L2 = ctx._L
    assert L2.rzk_sample_challenge_dev(ctx._h, 78, 2, C.c_void_p(view.data_ptr()), 3) == 0
    assert np.array_equal(view.cpu().numpy().reshape(3, N), got[:3])
    assert L2.rzk_sample_uniform_dev(ctx._h, 77, 3, 1, C.c_void_p(view.data_ptr()), 3) == 0
    assert np.array_equal(view.cpu().numpy().reshape(3, N), ctx.sample_uniform(77, 3, 1, (3,)).cpu().numpy())
    assert L2.rzk_sample_gauss_dev(ctx._h, 79, 0, C.c_double(1000.0), C.c_void_p(view.data_ptr()), 3) == 0
    assert np.array_equal(view.cpu().numpy().reshape(3, N), ctx.sample_gauss(79, 0, 1000.0, (3,)).cpu().numpy())
