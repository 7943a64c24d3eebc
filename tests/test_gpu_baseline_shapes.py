"""GPU parity at the BASELINE Sum shapes, the kernel branches only those shapes (or tuning knobs) reach,
non-canonical inputs, and the wire format feeding the GPU verifier.

  * (n,k,l) = (4,9,4), V = 8, N = 1024 (BASELINE config 3) and (8,17,8), V = 32, N = 2048 (config 5): the key
    products, the commitment, the Open cycle and every phase of the Sum cycle against oracle/rzk_oracle.c,
    bit for bit, plus a tampered proof.  These shapes select row groups with 4 accumulators, row blocks with
    several blocks per proof, the two-step A1 relation and the vector x vector sum programs at V = 8 / 32.
    Follows /root/reference/src/prove/sum.rs:99-320 and src/commit.rs:109-125 (through the oracle).
  * the same programs under RZK_ROW_GROUPS=0 / RZK_BLOCK_MIN_LOGN / RZK_SLOT_SHARE_MIN knobs, which force the
    shared-operand path and its "more primes than stored" fallback, and multi-block plans at N = 1024;
  * a coefficient k*2^32 + s (or any value outside the centred range) is never read as s: verifier entry points
    reject the proof, everything else fails the call — on the fused, the unfused and the small-N paths alike;
  * bincode messages (src/mat.rs:425-438, src/prove/open.rs:180-228) decoded by the wire codec and verified on
    the GPU.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from ring_zk_amd import synth, wire

pytestmark = pytest.mark.gpu

Q = O.Q_DEFAULT
HALF = (Q - 1) // 2


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


def make_ctx(N, n, k, l, env=None, **kw):
    """Context created under tuning knobs (the library reads them in rzk_ctx_create)."""
    from ring_zk_amd import Context

    env = env or {}
    old = {key: os.environ.get(key) for key in env}
    os.environ.update({key: str(v) for key, v in env.items()})
    try:
        return Context(N, n, k, l, **kw)
    finally:
        for key, v in old.items():
            if v is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = v


def P_of(ctx):
    return O.Params(N=ctx.N, n=ctx.n, k=ctx.k, l=ctx.l, kappa=ctx.kappa, b=ctx.b)


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def sum_inputs(rng, P, B, V):
    N, k, l = P.N, P.k, P.l
    gs = synth.uniform(rng, (B, V, N))
    xs = synth.uniform(rng, (B, V, l, N))
    rs, rp = synth.small(rng, (B, V, k, N), P.b), synth.small(rng, (B, k, N), P.b)
    ys, yp = synth.gauss(rng, (B, V, k, N), P.sigma), synth.gauss(rng, (B, k, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)
    return gs, xs, rs, rp, ys, yp, d


def check_sum_cycle(torch, ctx, A, B, V, seed, device_too=True):
    """Every phase of the Sum cycle vs the oracle, one tampered proof; returns nothing, asserts."""
    P = P_of(ctx)
    rng = np.random.default_rng(seed)
    gs, xs, rs, rp, ys, yp, d = sum_inputs(rng, P, B, V)
    cs, cp, ts, tp, u, ok = ctx.sum_commit(gs, xs, rs, rp, ys, yp)
    zs, zp = ctx.sum_response(ys, yp, rs, rp, d)
    acc = ctx.sum_verify(zs, zp, cs, cp, gs, ts, tp, u, d)
    for b in range(B):
        ref = O.sum_commit(P, A, gs[b], xs[b], rs[b], rp[b], ys[b], yp[b])      # sum.rs:99-178
        for got, want, name in zip((cs, cp, ts, tp, u), ref[:5], ("cs", "cp", "ts", "tp", "u")):
            assert np.array_equal(got[b], want), (name, b)
        assert bool(ok[b]) == ref[5]
        zr, zpr = O.sum_response(P, ys[b], yp[b], rs[b], rp[b], d[b])             # sum.rs:182-200
        assert np.array_equal(zs[b], zr) and np.array_equal(zp[b], zpr)
        assert O.sum_verify(P, A, zs[b], zp[b], cs[b], cp[b], gs[b], ts[b], tp[b], u[b], d[b]) == 1   # sum.rs:257-320
    assert acc.tolist() == [1] * B
    # tampered proofs: a response coefficient, a t coefficient of the last summand, the masked sum u
    last = B - 1
    zst = zs.copy()
    zst[last, V - 1, ctx.k - 1, 7] = O.center(int(zst[last, V - 1, ctx.k - 1, 7]) + 1)
    assert ctx.sum_verify(zst, zp, cs, cp, gs, ts, tp, u, d).tolist() == [1] * last + [0]
    assert O.sum_verify(P, A, zst[last], zp[last], cs[last], cp[last], gs[last], ts[last], tp[last], u[last], d[last]) != 1
    ut = u.copy()
    ut[0, ctx.l - 1, ctx.N - 1] = O.center(int(ut[0, ctx.l - 1, ctx.N - 1]) - 1)
    assert ctx.sum_verify(zs, zp, cs, cp, gs, ts, tp, ut, d).tolist() == [0] + [1] * last
    if device_too:   # device-pointer entry points give the same bytes
        D = lambda a: dev(torch, a)
        outs = ctx.sum_commit(D(gs), D(xs), D(rs), D(rp), D(ys), D(yp))
        for got, want in zip(outs, (cs, cp, ts, tp, u, ok)):
            assert np.array_equal(got.cpu().numpy(), want)
        zsd, zpd = ctx.sum_response(D(ys), D(yp), D(rs), D(rp), D(d))
        assert np.array_equal(zsd.cpu().numpy(), zs) and np.array_equal(zpd.cpu().numpy(), zp)
        accd = ctx.sum_verify(D(zst), zpd, outs[0], outs[1], D(gs), outs[2], outs[3], outs[4], D(d))
        assert accd.cpu().numpy().tolist() == [1] * last + [0]


def check_key_products_and_open(ctx, A, B, seed):
    """matvec(A1 / A2 / A), commit, Commitment::verify and the Open cycle vs the oracle (commit.rs:109-125,
    open.rs:80-174), with full-range and ternary vectors."""
    P = P_of(ctx)
    N, n, k, l = ctx.N, ctx.n, ctx.k, ctx.l
    rng = np.random.default_rng(seed)
    v = synth.uniform(rng, (B, k, N))
    v[B - 1] = synth.small(rng, (k, N))
    for which, sl in ((0, slice(0, n)), (1, slice(n, n + l)), (2, slice(0, n + l))):
        out = ctx.matvec(which, v)
        for b in range(B):
            assert np.array_equal(out[b], O.mat_dot(A[sl], v[b][:, None, :])[:, 0, :]), (which, b)
    x = synth.uniform(rng, (B, l, N))
    r = synth.small(rng, (B, k, N))
    y = synth.gauss(rng, (B, k, N), P.sigma)
    d = synth.challenge(rng, (B,), N, P.kappa)
    cm, okc = ctx.commit(x, r)
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    zt = z.copy()
    zt[0, 0, 0] = O.center(int(zt[0, 0, 0]) + 1)
    acc, acct = ctx.open_verify(z, t, c, d), ctx.open_verify(zt, t, c, d)
    for b in range(B):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[b], r[b], y[b])
        assert np.array_equal(c[b], c_ref) and np.array_equal(t[b], t_ref) and bool(ok[b]) == ok_ref
        assert np.array_equal(cm[b], c_ref) and bool(okc[b]) == ok_ref
        assert np.array_equal(z[b], O.open_response(P, y[b], r[b], d[b]))
        assert O.open_verify(P, A, z[b], t[b], c[b], d[b]) == 1
    assert acc.tolist() == [1] * B and acct.tolist() == [0] + [1] * (B - 1)
    assert ctx.commitment_verify(c, x, r).tolist() == [1] * B


# ---- BASELINE config 3: SumProof, N = 1024, (4,9,4), V = 8 ----------------------------------------------------------
def test_config3_shape_vs_oracle(torch_mod):
    N, n, k, l, V = 1024, 4, 9, 4, 8
    ctx = make_ctx(N, n, k, l)
    A = synth.key(np.random.default_rng(403), N, n, k, l)
    ctx.load_key(A)
    check_key_products_and_open(ctx, A, 2, 404)
    check_sum_cycle(torch_mod, ctx, A, 2, V, 405)


def test_config3_full_batch_properties(torch_mod):
    """config 3 at its full batch (4096 proofs): completeness, and exactly the tampered proofs rejected."""
    T = torch_mod
    N, n, k, l, V, B = 1024, 4, 9, 4, 8, 4096
    ctx = make_ctx(N, n, k, l)
    ctx.generate_key(77)
    half = (ctx.q - 1) // 2
    sid = iter(range(64))
    uni = lambda *lead: ctx.sample_uniform(5, next(sid), half, lead)
    small = lambda *lead: ctx.sample_uniform(5, next(sid), ctx.b, lead)
    gauss = lambda *lead: ctx.sample_gauss(5, next(sid), float(ctx.sigma), lead)
    gs, xs = uni(B, V), uni(B, V, l)
    rs, rp = small(B, V, k), small(B, k)
    ys, yp = gauss(B, V, k), gauss(B, k)
    d = ctx.sample_challenge(5, next(sid), (B,))
    cs, cp, ts, tp, u, ok = ctx.sum_commit(gs, xs, rs, rp, ys, yp)
    zs, zp = ctx.sum_response(ys, yp, rs, rp, d)
    acc = ctx.sum_verify(zs, zp, cs, cp, gs, ts, tp, u, d)
    assert int(ok.sum()) == B and int(acc.sum()) == B
    idx = T.arange(5, B, 97, device=zs.device)
    zs2 = zs.clone()
    zs2[idx, V - 1, k - 1, N - 1] += 1
    cp2 = cp.clone()
    idc = T.arange(11, B, 131, device=zs.device)
    cp2[idc, n + l - 1, 3] = T.where(cp2[idc, n + l - 1, 3] >= half, cp2[idc, n + l - 1, 3] - 1, cp2[idc, n + l - 1, 3] + 1)
    acc2 = ctx.sum_verify(zs2, zp, cs, cp2, gs, ts, tp, u, d).cpu().numpy()
    expect = np.ones(B, dtype=np.uint8)
    expect[idx.cpu().numpy()] = 0
    expect[idc.cpu().numpy()] = 0
    assert np.array_equal(acc2, expect)
    # u is what the Mat-level primitives give for one proof of the batch: sum_i (a2.y_i)(.)g_i - a2.yp (sum.rs:154-160)
    b0 = 1234
    w = ctx.matvec(1, ys[b0].contiguous())                              # [V][l][N]
    prod = ctx.cmul(w, gs[b0].contiguous())
    accu = prod[0]
    for i in range(1, V):
        accu = ctx.add(accu, prod[i])
    assert T.equal(ctx.sub(accu, ctx.matvec(1, yp[b0:b0 + 1].contiguous())[0]), u[b0])


# ---- BASELINE config 5: SumProof, N = 2048, (8,17,8), V = 32 ----------------------------------------------------------
def test_config5_shape_vs_oracle(torch_mod):
    N, n, k, l, V = 2048, 8, 17, 8, 32
    ctx = make_ctx(N, n, k, l)
    A = synth.key(np.random.default_rng(503), N, n, k, l)
    ctx.load_key(A)
    check_key_products_and_open(ctx, A, 1, 504)
    check_sum_cycle(torch_mod, ctx, A, 1, V, 505, device_too=False)


# ---- branches that only tuning knobs reach ----------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    # shared-operand path (fwd_slots + row_slots), key-only programs store 2 primes: full-range operands need a third
    # one -> "more primes than stored" fallback of row_slots_kernel
    dict(N=512, shape=(2, 5, 2), V=3, env={"RZK_ROW_GROUPS": 0, "RZK_BLOCK_MIN_LOGN": 12}),
    dict(N=1024, shape=(4, 9, 4), V=2, env={"RZK_ROW_GROUPS": 0, "RZK_BLOCK_MIN_LOGN": 12}),
    dict(N=2048, shape=(2, 5, 2), V=2, env={"RZK_ROW_GROUPS": 0, "RZK_BLOCK_MIN_LOGN": 12}),
    # row blocks at N = 1024: OPEN_COMMIT at (4,9,4) needs two blocks per proof (5 + 5 operands > 9 slots)
    dict(N=1024, shape=(4, 9, 4), V=2, env={"RZK_BLOCK_MIN_LOGN": 10}),
    # no sharing at all: every row transforms its own operands (plain row kernel), transform products for the challenge
    dict(N=1024, shape=(2, 5, 2), V=2, env={"RZK_ROW_GROUPS": 0, "RZK_BLOCK_MIN_LOGN": 12, "RZK_SLOT_SHARE_MIN": 0,
                                            "RZK_SHIFT": 0}),
    # groups of two rows only
    dict(N=1024, shape=(4, 9, 4), V=2, env={"RZK_GROUP_MAX": 2}),
    # vector x vector programs through unit_kernel's HAS_VEC variants (the default sends them to row_kernel)
    dict(N=1024, shape=(1, 3, 1), V=3, env={"RZK_VEC_ROWS": 0}),
    dict(N=512, shape=(2, 5, 2), V=2, env={"RZK_VEC_ROWS": 0, "RZK_SLOT_SHARE_MIN": 0}),
    dict(N=2048, shape=(1, 3, 1), V=2, env={"RZK_VEC_ROWS": 0}),
    # N = 2048 with one wavefront per polynomial (default: two), plain rows and through unit_kernel's HAS_VEC variant
    dict(N=2048, shape=(1, 3, 1), V=2, env={"RZK_PAIR_POLY": 0}),
    dict(N=2048, shape=(2, 5, 2), V=2, env={"RZK_PAIR_POLY": 0, "RZK_VEC_ROWS": 0, "RZK_BLOCK_MIN_LOGN": 12}),
    # ... and two per polynomial without row blocks: grouped rows fall to the unit kernel's pairs
    dict(N=2048, shape=(2, 5, 2), V=3, env={"RZK_BLOCK_MIN_LOGN": 12}),
    # unit_io_kernel (operands read once) where it is not the default: global parking lines, one- and two-wavefront teams;
    # and unit_kernel where unit_io_kernel is the default
    dict(N=1024, shape=(1, 3, 1), V=2, env={"RZK_UNIT_IO": 1}),
    dict(N=1024, shape=(2, 5, 2), V=2, env={"RZK_UNIT_IO": 1, "RZK_ROW_GROUPS": 0}),
    dict(N=2048, shape=(1, 3, 1), V=2, env={"RZK_UNIT_IO": 1}),
    dict(N=2048, shape=(2, 5, 2), V=2, env={"RZK_UNIT_IO": 1, "RZK_BLOCK_MIN_LOGN": 12}),
    dict(N=512, shape=(2, 5, 2), V=2, env={"RZK_UNIT_IO": 0, "RZK_ROW_GROUPS": 0}),
    dict(N=512, shape=(2, 5, 2), V=2, env={"RZK_ROW_GROUPS": 0}),
    # N = 2048, two wavefronts per polynomial: challenge products by transforms (the default takes rotations with the
    # non-zero list in LDS: shift_row_kernel<11, ., PairTeam> and the HAS_SHIFT variants of unit_kernel / row_kernel)
    dict(N=2048, shape=(1, 3, 1), V=2, env={"RZK_SHIFT": 0}),
    dict(N=2048, shape=(2, 5, 2), V=2, env={"RZK_SHIFT": 0}),
    dict(N=2048, shape=(2, 5, 2), V=2, env={"RZK_VEC_ROWS": 0, "RZK_BLOCK_MIN_LOGN": 12}),   # unit_kernel<11, true, true, PairTeam>
    # Sum proof's u and final relation through a2.(sum_i g_i v_i - v') (the default where it saves transforms: large V,
    # the reference's key shape) and row by row, each forced where the cost model would choose the other
    dict(N=512, shape=(1, 3, 1), V=3, env={"RZK_SUM_D": 1}),
    dict(N=1024, shape=(2, 5, 2), V=2, env={"RZK_SUM_D": 1}),
    dict(N=2048, shape=(2, 5, 2), V=3, env={"RZK_SUM_D": 1}),
    dict(N=1024, shape=(4, 9, 4), V=8, env={"RZK_SUM_D": 0}),
    # the scalar multipliers as prepared images (TERM_DKEY) forced at shapes below the use threshold, and switched off
    # where they are the default
    dict(N=512, shape=(1, 3, 1), V=3, env={"RZK_DKEY": 2}),
    dict(N=2048, shape=(1, 3, 1), V=2, env={"RZK_DKEY": 2}),
    dict(N=1024, shape=(2, 5, 2), V=2, env={"RZK_DKEY": 2, "RZK_SUM_D": 1}),
    dict(N=1024, shape=(4, 9, 4), V=8, env={"RZK_DKEY": 0}),
    dict(N=2048, shape=(2, 5, 2), V=3, env={"RZK_DKEY": 0}),
    # operand images (the D rows read the transforms the a1.v_i products left): producers on the group kernel
    # (N <= 1024), the block kernel (N = 2048), none at all (unit path: the rows fall back to their own transforms)
    dict(N=1024, shape=(2, 5, 2), V=3, env={"RZK_DKEY": 2, "RZK_SUM_D": 1}),
    dict(N=2048, shape=(2, 5, 2), V=3, env={"RZK_DKEY": 2, "RZK_SUM_D": 1}),
    dict(N=512, shape=(2, 5, 2), V=2, env={"RZK_DKEY": 2, "RZK_SUM_D": 1, "RZK_ROW_GROUPS": 0, "RZK_BLOCK_MIN_LOGN": 12}),
    dict(N=1024, shape=(4, 9, 4), V=8, env={"RZK_OIMG": 0}),
])
def test_forced_kernel_paths_vs_oracle(torch_mod, cfg):
    N, (n, k, l), V = cfg["N"], cfg["shape"], cfg["V"]
    ctx = make_ctx(N, n, k, l, env=cfg["env"])
    A = synth.key(np.random.default_rng(600 + N + n), N, n, k, l)
    ctx.load_key(A)
    check_key_products_and_open(ctx, A, 2, 601 + n)
    check_sum_cycle(torch_mod, ctx, A, 2, V, 602 + n, device_too=False)


# ---- operands of very different sizes in one batch: the per-proof prime count (1, 2 or 3) is decided from the norms
@pytest.mark.parametrize("io", [None, 0, 1])
@pytest.mark.parametrize("upt", [0, 64])
@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_prime_count_per_proof_vs_oracle(torch_mod, N, upt, io):
    """Randomness of six different magnitudes in one batch — ternary, two sparse spikes, constant 8, one large pair, a
    single -1, full range — so that neighbouring proofs of one launch need 1, 2 and 3 auxiliary primes; upt = 64 walks
    all units of a proof in ONE wavefront (the path the 4096-proof batches take) at this small batch.
    commit.rs:109-125 (oracle: O.commit)."""
    n, k, l = 1, 3, 1
    env = {"RZK_UPT": upt} if upt else {}
    if io is not None:
        env["RZK_UNIT_IO"] = io   # unit_io_kernel everywhere / nowhere (its third-prime pass and its two-prime rows)
    ctx = make_ctx(N, n, k, l, env=env)
    P = P_of(ctx)
    rng = np.random.default_rng(700 + N)
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    B = 6
    x = synth.uniform(rng, (B, l, N))
    r = np.zeros((B, k, N), dtype=np.int64)
    r[0] = synth.small(rng, (k, N))                       # ternary
    r[1, 1, 0], r[1, 2, 3] = 16000, -383                  # two spikes
    r[1, 0] = synth.small(rng, (N,))
    r[2, 1, :N] = 8
    r[2, 2, :N] = 8                                       # constant 8
    r[3, 1, 0], r[3, 1, 1] = 16383, 1
    r[4, 2, 5] = -1                                       # a single -1: one prime suffices
    r[5] = synth.uniform(rng, (k, N))                     # full range: three primes
    y = synth.gauss(rng, (B, k, N), P.sigma)
    cm, okc = ctx.commit(x, r)
    c, t, ok = ctx.open_commit(x, r, y)
    for b in range(B):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[b], r[b], y[b])
        assert np.array_equal(c[b], c_ref) and np.array_equal(t[b], t_ref) and bool(ok[b]) == ok_ref, b
        assert np.array_equal(cm[b], c_ref) and bool(okc[b]) == ok_ref, b
    # Commitment::verify (commit.rs:199-209) runs its rows through the same units
    assert ctx.commitment_verify(c, x, r).tolist() == [int(bool(v)) for v in ok]


# ---- non-canonical inputs ------------------------------------------------------------------------------------------
def _open_proof(ctx, B, seed):
    P = P_of(ctx)
    rng = np.random.default_rng(seed)
    A = synth.key(rng, ctx.N, ctx.n, ctx.k, ctx.l)
    ctx.load_key(A)
    x = synth.uniform(rng, (B, ctx.l, ctx.N))
    r = synth.small(rng, (B, ctx.k, ctx.N), ctx.b)
    y = synth.gauss(rng, (B, ctx.k, ctx.N), P.sigma)
    d = synth.challenge(rng, (B,), ctx.N, P.kappa)
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    assert ok.tolist() == [1] * B and ctx.open_verify(z, t, c, d).tolist() == [1] * B
    return A, x, r, y, d, c, t, z


@pytest.mark.parametrize("cfg", [
    dict(N=512),                               # fused norm predicate + rotation term (the fast path)
    dict(N=1024, env={"RZK_SHIFT": 0}),        # challenge product by transforms
    dict(N=512, b=64),                         # verify bound >= 2^24: separate exact norm kernel
    dict(N=2048),                              # rotation term walked from the non-zero list of a two-wavefront team
    dict(N=2048, env={"RZK_PAIR_POLY": 0}),    # one wavefront per polynomial: no rotations at this size
    dict(N=16, kappa=8),                       # schoolbook kernel (the reference's test size, tests/test.rs:8)
    dict(N=1024, shape=(2, 5, 2)),             # two-step A1 relation: grouped a1.z + rotation rows
])
def test_noncanonical_verifier_inputs_reject_the_proof(torch_mod, cfg):
    """z, t, c or d with a coefficient outside [-(q-1)/2, (q-1)/2] — in particular s + 2^32, whose low word is the
    honest value — must clear accept for that proof and only that proof; ZqI64::from (params.rs:126) would have
    produced a different residue, never s."""
    n, k, l = cfg.get("shape", (1, 3, 1))
    ctx = make_ctx(cfg["N"], n, k, l, env=cfg.get("env"), b=cfg.get("b", 1), kappa=cfg.get("kappa", 36))
    B = 4
    A, x, r, y, d, c, t, z = _open_proof(ctx, B, 700 + cfg["N"])
    bad_words = (1 << 32, -(1 << 32), 5 << 32, 1 << 62)
    for name, arr, pos in (("z", z, (1, k - 1, 5)), ("z", z, (2, 0, 0)), ("t", t, (1, 0, 3)), ("c", c, (2, 0, ctx.N - 1)),
                           ("d", d, (3, 1))):
        for delta in bad_words[:2] if name != "z" else bad_words:
            m = arr.copy()
            m[pos] += delta                      # same low word, different integer
            args = dict(z=z, t=t, c=c, d=d)
            args[name] = m
            acc = ctx.open_verify(args["z"], args["t"], args["c"], args["d"])
            want = [1] * B
            want[pos[0]] = 0
            assert acc.tolist() == want, (name, pos, delta)
            accd = ctx.open_verify(*(dev(torch_mod, args[key]) for key in ("z", "t", "c", "d")))
            assert accd.cpu().numpy().tolist() == want
            ctx.synchronize()                    # verifier-side faults never fail the call
    # in the 32-bit range but not the centred representative: v + q for a negative v (same residue class)
    m = z.copy()
    neg = np.argwhere(m[0, 0] < 0)[0][0]
    m[0, 0, neg] += Q
    assert ctx.open_verify(m, t, c, d).tolist() == [0] + [1] * (B - 1)
    # Commitment::verify is a verifier too
    cm = c.copy()
    cm[1, 0, 0] += 1 << 32
    assert ctx.commitment_verify(cm, x, r).tolist() == [1, 0, 1, 1]


@pytest.mark.parametrize("N", [512, 16])
def test_noncanonical_prover_and_mat_inputs_fail_the_call(torch_mod, N):
    from ring_zk_amd.backend import RzkError

    ctx = make_ctx(N, 1, 3, 1, kappa=36 if N >= 64 else 8)
    B = 3
    A, x, r, y, d, c, t, z = _open_proof(ctx, B, 800 + N)
    xb = x.copy()
    xb[1, 0, 2] += 1 << 32
    with pytest.raises(RzkError):
        ctx.open_commit(xb, r, y)
    with pytest.raises(RzkError):
        ctx.commit(xb, r)
    yb = y.copy()
    yb[0, 2, 0] -= 1 << 32
    with pytest.raises(RzkError):
        ctx.open_commit(x, r, yb)
    with pytest.raises(RzkError):
        ctx.open_response(yb, r, d)
    db = d.copy()
    db[2, 1] += 1 << 32
    with pytest.raises(RzkError):
        ctx.open_response(y, r, db)
    a = synth.uniform(np.random.default_rng(3), (B, N))
    ab = a.copy()
    ab[2, N - 1] = HALF + 1                       # one past the range
    for call in (lambda: ctx.polymul(ab, a), lambda: ctx.polymul(a, ab), lambda: ctx.add(ab, a), lambda: ctx.sub(a, ab),
                 lambda: ctx.cmul(x, ab), lambda: ctx.matvec(2, np.stack([ab, a, a], axis=1)),
                 lambda: ctx.eq(ab[:, None, :], ab[:, None, :]), lambda: ctx.norm2_le(ab[:, None, :], 2 ** 40)):
        with pytest.raises(RzkError):
            call()
    # the sticky condition is cleared by the failing call: the same context keeps working
    assert np.array_equal(ctx.polymul(a, a)[0], O.poly_mul(a[0], a[0]))
    assert np.array_equal(ctx.canonicalize(ab), np.where(ab > HALF, ab - Q, ab))
    # device-pointer variants report at the next synchronisation
    out = ctx.polymul(dev(torch_mod, ab), dev(torch_mod, a))
    with pytest.raises(RzkError):
        ctx.synchronize()
    ctx.synchronize()
    del out
    c2, t2, ok2 = ctx.open_commit(dev(torch_mod, xb), dev(torch_mod, r), dev(torch_mod, y))
    with pytest.raises(RzkError):
        ctx.synchronize()
    assert ok2.cpu().numpy().tolist() == [1, 0, 1]          # the offending proof's flag is cleared as well


# ---- wire format -> GPU verifier ---------------------------------------------------------------------
@pytest.mark.parametrize("coef_bytes", [8, 4])
def test_wire_messages_feed_the_gpu_verifier(torch_mod, coef_bytes):
    """OpenProofCommitment { c: Commitment { c: Mat }, t: Mat } and OpenProofResponse { z: Mat }
    (src/prove/open.rs:180-228; Mat serde src/mat.rs:11-14, pinned by src/mat.rs:425-438) serialised, decoded with
    the range check and verified with rzk_open_verify_batch(_dev); verdicts equal the oracle's."""
    N, n, k, l, B = 512, 1, 3, 1, 3
    ctx = make_ctx(N, n, k, l)
    P = P_of(ctx)
    A, x, r, y, d, c, t, z = _open_proof(ctx, B, 900)
    z[1, 2, 9] = O.center(int(z[1, 2, 9]) + 1)             # proof 1 is wrong
    # (q-1)/2 = 1757668526 < 2^31: every centred residue also fits the 4-byte width of the reference's test vector
    msgs = []
    for b in range(B):
        commitment = wire.mat_encode(c[b][:, None, :], coef_bytes) + wire.mat_encode(t[b][:, None, :], coef_bytes)
        response = wire.mat_encode(z[b][:, None, :], coef_bytes)
        msgs.append((commitment, response))
    cs, ts, zs = [], [], []
    for commitment, response in msgs:
        cb, used = wire.mat_decode(commitment, N, coef_bytes, q=Q)
        tb, used2 = wire.mat_decode(commitment[used:], N, coef_bytes, q=Q)
        zb, used3 = wire.mat_decode(response, N, coef_bytes, q=Q)
        assert used + used2 == len(commitment) and used3 == len(response)
        assert cb.shape == (n + l, 1, N) and tb.shape == (n, 1, N) and zb.shape == (k, 1, N)
        cs.append(cb[:, 0]), ts.append(tb[:, 0]), zs.append(zb[:, 0])
    cs, ts, zs = np.stack(cs), np.stack(ts), np.stack(zs)
    assert np.array_equal(cs, c) and np.array_equal(ts, t) and np.array_equal(zs, z)
    want = [int(O.open_verify(P, A, zs[b], ts[b], cs[b], d[b]) == 1) for b in range(B)]
    assert want == [1, 0, 1]
    assert ctx.open_verify(zs, ts, cs, d).tolist() == want
    assert ctx.open_verify(dev(torch_mod, zs), dev(torch_mod, ts), dev(torch_mod, cs), dev(torch_mod, d)).cpu().tolist() == want
    # truncated message
    with pytest.raises(ValueError):
        wire.mat_decode(msgs[0][1][:-3], N, coef_bytes, q=Q)
    # out-of-range coefficient on the wire: rejected by the codec when it knows q ...
    evil = z[0][:, None, :].copy()
    evil[1, 0, 4] += 1 << 32
    if coef_bytes == 8:
        data = wire.mat_encode(evil, 8)
        with pytest.raises(ValueError):
            wire.mat_decode(data, N, 8, q=Q)
        # ... and, decoded as plain integers (q = 0), by the verifier kernels: the proof is rejected, not read as z[0]
        raw, _ = wire.mat_decode(data, N, 8)
        zz = zs.copy()
        zz[0] = raw[:, 0]
        assert ctx.open_verify(zz, ts, cs, d).tolist() == [0, 0, 1]
