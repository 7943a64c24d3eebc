"""Size-independent properties at BASELINE.json's full sizes (the schoolbook oracle cannot check
4096-proof batches at N=1024 in test time): completeness of the whole cycle, soundness probes
(tampered responses / commitments are rejected and only those), ring laws of the multiply
(commutativity, distributivity over add, x^N = -1), Commitment::verify as an identity, and agreement
between the fused phase entry points and the Mat-level primitives they are made of.
"""
import numpy as np
import pytest

from ring_zk_amd import synth

pytestmark = pytest.mark.gpu

Q = 3515337053
HALF = (Q - 1) // 2


@pytest.fixture(scope="module")
def T():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    return torch


def make(T, N, n, k, l, B, seed):
    from ring_zk_amd import Context

    ctx = Context(N, n, k, l)
    dev = T.device("cuda", 0)
    g = T.Generator(device=dev)
    g.manual_seed(seed)
    A = synth.t_key(g, N, n, k, l, dev)
    ctx.load_key(A)
    x = synth.t_uniform(g, (B, l, N), dev)
    r = synth.t_small(g, (B, k, N), dev)
    y = synth.t_gauss(g, (B, k, N), dev, ctx.sigma)
    d = synth.t_challenge(g, B, N, ctx.kappa, dev)
    return ctx, A, x, r, y, d, g


@pytest.mark.parametrize("cfg", [(512, 1, 3, 1, 4096), (1024, 1, 3, 1, 4096), (2048, 1, 3, 1, 512)])
def test_open_cycle_full_batch(T, cfg):
    """BASELINE configs 2 and M: batched OpenProof, N=512/1024, (1,3,1), 4096 proofs."""
    N, n, k, l, B = cfg
    ctx, A, x, r, y, d, g = make(T, N, n, k, l, B, 5)
    c, t, ok = ctx.open_commit(x, r, y)
    z = ctx.open_response(y, r, d)
    acc = ctx.open_verify(z, t, c, d)
    assert int(ok.sum()) == B and int(acc.sum()) == B
    # Commitment::verify (commit.rs:173-210, f = None): [a1;a2].r + [0;x] == c, recomputed from Mat primitives
    zx = T.cat([T.zeros((B, n, N), dtype=T.int64, device=x.device), x], dim=1)
    assert bool(ctx.eq(ctx.matvec(2, r, zx), c).all())
    assert T.equal(ctx.matvec(0, y), t)
    # z = y + r (.) d from primitives
    assert T.equal(ctx.add(y, ctx.cmul(r, d)), z)
    # verifier relation from primitives: a1.z == t + c1 (.) d
    lhs = ctx.matvec(0, z)
    rhs = ctx.add(t, ctx.cmul(c[:, :n].contiguous(), d))
    assert T.equal(lhs, rhs)
    # soundness probes: every 7th response tampered, every 11th commitment tampered
    idx_z = T.arange(0, B, 7, device=z.device)
    idx_c = T.arange(3, B, 11, device=z.device)
    z2 = z.clone()
    z2[idx_z, k - 1, N - 1] += 1
    c2 = c.clone()
    c2[idx_c, 0, 0] = T.where(c2[idx_c, 0, 0] >= HALF, c2[idx_c, 0, 0] - 1, c2[idx_c, 0, 0] + 1)
    acc2 = ctx.open_verify(z2, t, c2, d).cpu().numpy()
    expect = np.ones(B, dtype=np.uint8)
    expect[idx_z.cpu().numpy()] = 0
    expect[idx_c.cpu().numpy()] = 0
    assert np.array_equal(acc2, expect)
    # a response that breaks the norm bound is rejected even though the linear relation could hold
    z3 = z.clone()
    z3[0, 0, :] = ctx.verify_bound  # norm way above 2*sigma*sqrt(N)
    assert int(ctx.open_verify(z3, t, c, d)[0]) == 0
    assert np.all(c.cpu().numpy() <= HALF) and np.all(c.cpu().numpy() >= -HALF)


def test_ring_laws_full_range(T):
    N, B = 1024, 2048
    ctx, A, x, r, y, d, g = make(T, N, 1, 3, 1, 8, 9)
    dev = x.device
    a = synth.t_uniform(g, (B, N), dev)
    b = synth.t_uniform(g, (B, N), dev)
    c = synth.t_uniform(g, (B, N), dev)
    ab = ctx.polymul(a, b)
    assert T.equal(ab, ctx.polymul(b, a))                                     # commutative
    assert T.equal(ctx.polymul(a, ctx.add(b, c)), ctx.add(ab, ctx.polymul(a, c)))   # distributive
    assert T.equal(ctx.polymul(ctx.polymul(a, b), c), ctx.polymul(a, ctx.polymul(b, c)))  # associative
    one = T.zeros((B, N), dtype=T.int64, device=dev)
    one[:, 0] = 1
    assert T.equal(ctx.polymul(a, one), a)
    xm = T.zeros((B, N), dtype=T.int64, device=dev)
    xm[:, 1] = 1                                                              # multiply by X: negacyclic shift
    ax = ctx.polymul(a, xm)
    assert T.equal(ax[:, 1:], a[:, :-1]) and T.equal(ax[:, 0], -a[:, -1])
    assert T.equal(ctx.sub(ctx.add(a, b), b), a)
    out = ab.cpu().numpy()
    assert out.max() <= HALF and out.min() >= -HALF


def test_linear_and_sum_completeness_batch(T):
    """BASELINE config 4 / 3 shapes at reduced batch: LinearProof N=1024 (1,3,1); SumProof N=1024 (4,9,4) V=8."""
    N = 1024
    ctx, A, x, r, y, d, g = make(T, N, 1, 3, 1, 256, 21)
    dev = x.device
    B, k = 256, 3
    gpoly = synth.t_uniform(g, (B, N), dev)
    rp = synth.t_small(g, (B, k, N), dev)
    yp = synth.t_gauss(g, (B, k, N), dev, ctx.sigma)
    c, cp, t, tp, u, ok = ctx.linear_commit(gpoly, x, r, rp, y, yp)
    z, zp = ctx.linear_response(y, yp, r, rp, d)
    acc = ctx.linear_verify(z, zp, c, cp, gpoly, t, tp, u, d)
    assert int(acc.sum()) == B and int((ok == 3).sum()) == B
    u2 = u.clone()
    u2[5, 0, 0] += 1
    assert ctx.linear_verify(z, zp, c, cp, gpoly, t, tp, u2, d).cpu().tolist() == [1] * 5 + [0] + [1] * (B - 6)

    from ring_zk_amd import Context

    n, k, l, V, B = 4, 9, 4, 8, 16
    ctx3 = Context(N, n, k, l)
    A3 = synth.t_key(g, N, n, k, l, dev)
    ctx3.load_key(A3)
    gs = synth.t_uniform(g, (B, V, N), dev)
    xs = synth.t_uniform(g, (B, V, l, N), dev)
    rs = synth.t_small(g, (B, V, k, N), dev)
    rp = synth.t_small(g, (B, k, N), dev)
    ys = synth.t_gauss(g, (B, V, k, N), dev, ctx3.sigma)
    yp = synth.t_gauss(g, (B, k, N), dev, ctx3.sigma)
    d3 = synth.t_challenge(g, B, N, ctx3.kappa, dev)
    cs, cp, ts, tp, u, ok = ctx3.sum_commit(gs, xs, rs, rp, ys, yp)
    zs, zp = ctx3.sum_response(ys, yp, rs, rp, d3)
    acc = ctx3.sum_verify(zs, zp, cs, cp, gs, ts, tp, u, d3)
    assert int(ok.sum()) == B and int(acc.sum()) == B
    zs2 = zs.clone()
    zs2[3, V - 1, k - 1, 0] += 1
    assert ctx3.sum_verify(zs2, zp, cs, cp, gs, ts, tp, u, d3).cpu().tolist() == [1, 1, 1, 0] + [1] * (B - 4)


def test_stress_shape_sum_2048(T):
    """BASELINE config 5 shape (N=2048, (8,17,8), V=32) on two proofs: completeness + one tampered summand."""
    from ring_zk_amd import Context

    N, n, k, l, V, B = 2048, 8, 17, 8, 32, 2
    dev = T.device("cuda", 0)
    g = T.Generator(device=dev)
    g.manual_seed(77)
    ctx = Context(N, n, k, l)
    ctx.load_key(synth.t_key(g, N, n, k, l, dev))
    gs = synth.t_uniform(g, (B, V, N), dev)
    xs = synth.t_uniform(g, (B, V, l, N), dev)
    rs = synth.t_small(g, (B, V, k, N), dev)
    rp = synth.t_small(g, (B, k, N), dev)
    ys = synth.t_gauss(g, (B, V, k, N), dev, ctx.sigma)
    yp = synth.t_gauss(g, (B, k, N), dev, ctx.sigma)
    d = synth.t_challenge(g, B, N, ctx.kappa, dev)
    cs, cp, ts, tp, u, ok = ctx.sum_commit(gs, xs, rs, rp, ys, yp)
    zs, zp = ctx.sum_response(ys, yp, rs, rp, d)
    assert ctx.sum_verify(zs, zp, cs, cp, gs, ts, tp, u, d).cpu().tolist() == [1, 1]
    ts2 = ts.clone()
    ts2[1, 17, 3, 100] += 1
    assert ctx.sum_verify(zs, zp, cs, cp, gs, ts2, tp, u, d).cpu().tolist() == [1, 0]
