"""Contracts of the C ABI that no parity test exercises by itself (GPU):

  * trusted-producer mode (rzk_ctx_trust_device_outputs): default OFF rejects k*2^32 + s exactly as before; ON gives
    the same bytes on canonical data for every phase of the three protocols;
  * a coefficient of exactly 2^31 (the one value whose low word is INT32_MIN) in verifier operands WITHOUT a norm
    check — gs, cs, w-like inputs of sum_verify / linear_verify — and in z is rejected, not wrapped;
  * "one context per (GPU, host thread)" (include/rzk.h): two contexts of different shapes, (8,17,8)@2048 (row blocks:
    > 64 KiB dynamic LDS, per-launch hipFuncSetAttribute) and (1,3,1)@1024, used alternately from two host threads
    in one process, every result against the oracle;
  * rzk_prof_read_kernels names what ran.
Reference behaviour: ZqI64::from (src/params.rs:126), open.rs:162-174, sum.rs:257-320, linear.rs:213-250.
"""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from ring_zk_amd import synth

from test_gpu_baseline_shapes import P_of, _open_proof, dev, make_ctx, sum_inputs, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu
Q = O.Q_DEFAULT


@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_trusted_mode_same_bytes_default_still_checks(torch_mod, N):
    n, k, l, B, V = 1, 3, 1, 3, 2
    ctx = make_ctx(N, n, k, l)
    P = P_of(ctx)
    A, x, r, y, d, c, t, z = _open_proof(ctx, B, 4100 + N)
    zbad = z.copy()
    zbad[1, 1, 7] += 1 << 32
    assert ctx.open_verify(zbad, t, c, d).tolist() == [1, 0, 1]            # default: checked
    rng = np.random.default_rng(4200 + N)
    g = synth.uniform(rng, (B, N))
    rp, yp = synth.small(rng, (B, k, N)), synth.gauss(rng, (B, k, N), P.sigma)
    gs, xs, rs, rp2, ys, yp2, d2 = sum_inputs(rng, P, B, V)

    def everything():
        out = []
        c1, t1, ok1 = ctx.open_commit(x, r, y)
        z1 = ctx.open_response(y, r, d)
        out += [c1, t1, ok1, z1, ctx.open_verify(z1, t1, c1, d)]
        lc = ctx.linear_commit(g, x, r, rp, y, yp)
        lz = ctx.linear_response(y, yp, r, rp, d)
        out += list(lc) + list(lz) + [ctx.linear_verify(lz[0], lz[1], lc[0], lc[1], g, lc[2], lc[3], lc[4], d)]
        sc = ctx.sum_commit(gs, xs, rs, rp2, ys, yp2)
        sz = ctx.sum_response(ys, yp2, rs, rp2, d2)
        out += list(sc) + list(sz) + [ctx.sum_verify(sz[0], sz[1], sc[0], sc[1], gs, sc[2], sc[3], sc[4], d2)]
        out += [ctx.polymul(g, g), ctx.matvec(2, np.ascontiguousarray(ys[:, 0])), ctx.commit(x, r)[0]]
        return out

    checked = everything()
    ctx.trust_device_outputs(True)
    trusted = everything()
    ctx.trust_device_outputs(False)
    assert len(checked) == len(trusted)
    for i, (a, b) in enumerate(zip(checked, trusted)):
        assert np.array_equal(a, b), i
    assert checked[4].tolist() == [1] * B
    assert ctx.open_verify(zbad, t, c, d).tolist() == [1, 0, 1]            # and checked again after switching back
    # against the oracle once, in trusted mode
    ctx.trust_device_outputs(True)
    c1, t1, ok1 = ctx.open_commit(x, r, y)
    for b in range(B):
        c_ref, t_ref, ok_ref = O.open_commit(P, A, x[b], r[b], y[b])
        assert np.array_equal(c1[b], c_ref) and np.array_equal(t1[b], t_ref) and bool(ok1[b]) == ok_ref


@pytest.mark.parametrize("dkey", [1, 2, 0])
@pytest.mark.parametrize("N", [512, 2048])
def test_coefficient_of_exactly_two_to_31_is_rejected(torch_mod, N, dkey):
    """2^31 has the low word INT32_MIN and a zero high word after the +h shift: the canonical test must catch it through
    the low-word bound, in operands that carry no norm predicate as well (gs, cs of sum_verify; g, cp of
    linear_verify) and in z.  dkey = 2: the multipliers g / g_i go through dkey_transform_kernel (which then is the only
    place that sees their raw coefficients and must clear the verdict itself); 0: every row loads them."""
    n, k, l, B, V = 1, 3, 1, 3, 2
    ctx = make_ctx(N, n, k, l, env={"RZK_DKEY": dkey})
    P = P_of(ctx)
    rng = np.random.default_rng(4300 + N)
    A = synth.key(rng, N, n, k, l)
    ctx.load_key(A)
    gs, xs, rs, rp, ys, yp, d = sum_inputs(rng, P, B, V)
    cs, cp, ts, tp, u, ok = ctx.sum_commit(gs, xs, rs, rp, ys, yp)
    zs, zp = ctx.sum_response(ys, yp, rs, rp, d)
    assert ctx.sum_verify(zs, zp, cs, cp, gs, ts, tp, u, d).tolist() == [1] * B
    for name, arr, pos in (("zs", zs, (1, 0, 1, 3)), ("gs", gs, (2, 1, 0)), ("cs", cs, (0, 1, 1, N - 1)), ("cp", cp, (1, 1, 5)),
                           ("u", u, (2, 0, 9))):
        for val in (1 << 31, -(1 << 31)):
            m = arr.copy()
            m[pos] = val
            args = dict(zs=zs, zp=zp, cs=cs, cp=cp, gs=gs, ts=ts, tp=tp, u=u, d=d)
            args[name] = m
            acc = ctx.sum_verify(*(args[key] for key in ("zs", "zp", "cs", "cp", "gs", "ts", "tp", "u", "d")))
            want = [1] * B
            want[pos[0]] = 0
            assert acc.tolist() == want, (name, val)
    C = np.ascontiguousarray
    g, x0, r0, y0 = C(gs[:, 0]), C(xs[:, 0]), C(rs[:, 0]), C(ys[:, 0])
    lc = ctx.linear_commit(g, x0, r0, rp, y0, yp)
    lz = ctx.linear_response(y0, yp, r0, rp, d)
    assert ctx.linear_verify(lz[0], lz[1], lc[0], lc[1], g, lc[2], lc[3], lc[4], d).tolist() == [1] * B
    gb = g.copy()
    gb[1, 4] = 1 << 31
    assert ctx.linear_verify(lz[0], lz[1], lc[0], lc[1], gb, lc[2], lc[3], lc[4], d).tolist() == [1, 0, 1]
    cpb = lc[1].copy()
    cpb[2, 1, 0] = -(1 << 31)
    assert ctx.linear_verify(lz[0], lz[1], lc[0], cpb, g, lc[2], lc[3], lc[4], d).tolist() == [1, 1, 0]


def test_two_contexts_two_threads_alternating(torch_mod):
    """Two contexts of different shapes in one process, each driven by its own host thread, launches interleaved."""
    big = make_ctx(2048, 8, 17, 8)       # row blocks (138 KiB dynamic LDS), transform products for the challenge
    small = make_ctx(1024, 1, 3, 1)      # unit kernels, rotations
    rng = np.random.default_rng(4400)
    Ab, As = synth.key(rng, 2048, 8, 17, 8), synth.key(rng, 1024, 1, 3, 1)
    big.load_key(Ab)
    small.load_key(As)
    Pb, Ps = P_of(big), P_of(small)
    rounds = 3
    data = {}
    for name, ctx, P, B in (("big", big, Pb, 1), ("small", small, Ps, 3)):
        data[name] = [dict(x=synth.uniform(rng, (B, ctx.l, ctx.N)), r=synth.small(rng, (B, ctx.k, ctx.N)),
                           y=synth.gauss(rng, (B, ctx.k, ctx.N), P.sigma), d=synth.challenge(rng, (B,), ctx.N, P.kappa))
                      for _ in range(rounds)]
    results, errors = {"big": [], "small": []}, []
    turn = threading.Barrier(2)

    def worker(name, ctx):
        try:
            for i in range(rounds):
                I = data[name][i]
                turn.wait(timeout=300)               # both threads issue their launches at the same time
                c, t, ok = ctx.open_commit(I["x"], I["r"], I["y"])
                z = ctx.open_response(I["y"], I["r"], I["d"])
                acc = ctx.open_verify(z, t, c, I["d"])
                results[name].append((c, t, ok, z, acc))
        except Exception as e:   # noqa: BLE001 - reported by the main thread
            errors.append((name, repr(e)))
            try:
                turn.abort()
            except Exception:
                pass

    th = [threading.Thread(target=worker, args=("big", big)), threading.Thread(target=worker, args=("small", small))]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=600)
    assert not errors, errors
    for name, ctx, P, A in (("big", big, Pb, Ab), ("small", small, Ps, As)):
        assert len(results[name]) == rounds
        for i, (c, t, ok, z, acc) in enumerate(results[name]):
            I = data[name][i]
            for b in range(I["x"].shape[0]):
                c_ref, t_ref, ok_ref = O.open_commit(P, A, I["x"][b], I["r"][b], I["y"][b])
                assert np.array_equal(c[b], c_ref) and np.array_equal(t[b], t_ref) and bool(ok[b]) == ok_ref, (name, i, b)
                assert np.array_equal(z[b], O.open_response(P, I["y"][b], I["r"][b], I["d"][b])), (name, i, b)
            assert acc.tolist() == [1] * I["x"].shape[0]
    # and sequentially on one thread, the other way round (the dynamic-LDS attribute is per launch, not per process)
    I = data["small"][0]
    c, t, ok = small.open_commit(I["x"], I["r"], I["y"])
    assert np.array_equal(c, results["small"][0][0])
    I = data["big"][0]
    c, t, ok = big.open_commit(I["x"], I["r"], I["y"])
    assert np.array_equal(c, results["big"][0][0])


def test_prof_names_the_kernels(torch_mod):
    ctx = make_ctx(1024, 1, 3, 1)
    A, x, r, y, d, c, t, z = _open_proof(ctx, 2, 4500)
    D = lambda a: dev(torch_mod, a)
    ctx.prof_enable(True)
    ctx.prof_reset()
    c2, t2, ok2 = ctx.open_commit(D(x), D(r), D(y))
    z2 = ctx.open_response(D(y), D(r), D(d))
    acc = ctx.open_verify(z2, t2, c2, D(d))
    names = ctx.prof_read_kernels()
    durs = ctx.prof_read_all()
    ctx.prof_enable(False)
    assert acc.cpu().tolist() == [1, 1]
    assert [nm for nm, _ in names] == ["unit_kernel<10, false, false>", "shift_row_kernel<10, false>", "unit_kernel<10, false, true>"]
    # Open at (1,3,1): commit reads x, r(3), y(3) and stores c(2), t(1); response reads d, y(3), r(3), stores z(3);
    # verify reads z(3), t, c1, d
    assert [nb for _, nb in names] == [10 * 8 * 1024 * 2, 10 * 8 * 1024 * 2, 6 * 8 * 1024 * 2]
    assert len(durs) == 3 and all(v > 0 for v in durs)


@pytest.mark.parametrize("N", [512, 1024, 2048])
@pytest.mark.parametrize("in_kernel", [1, 0])
def test_verdict_flags_initialised_by_the_launch_or_by_a_fill(torch_mod, N, in_kernel):
    """Verdict flags start at "ok" either inside the unit kernels (one team per batch entry: RZK_UPT=64 forces that at a
    small batch, as batches >= 4096 have it) or by a fill launch in front (RZK_PRESET_IN_KERNEL=0): same verdicts, and a
    stale buffer never leaks through — the flag array is handed over full of garbage both times.
    check_commit_constraint / check_verify_constraint + the relation: params.rs:102-118, open.rs:167-173."""
    n, k, l, B = 1, 3, 1, 6
    ctx = make_ctx(N, n, k, l, env={"RZK_UPT": 64, "RZK_PRESET_IN_KERNEL": in_kernel})
    P = P_of(ctx)
    A, x, r, y, d, c, t, z = _open_proof(ctx, B, 4700 + N)
    # commit side: r of proof 2 beyond the commit bound (norm_2 <= 4 sigma floor(sqrt N), params.rs:102-108)
    rbad = r.copy()
    rbad[2, 1, :] = 5 * P.sigma
    want_ok = [int(O.open_commit(P, A, x[b], rbad[b], y[b])[2]) for b in range(B)]
    assert want_ok == [1, 1, 0, 1, 1, 1]
    torch = torch_mod
    for _ in range(2):   # device path twice into the same (dirty) verdict buffers
        okd = torch.full((B,), 0xAB, dtype=torch.uint8, device="cuda")
        cm, tt, okd2 = ctx.open_commit(dev(torch, x), dev(torch, rbad), dev(torch, y))
        assert okd2.cpu().numpy().tolist() == want_ok
        del okd
    assert ctx.open_commit(x, rbad, y)[2].tolist() == want_ok
    # verify side: z of proof 4 beyond the verify bound, proof 1 with a broken relation
    zbad = z.copy()
    zbad[4, 0, :] = 3 * P.sigma
    tb = t.copy()
    tb[1, 0, 3] = O.center(int(tb[1, 0, 3]) + 1)
    want = [int(O.open_verify(P, A, zbad[b], tb[b], c[b], d[b]) == 1) for b in range(B)]
    assert want == [1, 0, 1, 1, 0, 1]
    assert ctx.open_verify(zbad, tb, c, d).tolist() == want
    assert ctx.open_verify(*(dev(torch, v) for v in (zbad, tb, c, d))).cpu().numpy().tolist() == want
