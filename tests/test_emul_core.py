"""CPU lane-emulation of the kernel core (ring_zk_amd/csrc/rzk_core.h) against the oracle.

tests/emul/emul.cpp compiles the exact header the HIP kernels use with g++ and replays the 64 lanes
of a wavefront phase by phase.  This checks the register/LDS geometry, the twiddle indexing, the
Montgomery/lazy arithmetic and the CRT + mod-q centring bit-for-bit, without a GPU.  It is host
logic testing: the product path never loads this emulator.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(HERE, "emul")
Q = O.Q_DEFAULT
HALF = (Q - 1) // 2


def load_emul():
    """Builds (when stale) and loads the CPU emulator; also used by GPU tests that pin kernels to its functions."""
    so = os.path.join(EMUL_DIR, "libemul.so")
    srcs = [os.path.join(EMUL_DIR, "emul.cpp"),
            os.path.join(HERE, "..", "ring_zk_amd", "csrc", "rzk_rng.h"),
            os.path.join(HERE, "..", "ring_zk_amd", "csrc", "rzk_core.h"),
            os.path.join(HERE, "..", "ring_zk_amd", "csrc", "rzk_tables.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas",
                               "-o", so, srcs[0]])
    L = C.CDLL(so)
    L.emul_prime.restype = C.c_uint32
    L.emul_psi.restype = C.c_uint32
    L.emul_capacity.restype = C.c_double
    return L


@pytest.fixture(scope="module")
def emul():
    return load_emul()


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


@pytest.mark.parametrize("logn", [9, 10, 11])
@pytest.mark.parametrize("pi", [0, 1, 2])
def test_wave_ntt_matches_oracle_ntt(emul, logn, pi):
    N = 1 << logn
    p = emul.emul_prime(pi)
    psi = emul.emul_psi(pi, N)
    assert O.powmod(psi, N, p) == p - 1  # primitive 2N-th root of unity
    rng = np.random.default_rng(100 * logn + pi)
    a = rng.integers(0, p, N, dtype=np.uint32)
    a[:4] = [0, 1, p - 1, p - 2]
    out_std = np.empty(N, dtype=np.uint32)
    out_mem = np.empty(N, dtype=np.uint32)
    assert emul.emul_ntt_fwd(logn, pi, _u32p(a), _u32p(out_std), _u32p(out_mem)) == 0
    ref = O.ntt_forward(a, p, psi)
    assert np.array_equal(out_std, ref)
    # the global-memory ("RZK NTT") layout is a permutation of the standard order
    assert sorted(out_mem.tolist()) == sorted(ref.tolist())
    back = np.empty(N, dtype=np.uint32)
    assert emul.emul_ntt_inv(logn, pi, _u32p(out_std), _u32p(back)) == 0
    assert np.array_equal(back, a)
    assert np.array_equal(O.ntt_inverse(ref, p, psi), a)


@pytest.mark.parametrize("pi", [0, 1, 2])
def test_two_wavefront_team_ntt_matches_oracle_and_one_wave_layout(emul, pi):
    """N = 2048 with a team of two wavefronts (Geo<11, 7>: 16 coefficients per thread, code 1011 in the emulator):
    same transform, and the SAME words of the global NTT layout as the one-wavefront geometry — the resident key is
    stored once and read by both."""
    logn, N = 11, 2048
    p = emul.emul_prime(pi)
    psi = emul.emul_psi(pi, N)
    rng = np.random.default_rng(1011 + pi)
    a = rng.integers(0, p, N, dtype=np.uint32)
    a[:4] = [0, 1, p - 1, p - 2]
    std1, mem1 = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.uint32)
    std2, mem2 = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.uint32)
    assert emul.emul_ntt_fwd(logn, pi, _u32p(a), _u32p(std1), _u32p(mem1)) == 0
    assert emul.emul_ntt_fwd(1011, pi, _u32p(a), _u32p(std2), _u32p(mem2)) == 0
    ref = O.ntt_forward(a, p, psi)
    assert np.array_equal(std2, ref) and np.array_equal(std1, ref)
    assert np.array_equal(mem1, mem2)
    back = np.empty(N, dtype=np.uint32)
    assert emul.emul_ntt_inv(1011, pi, _u32p(std2), _u32p(back)) == 0
    assert np.array_equal(back, a)


def _polymul(emul, logn, np_, a, b):
    out = np.empty(1 << logn, dtype=np.int64)
    a = np.ascontiguousarray(a, dtype=np.int64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    assert emul.emul_polymul(logn, np_, C.c_uint64(Q), _i64p(a), _i64p(b), _i64p(out)) == 0
    return out


@pytest.mark.parametrize("logn", [9, 10, 11])
def test_emulated_polymul_full_range(emul, logn, golden):
    N = 1 << logn
    rng = np.random.default_rng(logn)
    a = rng.integers(-HALF, HALF + 1, N, dtype=np.int64)
    b = rng.integers(-HALF, HALF + 1, N, dtype=np.int64)
    assert np.array_equal(_polymul(emul, logn, 3, a, b), O.poly_mul(a, b))
    for case in golden["extreme_products"] + golden["random_products"]:
        if case["N"] == N:
            assert _polymul(emul, logn, 3, case["a"], case["b"]).tolist() == case["out"], case.get("name")


@pytest.mark.parametrize("logn", [9, 10])
def test_emulated_polymul_fewer_primes(emul, logn):
    """Small operands fit one or two primes: |result| <= ||a||_1 * ||b||_inf < capacity(np)."""
    N = 1 << logn
    rng = np.random.default_rng(77 + logn)
    r = rng.integers(-1, 2, N, dtype=np.int64)           # ternary, like the commitment randomness
    d = np.zeros(N, dtype=np.int64)                      # kappa-sparse challenge
    pos = rng.choice(N, 36, replace=False)
    d[pos] = rng.choice([-1, 1], 36)
    assert 36 * 1 < emul.emul_capacity(1)
    assert np.array_equal(_polymul(emul, logn, 1, r, d), O.poly_mul(r, d))
    k = rng.integers(-HALF, HALF + 1, N, dtype=np.int64)  # full-range key entry
    assert N * HALF < emul.emul_capacity(2)
    assert np.array_equal(_polymul(emul, logn, 2, k, r), O.poly_mul(k, r))
    y = np.trunc(rng.normal(0, 21780, N)).astype(np.int64)
    assert float(np.abs(y).sum()) * HALF < emul.emul_capacity(2)
    assert np.array_equal(_polymul(emul, logn, 2, k, y), O.poly_mul(k, y))
    # boundary: values that land exactly on +/- (P-1)/2-ish are covered by the extreme-product test;
    # here check negative results with a single prime
    a = np.zeros(N, dtype=np.int64)
    a[0] = -5
    b = np.zeros(N, dtype=np.int64)
    b[N - 1] = 7
    b[0] = 3
    assert np.array_equal(_polymul(emul, logn, 1, a, b), O.poly_mul(a, b))


@pytest.mark.parametrize("pair", [1, 0])
@pytest.mark.parametrize("logn", [9, 10, 11, 1011])
def test_emulated_shift_product(emul, logn, pair):
    """Challenge products as signed negacyclic rotations (ShiftGeo in rzk_core.h), one and two passes
    (1011: N = 2048 in the layout of a two-wavefront team)."""
    N = 1 << (logn % 1000)
    rng = np.random.default_rng(500 + logn)

    def run(passes, d, v):
        out = np.empty(N, dtype=np.int64)
        d = np.ascontiguousarray(d, dtype=np.int64)
        v = np.ascontiguousarray(v, dtype=np.int64)
        assert emul.emul_shift_product(logn, pair, passes, C.c_uint64(Q), _i64p(d), _i64p(v), _i64p(out)) == 0
        return out

    d = np.zeros(N, dtype=np.int64)
    pos = rng.choice(N, 36, replace=False)
    d[pos] = rng.choice([-1, 1], 36)
    d[0], d[1], d[N - 1], d[N - 2] = 1, -1, -1, 1            # both parities at both ends
    r = rng.integers(-1, 2, N, dtype=np.int64)
    assert np.array_equal(run(1, d, r), O.poly_mul(d, r))
    c1 = rng.integers(-HALF, HALF + 1, N, dtype=np.int64)     # full-range commitment row
    c1[:2] = [HALF, -HALF]
    for passes in (1, 2):
        assert np.array_equal(run(passes, d, c1), O.poly_mul(d, c1))
    dd = rng.integers(-HALF, HALF + 1, N, dtype=np.int64)     # dense full-range multiplier: sums up to 2^72
    dd[:2] = [-HALF, HALF]
    assert np.array_equal(run(2, dd, c1), O.poly_mul(dd, c1))
    small = rng.integers(-1000, 1001, N, dtype=np.int64)
    assert np.array_equal(run(1, small, c1), O.poly_mul(small, c1))


def test_philox_known_answers(emul):
    """Philox4x32-10 vectors of the Random123 distribution (kat_vectors): the generator behind the samplers."""
    kats = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kats:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        out = (C.c_uint32 * 4)()
        emul.emul_philox(c, k, out)
        assert tuple(out) == want
    emul.emul_uniform_below.restype = C.c_uint32
    emul.emul_uniform_below.argtypes = [C.c_uint32] * 3
    assert emul.emul_uniform_below(0, 0, 3) == 0
    assert emul.emul_uniform_below(0xffffffff, 0xffffffff, 3) == 2
    assert emul.emul_uniform_below(0x80000000, 0, 3) == 1
    assert emul.emul_uniform_below(0xffffffff, 0xffffffff, 0xffffffff) == 0xfffffffe
    rng = np.random.default_rng(9)
    for _ in range(200):
        hi, lo, rg = (int(v) for v in rng.integers(0, 2 ** 32, 3))
        rg = max(rg, 1)
        assert emul.emul_uniform_below(hi, lo, rg) == (((hi << 32) | lo) * rg) >> 64
