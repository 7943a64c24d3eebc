"""ctypes wrapper around oracle/_build/librzk_oracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the product package ring_zk_amd.  See oracle/rzk_oracle.h for the reference
file:line each function follows and for the parity status ("parity unpinned" for products mod q).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librzk_oracle.so")

Q_DEFAULT = 3515337053  # ZqI64<3515337053>, src/params.rs:121


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Returns the .so path."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("rzk_oracle.c", "rzk_oracle.h"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _SO


class _P(C.Structure):
    _fields_ = [
        ("q", C.c_int64),
        ("N", C.c_uint32),
        ("n", C.c_uint32),
        ("k", C.c_uint32),
        ("l", C.c_uint32),
        ("kappa", C.c_uint32),
        ("b", C.c_uint64),
    ]


@dataclass(frozen=True)
class Params:
    """Mirror of Params<ZqI64<Q>> (src/params.rs:18-36) plus the ring degree N and modulus Q."""

    N: int
    n: int = 1
    k: int = 3
    l: int = 1
    kappa: int = 36
    b: int = 1
    q: int = Q_DEFAULT

    def c(self) -> _P:
        return _P(self.q, self.N, self.n, self.k, self.l, self.kappa, self.b)

    @property
    def sigma(self) -> int:
        return lib().rzko_sigma(self.b, self.kappa, self.k, self.N)

    @property
    def commit_bound(self) -> int:
        return lib().rzko_commit_bound(self.b, self.kappa, self.k, self.N)

    @property
    def verify_bound(self) -> int:
        return lib().rzko_verify_bound(self.b, self.kappa, self.k, self.N)


_lib = None
_I64P = C.POINTER(C.c_int64)
_U32P = C.POINTER(C.c_uint32)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.rzko_center.restype = C.c_int64
        L.rzko_center.argtypes = [C.c_int64, C.c_int64]
        L.rzko_isqrt_u64.restype = C.c_uint64
        L.rzko_isqrt_u64.argtypes = [C.c_uint64]
        for name in ("rzko_norm2", "rzko_norm1", "rzko_norm_inf"):
            f = getattr(L, name)
            f.restype = C.c_uint64
            f.argtypes = [C.c_uint32, _I64P]
        for name in ("rzko_sigma", "rzko_commit_bound", "rzko_verify_bound"):
            f = getattr(L, name)
            f.restype = C.c_uint64
            f.argtypes = [C.c_uint64] * 4
        L.rzko_check_norm.restype = C.c_int
        L.rzko_check_norm.argtypes = [C.c_uint32, C.c_uint32, _I64P, C.c_uint64]
        L.rzko_powmod.restype = C.c_uint32
        L.rzko_powmod.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32]
        L.rzko_open_cycle_batch.restype = C.c_int64
        L.rzko_hw_threads.restype = C.c_int
        _lib = L
    return _lib


def _a(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.int64)


def _p(x: np.ndarray):
    return x.ctypes.data_as(_I64P)


# ---- ring ops -----------------------------------------------------------------------------------
def poly_mul(a, b, q: int = Q_DEFAULT) -> np.ndarray:
    a, b = _a(a), _a(b)
    out = np.empty_like(a)
    lib().rzko_poly_mul(C.c_int64(q), C.c_uint32(a.shape[-1]), _p(a), _p(b), _p(out))
    return out


def poly_add(a, b, q: int = Q_DEFAULT) -> np.ndarray:
    a, b = _a(a), _a(b)
    out = np.empty_like(a)
    lib().rzko_poly_add(C.c_int64(q), C.c_uint32(a.shape[-1]), _p(a), _p(b), _p(out))
    return out


def poly_sub(a, b, q: int = Q_DEFAULT) -> np.ndarray:
    a, b = _a(a), _a(b)
    out = np.empty_like(a)
    lib().rzko_poly_sub(C.c_int64(q), C.c_uint32(a.shape[-1]), _p(a), _p(b), _p(out))
    return out


def center(v: int, q: int = Q_DEFAULT) -> int:
    return lib().rzko_center(v, q)


# ---- Mat ops (arrays shaped [m, n, N]) -------------------------------------------------------------
def mat_dot(A, B, q: int = Q_DEFAULT) -> np.ndarray:
    A, B = _a(A), _a(B)
    m, n, N = A.shape
    n2, p, _ = B.shape
    assert n == n2, "Mat::dot dimension mismatch (mat.rs:103)"
    out = np.empty((m, p, N), dtype=np.int64)
    lib().rzko_mat_dot(C.c_int64(q), C.c_uint32(N), C.c_uint32(m), C.c_uint32(n), C.c_uint32(p),
                       _p(A), _p(B), _p(out))
    return out


def mat_add(A, B, q: int = Q_DEFAULT) -> np.ndarray:
    A, B = _a(A), _a(B)
    assert A.shape == B.shape, "Mat::add dimension mismatch (mat.rs:129-130)"
    m, n, N = A.shape
    out = np.empty_like(A)
    lib().rzko_mat_add(C.c_int64(q), C.c_uint32(N), C.c_uint32(m), C.c_uint32(n), _p(A), _p(B), _p(out))
    return out


def mat_sub(A, B, q: int = Q_DEFAULT) -> np.ndarray:
    A, B = _a(A), _a(B)
    assert A.shape == B.shape, "Mat::sub dimension mismatch (mat.rs:154-155)"
    m, n, N = A.shape
    out = np.empty_like(A)
    lib().rzko_mat_sub(C.c_int64(q), C.c_uint32(N), C.c_uint32(m), C.c_uint32(n), _p(A), _p(B), _p(out))
    return out


def mat_cmul(A, elem, q: int = Q_DEFAULT) -> np.ndarray:
    A, elem = _a(A), _a(elem)
    m, n, N = A.shape
    out = np.empty_like(A)
    lib().rzko_mat_cmul(C.c_int64(q), C.c_uint32(N), C.c_uint32(m), C.c_uint32(n), _p(A), _p(elem), _p(out))
    return out


# ---- norms -------------------------------------------------------------------------------------------
def norm2(p) -> int:
    p = _a(p)
    return lib().rzko_norm2(p.shape[-1], _p(p))


def norm1(p) -> int:
    p = _a(p)
    return lib().rzko_norm1(p.shape[-1], _p(p))


def norm_inf(p) -> int:
    p = _a(p)
    return lib().rzko_norm_inf(p.shape[-1], _p(p))


def check_norm(polys, bound: int) -> bool:
    polys = _a(polys)
    N = polys.shape[-1]
    return bool(lib().rzko_check_norm(N, polys.size // N, _p(polys), bound))


# ---- commitment scheme & protocols (single proof; arrays [rows, N]) -------------------------------------
def key_build(P: Params, a1p, a2p) -> np.ndarray:
    a1p, a2p = _a(a1p), _a(a2p)
    A = np.empty((P.n + P.l, P.k, P.N), dtype=np.int64)
    c = P.c()
    lib().rzko_key_build(C.byref(c), _p(a1p), _p(a2p), _p(A))
    return A


def commit(P: Params, A, x, r):
    A, x, r = _a(A), _a(x), _a(r)
    c = np.empty((P.n + P.l, P.N), dtype=np.int64)
    cp = P.c()
    ok = lib().rzko_commit(C.byref(cp), _p(A), _p(x), _p(r), _p(c))
    return c, bool(ok)


def commitment_verify(P: Params, A, c, x, r, f=None) -> bool:
    """Commitment::verify (commit.rs:173-210); f = None or the scalar polynomial of the relaxed opening."""
    A, c, x, r = _a(A), _a(c), _a(x), _a(r)
    cp = P.c()
    if f is None:
        return bool(lib().rzko_commitment_verify(C.byref(cp), _p(A), _p(c), _p(x), _p(r)))
    f = _a(f)
    return bool(lib().rzko_commitment_verify_f(C.byref(cp), _p(A), _p(c), _p(x), _p(r), _p(f)))


def open_commit(P: Params, A, x, r, y):
    A, x, r, y = _a(A), _a(x), _a(r), _a(y)
    c = np.empty((P.n + P.l, P.N), dtype=np.int64)
    t = np.empty((P.n, P.N), dtype=np.int64)
    cp = P.c()
    ok = lib().rzko_open_commit(C.byref(cp), _p(A), _p(x), _p(r), _p(y), _p(c), _p(t))
    return c, t, bool(ok)


def open_response(P: Params, y, r, d):
    y, r, d = _a(y), _a(r), _a(d)
    z = np.empty((P.k, P.N), dtype=np.int64)
    cp = P.c()
    lib().rzko_open_response(C.byref(cp), _p(y), _p(r), _p(d), _p(z))
    return z


def open_verify(P: Params, A, z, t, c, d) -> int:
    A, z, t, c, d = _a(A), _a(z), _a(t), _a(c), _a(d)
    cp = P.c()
    return lib().rzko_open_verify(C.byref(cp), _p(A), _p(z), _p(t), _p(c), _p(d))


def linear_commit(P: Params, A, g, x, r, rp, y, yp):
    A, g, x, r, rp, y, yp = map(_a, (A, g, x, r, rp, y, yp))
    nl = P.n + P.l
    c = np.empty((nl, P.N), dtype=np.int64)
    cpm = np.empty((nl, P.N), dtype=np.int64)
    t = np.empty((P.n, P.N), dtype=np.int64)
    tp = np.empty((P.n, P.N), dtype=np.int64)
    u = np.empty((P.l, P.N), dtype=np.int64)
    cp = P.c()
    ok = lib().rzko_linear_commit(C.byref(cp), _p(A), _p(g), _p(x), _p(r), _p(rp), _p(y), _p(yp),
                                  _p(c), _p(cpm), _p(t), _p(tp), _p(u))
    return c, cpm, t, tp, u, ok


def linear_response(P: Params, y, yp, r, rp, d):
    y, yp, r, rp, d = map(_a, (y, yp, r, rp, d))
    z = np.empty((P.k, P.N), dtype=np.int64)
    zp = np.empty((P.k, P.N), dtype=np.int64)
    cp = P.c()
    lib().rzko_linear_response(C.byref(cp), _p(y), _p(yp), _p(r), _p(rp), _p(d), _p(z), _p(zp))
    return z, zp


def linear_verify(P: Params, A, z, zp, c, cpm, g, t, tp, u, d) -> int:
    A, z, zp, c, cpm, g, t, tp, u, d = map(_a, (A, z, zp, c, cpm, g, t, tp, u, d))
    cp = P.c()
    return lib().rzko_linear_verify(C.byref(cp), _p(A), _p(z), _p(zp), _p(c), _p(cpm), _p(g), _p(t),
                                    _p(tp), _p(u), _p(d))


def sum_commit(P: Params, A, gs, xs, rs, rp, ys, yp):
    A, gs, xs, rs, rp, ys, yp = map(_a, (A, gs, xs, rs, rp, ys, yp))
    V = gs.shape[0]
    nl = P.n + P.l
    cs = np.empty((V, nl, P.N), dtype=np.int64)
    cpm = np.empty((nl, P.N), dtype=np.int64)
    ts = np.empty((V, P.n, P.N), dtype=np.int64)
    tp = np.empty((P.n, P.N), dtype=np.int64)
    u = np.empty((P.l, P.N), dtype=np.int64)
    cp = P.c()
    ok = lib().rzko_sum_commit(C.byref(cp), C.c_uint32(V), _p(A), _p(gs), _p(xs), _p(rs), _p(rp),
                               _p(ys), _p(yp), _p(cs), _p(cpm), _p(ts), _p(tp), _p(u))
    return cs, cpm, ts, tp, u, bool(ok)


def sum_response(P: Params, ys, yp, rs, rp, d):
    ys, yp, rs, rp, d = map(_a, (ys, yp, rs, rp, d))
    V = ys.shape[0]
    zs = np.empty((V, P.k, P.N), dtype=np.int64)
    zp = np.empty((P.k, P.N), dtype=np.int64)
    cp = P.c()
    lib().rzko_sum_response(C.byref(cp), C.c_uint32(V), _p(ys), _p(yp), _p(rs), _p(rp), _p(d),
                            _p(zs), _p(zp))
    return zs, zp


def sum_verify(P: Params, A, zs, zp, cs, cpm, gs, ts, tp, u, d) -> int:
    A, zs, zp, cs, cpm, gs, ts, tp, u, d = map(_a, (A, zs, zp, cs, cpm, gs, ts, tp, u, d))
    V = gs.shape[0]
    cp = P.c()
    return lib().rzko_sum_verify(C.byref(cp), C.c_uint32(V), _p(A), _p(zs), _p(zp), _p(cs), _p(cpm),
                                 _p(gs), _p(ts), _p(tp), _p(u), _p(d))


# ---- auxiliary-prime NTT ---------------------------------------------------------------------------------
def ntt_forward(a, p: int, psi: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint32).copy()
    lib().rzko_ntt_forward(C.c_uint32(p), C.c_uint32(psi), C.c_uint32(a.shape[-1]), a.ctypes.data_as(_U32P))
    return a


def ntt_forward_batch(a, p: int, psi: int, threads: int = 0, inplace: bool = False) -> np.ndarray:
    """Table-driven forward transform of every row of a (count, N) on the host cores (bench.py: same-algorithm CPU leg)."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if not inplace:
        a = a.copy()
    fn = lib().rzko_ntt_forward_batch
    fn.restype = None
    fn(C.c_uint32(p), C.c_uint32(psi), C.c_uint32(a.shape[-1]), C.c_uint64(a.size // a.shape[-1]),
       a.ctypes.data_as(_U32P), C.c_int(threads))
    return a


def ntt_inverse(a, p: int, psi: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint32).copy()
    lib().rzko_ntt_inverse(C.c_uint32(p), C.c_uint32(psi), C.c_uint32(a.shape[-1]), a.ctypes.data_as(_U32P))
    return a


def powmod(b: int, e: int, p: int) -> int:
    return lib().rzko_powmod(b, e, p)


# ---- timed CPU baseline ---------------------------------------------------------------------------------------
def open_cycle_batch(P: Params, A, x, r, y, d, threads: int = 0) -> int:
    """Full OpenProof cycle for a batch on the host cores (bench.py cpu_baseline leg)."""
    A, x, r, y, d = map(_a, (A, x, r, y, d))
    B = d.shape[0]
    cp = P.c()
    L = lib()
    L.rzko_open_cycle_batch.argtypes = [C.POINTER(_P), C.c_uint32, _I64P, _I64P, _I64P, _I64P, _I64P, C.c_int]
    return L.rzko_open_cycle_batch(C.byref(cp), B, _p(A), _p(x), _p(r), _p(y), _p(d), threads)


def hw_threads() -> int:
    return lib().rzko_hw_threads()
