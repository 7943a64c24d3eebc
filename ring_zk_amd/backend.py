"""Thin Python host layer over the C ABI (include/rzk.h).

`Context` mirrors `Params<ZqI64<Q>>` + const generic `N` of the reference (src/params.rs:18-36) and
owns one `rzk_ctx`.  Every method takes either numpy arrays (host pointers -> the synchronous
`rzk_*_batch` entry points) or torch CUDA tensors (device pointers -> the asynchronous `*_dev` entry
points on torch's current stream).  PyTorch is used only for device memory and streams; all
arithmetic happens in the HIP library.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib

Q_DEFAULT = 3515337053  # ZqI64<3515337053>, src/params.rs:121


class RzkError(RuntimeError):
    """Non-zero status from the C ABI (the Rust shim turns it into panic!)."""

    def __init__(self, status: int, msg: str):
        super().__init__(f"rzk status {status}: {msg}")
        self.status = status


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class Context:
    def __init__(self, N: int, n: int = 1, k: int = 3, l: int = 1, kappa: int = 36, b: int = 1,
                 q: int = Q_DEFAULT, device: int = 0):
        self._L = _lib.lib()
        self.N, self.n, self.k, self.l, self.kappa, self.b, self.q = N, n, k, l, kappa, b, q
        self.device = device
        h = C.c_void_p()
        st = self._L.rzk_ctx_create(C.byref(h), q, N, n, k, l, kappa, b, device)
        if st != 0:
            raise RzkError(st, "rzk_ctx_create: " + self._L.rzk_last_error(None).decode())
        self._h = h
        self._stream = None
        self.half = (q - 1) // 2

    # ---- plumbing --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.rzk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int):
        if st != 0:
            raise RzkError(st, self._L.rzk_last_error(self._h).decode())

    def synchronize(self):
        self._check(self._L.rzk_ctx_synchronize(self._h))

    @property
    def sigma(self) -> int:
        return self._L.rzk_sigma(self._h)

    @property
    def commit_bound(self) -> int:
        return self._L.rzk_commit_bound(self._h)

    @property
    def verify_bound(self) -> int:
        return self._L.rzk_verify_bound(self._h)

    def _bind_torch_stream(self):
        import torch

        s = torch.cuda.current_stream(self.device).cuda_stream   # 0 = HIP's default stream
        if s != self._stream:
            self._check(self._L.rzk_ctx_set_stream(self._h, C.c_void_p(s)))
            self._stream = s

    def _prep(self, arrs: Sequence, dtypes: Sequence):
        """Validate inputs; returns (is_device, pointers)."""
        present = [a for a in arrs if a is not None]
        dev = _is_torch(present[0])
        ptrs = []
        for a, dt in zip(arrs, dtypes):
            if a is None:
                ptrs.append(None)
                continue
            if _is_torch(a) != dev:
                raise ValueError("mixing host (numpy) and device (torch) buffers in one call")
            if dev:
                import torch

                want = {np.int64: torch.int64, np.uint32: torch.int32, np.uint8: torch.uint8}[dt]
                if a.dtype != want or not a.is_cuda or not a.is_contiguous():
                    raise ValueError("device buffers must be contiguous CUDA tensors of the right dtype")
                ptrs.append(C.c_void_p(a.data_ptr()))
            else:
                if a.dtype != dt or not a.flags["C_CONTIGUOUS"]:
                    raise ValueError("host buffers must be C-contiguous numpy arrays of the right dtype")
                ptrs.append(C.c_void_p(a.ctypes.data))
        if dev:
            self._bind_torch_stream()
        return dev, ptrs

    def _empty(self, like, shape, dtype=np.int64):
        if _is_torch(like):
            import torch

            tdt = {np.int64: torch.int64, np.uint32: torch.int32, np.uint8: torch.uint8}[dtype]
            return torch.empty(shape, dtype=tdt, device=like.device)
        return np.empty(shape, dtype=dtype)

    def _fn(self, name: str, dev: bool):
        return getattr(self._L, name + ("_dev" if dev else ""))

    def _shape(self, a, *tail):
        if tuple(a.shape[-len(tail):]) != tuple(tail):
            raise ValueError(f"shape mismatch: expected trailing dims {tail}, got {tuple(a.shape)}")
        return int(np.prod(a.shape[:-len(tail)], dtype=np.int64)) if a.ndim > len(tail) else 1

    # ---- key ---------------------------------------------------------------------------------------
    def generate_key(self, seed: int) -> np.ndarray:
        """CommitmentKey::new (commit.rs:33-60) from the device-side sampler; loads the key and returns it."""
        a = np.empty((self.n + self.l, self.k, self.N), dtype=np.int64)
        self._check(self._L.rzk_key_generate(self._h, seed, C.c_void_p(a.ctypes.data)))
        return a

    def load_key(self, A):
        """CommitmentKey as the dense matrix [a1;a2] ((n+l) x k polynomials; src/commit.rs:109-114)."""
        self._shape(A, self.n + self.l, self.k, self.N)
        dev, (p,) = self._prep([A], [np.int64])
        self._check(self._fn("rzk_key_load", dev)(self._h, p))

    # ---- Mat seam ------------------------------------------------------------------------------------
    def polymul(self, a, b):
        cnt = self._shape(a, self.N)
        if self._shape(b, self.N) != cnt:
            raise ValueError("polymul: operand counts differ")
        out = self._empty(a, tuple(a.shape))
        dev, p = self._prep([a, b, out], [np.int64] * 3)
        self._check(self._fn("rzk_polymul_batch", dev)(self._h, p[0], p[1], p[2], cnt))
        return out

    def _rows(self, which: int) -> int:
        return {_lib.KEY_A1: self.n, _lib.KEY_A2: self.l, _lib.KEY_A: self.n + self.l}[which]

    def matvec(self, which: int, v, addend=None):
        """Mat::dot with the key on the left (src/mat.rs:95-115), optional `.add(&z)` (commit.rs:125)."""
        B = self._shape(v, self.k, self.N)
        rows = self._rows(which)
        if addend is not None and self._shape(addend, rows, self.N) != B:
            raise ValueError("matvec: addend batch differs")
        out = self._empty(v, tuple(v.shape[:-2]) + (rows, self.N))
        dev, p = self._prep([v, addend, out], [np.int64] * 3)
        self._check(self._fn("rzk_matvec_batch", dev)(self._h, which, p[0], p[1], p[2], B))
        return out

    def cmul(self, m, p_):
        """Mat::componentwise_mul (src/mat.rs:168-178): m [B, rows, N], p [B, N]."""
        rows = int(m.shape[-2])
        B = self._shape(m, rows, self.N)
        if self._shape(p_, self.N) != B:
            raise ValueError("cmul: batch differs")
        out = self._empty(m, tuple(m.shape))
        dev, p = self._prep([m, p_, out], [np.int64] * 3)
        self._check(self._fn("rzk_cmul_batch", dev)(self._h, p[0], rows, p[1], p[2], B))
        return out

    def _addsub(self, name, a, b):
        if tuple(a.shape) != tuple(b.shape):
            raise ValueError("Mat::add / Mat::sub dimension mismatch (mat.rs:129-130, 154-155)")
        cnt = self._shape(a, self.N)
        out = self._empty(a, tuple(a.shape))
        dev, p = self._prep([a, b, out], [np.int64] * 3)
        self._check(self._fn(name, dev)(self._h, p[0], p[1], p[2], cnt))
        return out

    def add(self, a, b):
        return self._addsub("rzk_add_batch", a, b)

    def sub(self, a, b):
        return self._addsub("rzk_sub_batch", a, b)

    def canonicalize(self, a):
        """Any int64 coefficients -> centred residues mod q (the other calls require canonical inputs)."""
        cnt = self._shape(a, self.N)
        out = self._empty(a, tuple(a.shape))
        dev, p = self._prep([a, out], [np.int64] * 2)
        self._check(self._fn("rzk_canonicalize_batch", dev)(self._h, p[0], p[1], cnt))
        return out

    def norm2_le(self, v, bound: int):
        rows = int(v.shape[-2])
        B = self._shape(v, rows, self.N)
        ok = self._empty(v, (B,), np.uint8)
        dev, p = self._prep([v, ok], [np.int64, np.uint8])
        self._check(self._fn("rzk_norm2_le_batch", dev)(self._h, p[0], rows, bound, p[1], B))
        return ok

    def eq(self, a, b):
        if tuple(a.shape) != tuple(b.shape):
            raise ValueError("eq: shape mismatch")
        rows = int(a.shape[-2])
        B = self._shape(a, rows, self.N)
        out = self._empty(a, (B,), np.uint8)
        dev, p = self._prep([a, b, out], [np.int64, np.int64, np.uint8])
        self._check(self._fn("rzk_eq_batch", dev)(self._h, p[0], p[1], rows, p[2], B))
        return out

    # ---- transforms -------------------------------------------------------------------------------------
    def ntt_forward(self, prime: int, x):
        cnt = self._shape(x, self.N)
        out = self._empty(x, tuple(x.shape), np.uint32)
        dev, p = self._prep([x, out], [np.uint32] * 2)
        self._check(self._fn("rzk_ntt_forward_batch", dev)(self._h, prime, p[0], p[1], cnt))
        return out

    def ntt_inverse(self, prime: int, x):
        cnt = self._shape(x, self.N)
        out = self._empty(x, tuple(x.shape), np.uint32)
        dev, p = self._prep([x, out], [np.uint32] * 2)
        self._check(self._fn("rzk_ntt_inverse_batch", dev)(self._h, prime, p[0], p[1], cnt))
        return out

    def ntt_prime(self, prime: int) -> int:
        return self._L.rzk_ntt_prime(prime)

    def ntt_psi(self, prime: int) -> int:
        return self._L.rzk_ntt_psi(prime, self.N)

    def ntt_layout(self) -> np.ndarray:
        """perm[j] = position in the library's NTT-domain layout of standard (bit-reversed) element j."""
        return np.array([self._L.rzk_ntt_layout_index(self.N, j) for j in range(self.N)], dtype=np.int64)

    def bench_ntt_forward(self, prime: int, x, out, iters: int) -> float:
        """Average duration (us) of one batched forward-NTT launch, HIP events on the launch stream."""
        cnt = self._shape(x, self.N)
        dev, p = self._prep([x, out], [np.uint32] * 2)
        if not dev:
            raise ValueError("bench_ntt_forward needs device buffers")
        us = self._L.rzk_bench_ntt_forward_dev(self._h, prime, p[0], p[1], cnt, iters)
        if us < 0:
            raise RzkError(int(us), self._L.rzk_last_error(self._h).decode())
        return us

    # ---- profiling of the row kernel (HIP events on the launch stream) ----------------------------------------
    def prof_enable(self, on: bool = True):
        self._check(self._L.rzk_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        self._check(self._L.rzk_prof_reset(self._h))

    def prof_read(self):
        us, cnt = C.c_double(), C.c_uint64()
        self._check(self._L.rzk_prof_read(self._h, C.byref(us), C.byref(cnt)))
        return us.value, cnt.value

    def prof_count(self) -> int:
        """Launches recorded so far (no synchronisation)."""
        return int(self._L.rzk_prof_count(self._h))

    def prof_read_all(self):
        """Durations (us) of the recorded launches, in launch order."""
        n = self.prof_count()
        buf = (C.c_double * max(n, 1))()
        cnt = C.c_size_t(0)
        self._check(self._L.rzk_prof_read_all(self._h, buf, n, C.byref(cnt)))
        return [buf[i] for i in range(min(n, cnt.value))]

    def prof_read_kernels(self):
        """[(kernel template instance, algorithmic bytes)] of the recorded launches, in launch order."""
        need = C.c_size_t(0)
        self._check(self._L.rzk_prof_read_kernels(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value + 1)
        self._check(self._L.rzk_prof_read_kernels(self._h, buf, need.value + 1, C.byref(need)))
        out = []
        for line in buf.value.decode().splitlines():
            name, nbytes = line.rsplit("\t", 1)
            out.append((name, int(nbytes)))
        return out

    def trust_device_outputs(self, on=True):
        """Trusted-producer mode (rzk_ctx_trust_device_outputs): skip the canonical test of loaded coefficients."""
        self._check(self._L.rzk_ctx_trust_device_outputs(self._h, 1 if on else 0))

    # ---- device-side samplers (statistical parity with src/polynomial.rs:14-44, src/challenge_space.rs:12-33) ----
    def _sample_out(self, lead):
        import torch

        self._bind_torch_stream()
        shape = tuple(lead) + (self.N,)
        out = torch.empty(shape, dtype=torch.int64, device=torch.device("cuda", self.device))
        return out, int(np.prod(shape[:-1], dtype=np.int64))

    def sample_uniform(self, seed: int, stream: int, bound: int, lead):
        """[*lead][N] coefficients uniform in [-bound, bound] (random_polynomial_within)."""
        out, cnt = self._sample_out(lead)
        self._check(self._L.rzk_sample_uniform_dev(self._h, seed, stream, bound, C.c_void_p(out.data_ptr()), cnt))
        return out

    def sample_gauss(self, seed: int, stream: int, sigma: float, lead):
        """[*lead][N] coefficients (i64) N(0, sigma) (random_polynomial_in_normal_distribution)."""
        out, cnt = self._sample_out(lead)
        self._check(self._L.rzk_sample_gauss_dev(self._h, seed, stream, float(sigma), C.c_void_p(out.data_ptr()), cnt))
        return out

    def sample_challenge(self, seed: int, stream: int, lead):
        """[*lead][N] challenges: kappa coefficients +-1 (random_polynomial_from_challenge_set)."""
        out, cnt = self._sample_out(lead)
        self._check(self._L.rzk_sample_challenge_dev(self._h, seed, stream, C.c_void_p(out.data_ptr()), cnt))
        return out

    # ---- commitment scheme (src/commit.rs) --------------------------------------------------------------------
    def commit(self, x, r):
        """CommitmentKey::commit (commit.rs:88-128) with caller-supplied r: (c, ok)."""
        B = self._shape(x, self.l, self.N)
        if self._shape(r, self.k, self.N) != B:
            raise ValueError("commit: batch / shape mismatch (commit.rs:95)")
        lead = tuple(x.shape[:-2])
        c = self._empty(x, lead + (self.n + self.l, self.N))
        ok = self._empty(x, (B,), np.uint8)
        dev, p = self._prep([x, r, c, ok], [np.int64] * 3 + [np.uint8])
        self._check(self._fn("rzk_commit_batch", dev)(self._h, *p, B))
        return c, ok

    def commitment_verify(self, c, x, r, f=None):
        """Commitment::verify (commit.rs:173-210); f = None or one scalar polynomial per opening."""
        B = self._shape(c, self.n + self.l, self.N)
        if self._shape(x, self.l, self.N) != B or self._shape(r, self.k, self.N) != B:
            raise ValueError("commitment_verify: batch / shape mismatch")
        if f is not None and self._shape(f, self.N) != B:
            raise ValueError("commitment_verify: one f per opening")
        ok = self._empty(c, (B,), np.uint8)
        dev, p = self._prep([c, x, r, f, ok], [np.int64] * 4 + [np.uint8])
        self._check(self._fn("rzk_commitment_verify_batch", dev)(self._h, *p, B))
        return ok

    # ---- OpenProof (src/prove/open.rs) -----------------------------------------------------------------------
    def open_commit(self, x, r, y):
        B = self._shape(x, self.l, self.N)
        if self._shape(r, self.k, self.N) != B or self._shape(y, self.k, self.N) != B:
            raise ValueError("open_commit: batch / shape mismatch (commit.rs:95)")
        lead = tuple(x.shape[:-2])
        c = self._empty(x, lead + (self.n + self.l, self.N))
        t = self._empty(x, lead + (self.n, self.N))
        ok = self._empty(x, (B,), np.uint8)
        dev, p = self._prep([x, r, y, c, t, ok], [np.int64] * 5 + [np.uint8])
        self._check(self._fn("rzk_open_commit_batch", dev)(self._h, *p, B))
        return c, t, ok

    def open_response(self, y, r, d):
        B = self._shape(y, self.k, self.N)
        if self._shape(r, self.k, self.N) != B or self._shape(d, self.N) != B:
            raise ValueError("open_response: batch / shape mismatch")
        z = self._empty(y, tuple(y.shape))
        dev, p = self._prep([y, r, d, z], [np.int64] * 4)
        self._check(self._fn("rzk_open_response_batch", dev)(self._h, *p, B))
        return z

    def open_verify(self, z, t, c, d):
        B = self._shape(z, self.k, self.N)
        if (self._shape(t, self.n, self.N) != B or self._shape(c, self.n + self.l, self.N) != B
                or self._shape(d, self.N) != B):
            raise ValueError("open_verify: batch / shape mismatch")
        acc = self._empty(z, (B,), np.uint8)
        dev, p = self._prep([z, t, c, d, acc], [np.int64] * 4 + [np.uint8])
        self._check(self._fn("rzk_open_verify_batch", dev)(self._h, *p, B))
        return acc

    # ---- LinearProof (src/prove/linear.rs) ------------------------------------------------------------------------
    def linear_commit(self, g, x, r, rp, y, yp):
        B = self._shape(x, self.l, self.N)
        for a, rows in ((r, self.k), (rp, self.k), (y, self.k), (yp, self.k)):
            if self._shape(a, rows, self.N) != B:
                raise ValueError("linear_commit: batch / shape mismatch")
        if self._shape(g, self.N) != B:
            raise ValueError("linear_commit: batch / shape mismatch")
        lead = tuple(x.shape[:-2])
        c = self._empty(x, lead + (self.n + self.l, self.N))
        cp = self._empty(x, lead + (self.n + self.l, self.N))
        t = self._empty(x, lead + (self.n, self.N))
        tp = self._empty(x, lead + (self.n, self.N))
        u = self._empty(x, lead + (self.l, self.N))
        ok = self._empty(x, (B,), np.uint8)
        dev, p = self._prep([g, x, r, rp, y, yp, c, cp, t, tp, u, ok], [np.int64] * 11 + [np.uint8])
        self._check(self._fn("rzk_linear_commit_batch", dev)(self._h, *p, B))
        return c, cp, t, tp, u, ok

    def linear_response(self, y, yp, r, rp, d):
        B = self._shape(y, self.k, self.N)
        z = self._empty(y, tuple(y.shape))
        zp = self._empty(y, tuple(y.shape))
        dev, p = self._prep([y, yp, r, rp, d, z, zp], [np.int64] * 7)
        self._check(self._fn("rzk_linear_response_batch", dev)(self._h, *p, B))
        return z, zp

    def linear_verify(self, z, zp, c, cp, g, t, tp, u, d):
        B = self._shape(z, self.k, self.N)
        acc = self._empty(z, (B,), np.uint8)
        dev, p = self._prep([z, zp, c, cp, g, t, tp, u, d, acc], [np.int64] * 9 + [np.uint8])
        self._check(self._fn("rzk_linear_verify_batch", dev)(self._h, *p, B))
        return acc

    # ---- SumProof (src/prove/sum.rs) --------------------------------------------------------------------------------
    def sum_commit(self, gs, xs, rs, rp, ys, yp):
        V = int(gs.shape[-2])
        if V == 0:
            raise ValueError("sum_commit: gs must not be empty (sum.rs:105)")
        B = self._shape(gs, V, self.N)
        if (self._shape(xs, V, self.l, self.N) != B or self._shape(rs, V, self.k, self.N) != B
                or self._shape(ys, V, self.k, self.N) != B or self._shape(rp, self.k, self.N) != B
                or self._shape(yp, self.k, self.N) != B):
            raise ValueError("sum_commit: gs.len() != xs.len() or shape mismatch (sum.rs:105)")
        lead = tuple(gs.shape[:-2])
        cs = self._empty(gs, lead + (V, self.n + self.l, self.N))
        cp = self._empty(gs, lead + (self.n + self.l, self.N))
        ts = self._empty(gs, lead + (V, self.n, self.N))
        tp = self._empty(gs, lead + (self.n, self.N))
        u = self._empty(gs, lead + (self.l, self.N))
        ok = self._empty(gs, (B,), np.uint8)
        dev, p = self._prep([gs, xs, rs, rp, ys, yp, cs, cp, ts, tp, u, ok], [np.int64] * 11 + [np.uint8])
        self._check(self._fn("rzk_sum_commit_batch", dev)(self._h, V, *p, B))
        return cs, cp, ts, tp, u, ok

    def sum_response(self, ys, yp, rs, rp, d):
        V = int(ys.shape[-3])
        B = self._shape(ys, V, self.k, self.N)
        zs = self._empty(ys, tuple(ys.shape))
        zp = self._empty(yp, tuple(yp.shape))
        dev, p = self._prep([ys, yp, rs, rp, d, zs, zp], [np.int64] * 7)
        self._check(self._fn("rzk_sum_response_batch", dev)(self._h, V, *p, B))
        return zs, zp

    def sum_verify(self, zs, zp, cs, cp, gs, ts, tp, u, d):
        V = int(zs.shape[-3])
        B = self._shape(zs, V, self.k, self.N)
        acc = self._empty(zs, (B,), np.uint8)
        dev, p = self._prep([zs, zp, cs, cp, gs, ts, tp, u, d, acc], [np.int64] * 9 + [np.uint8])
        self._check(self._fn("rzk_sum_verify_batch", dev)(self._h, V, *p, B))
        return acc
