"""bincode layout of the reference's `Mat<I, N>` <-> dense int64 slabs (host only; include/rzk.h "wire format").

The reference derives serde on `Mat` (src/mat.rs:11-14) and on the protocol messages, and round-trips them
with bincode's default options in its own test (src/mat.rs:424-438).  This module lets serialized
commitments / responses be turned into the `[rows][cols][N]` slabs the batched entry points take, and back.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib


def mat_encode(slab: np.ndarray, coef_bytes: int = 8) -> bytes:
    """slab: int64 [rows][cols][N] -> bincode bytes (polynomials trimmed of trailing zeros)."""
    slab = np.ascontiguousarray(slab, dtype=np.int64)
    if slab.ndim != 3:
        raise ValueError("expected a [rows][cols][N] array")
    rows, cols, N = slab.shape
    L = _lib.lib()
    size = L.rzk_wire_mat_size(C.c_void_p(slab.ctypes.data), rows, cols, N, coef_bytes)
    if size == 0:
        raise ValueError("coef_bytes must be 4 or 8")
    out = np.empty(size, dtype=np.uint8)
    written = C.c_size_t(0)
    rc = L.rzk_wire_mat_encode(C.c_void_p(slab.ctypes.data), rows, cols, N, coef_bytes, C.c_void_p(out.ctypes.data),
                               size, C.byref(written))
    if rc != _lib.RZK_OK:
        raise ValueError("a coefficient does not fit the requested width")
    return out[:written.value].tobytes()


def mat_decode(data: bytes, N: int, coef_bytes: int = 8, q: int = 0) -> Tuple[np.ndarray, int]:
    """bincode bytes -> (int64 [rows][cols][N] slab, bytes consumed).  Raises ValueError on malformed input and,
    with q > 0 (a message over ZqI64<q>), on any coefficient outside the centred range mod q."""
    buf = np.frombuffer(data, dtype=np.uint8)
    L = _lib.lib()
    rows, cols, used = C.c_uint32(0), C.c_uint32(0), C.c_size_t(0)
    ptr = C.c_void_p(buf.ctypes.data) if buf.size else C.c_void_p(0)
    rc = L.rzk_wire_mat_decode(ptr, buf.size, N, coef_bytes, q, C.byref(rows), C.byref(cols), None, 0, C.byref(used))
    if rc != _lib.RZK_OK:
        raise ValueError("malformed Mat encoding")
    slab = np.empty((rows.value, cols.value, N), dtype=np.int64)
    rc = L.rzk_wire_mat_decode(ptr, buf.size, N, coef_bytes, q, C.byref(rows), C.byref(cols),
                               C.c_void_p(slab.ctypes.data), rows.value * cols.value, C.byref(used))
    if rc != _lib.RZK_OK:
        raise ValueError("malformed Mat encoding")
    return slab, used.value
