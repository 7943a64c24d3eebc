"""ring_zk_amd — MI355X (gfx950) polynomial-ring backend for the ring-zk proof system.

The package holds the HIP kernels + C ABI (csrc/, include/rzk.h -> librzk_hip.so), a thin Python host
layer over that ABI (backend.Context) and the batched mirror of the reference's prover / verifier API
(protocols).  It never computes on the CPU: a missing HIP library or device raises.
"""
from .backend import Context, Q_DEFAULT, RzkError  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["Context", "Q_DEFAULT", "RzkError"]
