"""ctypes binding of include/rzk.h.  Fails loudly when the HIP library is missing: there is no
CPU fallback in this package."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("RZK_LIB", os.path.join(HERE, "librzk_hip.so"))  # RZK_LIB: tuning variants (tools/)

RZK_OK, RZK_E_ARG, RZK_E_HIP, RZK_E_STATE, RZK_E_UNSUPPORTED = 0, -1, -2, -3, -4
KEY_A1, KEY_A2, KEY_A = 0, 1, 2

_lib = None

_I64 = C.c_void_p  # int64_t* (host or device pointer, passed as an address)
_U8 = C.c_void_p
_U32P = C.c_void_p
_SZ = C.c_size_t
_CTX = C.c_void_p

# name -> (restype, argtypes); every symbol include/rzk.h declares
SIGNATURES = {
    "rzk_ctx_create": (C.c_int, [C.POINTER(_CTX), C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_uint64, C.c_int]),
    "rzk_ctx_destroy": (None, [_CTX]),
    "rzk_abi_version": (C.c_uint32, []),
    "rzk_ctx_trust_device_outputs": (C.c_int, [_CTX, C.c_int]),
    "rzk_ctx_set_stream": (C.c_int, [_CTX, C.c_void_p]),
    "rzk_ctx_use_own_stream": (C.c_int, [_CTX]),
    "rzk_ctx_synchronize": (C.c_int, [_CTX]),
    "rzk_ctx_check_inputs": (C.c_int, [_CTX]),
    "rzk_last_error": (C.c_char_p, [_CTX]),
    "rzk_sigma": (C.c_uint64, [_CTX]),
    "rzk_commit_bound": (C.c_uint64, [_CTX]),
    "rzk_verify_bound": (C.c_uint64, [_CTX]),
    "rzk_key_load": (C.c_int, [_CTX, _I64]),
    "rzk_key_load_dev": (C.c_int, [_CTX, _I64]),
    "rzk_key_generate": (C.c_int, [_CTX, C.c_uint64, _I64]),
    "rzk_polymul_batch": (C.c_int, [_CTX, _I64, _I64, _I64, _SZ]),
    "rzk_matvec_batch": (C.c_int, [_CTX, C.c_int, _I64, _I64, _I64, _SZ]),
    "rzk_cmul_batch": (C.c_int, [_CTX, _I64, C.c_uint32, _I64, _I64, _SZ]),
    "rzk_canonicalize_batch": (C.c_int, [_CTX, _I64, _I64, _SZ]),
    "rzk_add_batch": (C.c_int, [_CTX, _I64, _I64, _I64, _SZ]),
    "rzk_sub_batch": (C.c_int, [_CTX, _I64, _I64, _I64, _SZ]),
    "rzk_norm2_le_batch": (C.c_int, [_CTX, _I64, C.c_uint32, C.c_uint64, _U8, _SZ]),
    "rzk_eq_batch": (C.c_int, [_CTX, _I64, _I64, C.c_uint32, _U8, _SZ]),
    "rzk_ntt_forward_batch": (C.c_int, [_CTX, C.c_int, _U32P, _U32P, _SZ]),
    "rzk_ntt_inverse_batch": (C.c_int, [_CTX, C.c_int, _U32P, _U32P, _SZ]),
    "rzk_ntt_prime": (C.c_uint32, [C.c_int]),
    "rzk_ntt_psi": (C.c_uint32, [C.c_int, C.c_uint32]),
    "rzk_ntt_layout_index": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "rzk_commit_batch": (C.c_int, [_CTX, _I64, _I64, _I64, _U8, _SZ]),
    "rzk_commitment_verify_batch": (C.c_int, [_CTX, _I64, _I64, _I64, _I64, _U8, _SZ]),
    "rzk_open_commit_batch": (C.c_int, [_CTX, _I64, _I64, _I64, _I64, _I64, _U8, _SZ]),
    "rzk_open_response_batch": (C.c_int, [_CTX, _I64, _I64, _I64, _I64, _SZ]),
    "rzk_open_verify_batch": (C.c_int, [_CTX, _I64, _I64, _I64, _I64, _U8, _SZ]),
    "rzk_linear_commit_batch": (C.c_int, [_CTX] + [_I64] * 11 + [_U8, _SZ]),
    "rzk_linear_response_batch": (C.c_int, [_CTX] + [_I64] * 7 + [_SZ]),
    "rzk_linear_verify_batch": (C.c_int, [_CTX] + [_I64] * 9 + [_U8, _SZ]),
    "rzk_sum_commit_batch": (C.c_int, [_CTX, C.c_uint32] + [_I64] * 11 + [_U8, _SZ]),
    "rzk_sum_response_batch": (C.c_int, [_CTX, C.c_uint32] + [_I64] * 7 + [_SZ]),
    "rzk_sum_verify_batch": (C.c_int, [_CTX, C.c_uint32] + [_I64] * 9 + [_U8, _SZ]),
    "rzk_sample_uniform_dev": (C.c_int, [_CTX, C.c_uint64, C.c_uint32, C.c_uint64, _I64, _SZ]),
    "rzk_sample_gauss_dev": (C.c_int, [_CTX, C.c_uint64, C.c_uint32, C.c_double, _I64, _SZ]),
    "rzk_sample_challenge_dev": (C.c_int, [_CTX, C.c_uint64, C.c_uint32, _I64, _SZ]),
    "rzk_wire_mat_size": (C.c_size_t, [_I64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rzk_wire_mat_encode": (C.c_int, [_I64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _U8, _SZ,
                                      C.POINTER(C.c_size_t)]),
    "rzk_wire_mat_decode": (C.c_int, [_U8, _SZ, C.c_uint32, C.c_uint32, C.c_int64, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32), _I64, _SZ, C.POINTER(C.c_size_t)]),
    "rzk_bench_ntt_forward_dev": (C.c_double, [_CTX, C.c_int, _U32P, _U32P, _SZ, C.c_int]),
    "rzk_debug_read_scratch": (C.c_int, [_CTX, C.c_void_p, _SZ, C.POINTER(C.c_size_t)]),
    "rzk_prof_reset": (C.c_int, [_CTX]),
    "rzk_prof_enable": (C.c_int, [_CTX, C.c_int]),
    "rzk_prof_read": (C.c_int, [_CTX, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "rzk_prof_count": (C.c_uint64, [_CTX]),
    "rzk_prof_read_all": (C.c_int, [_CTX, C.POINTER(C.c_double), _SZ, C.POINTER(C.c_size_t)]),
    "rzk_prof_read_kernels": (C.c_int, [_CTX, C.c_char_p, _SZ, C.POINTER(C.c_size_t)]),
}
ABI_VERSION = 3   # include/rzk.h: RZK_ABI_VERSION
# every batched entry point also exists as a device-pointer variant with the same signature
for _name in list(SIGNATURES):
    if _name.endswith("_batch"):
        SIGNATURES[_name + "_dev"] = SIGNATURES[_name]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RuntimeError(
                f"{SO} is missing: build it with `python -m ring_zk_amd.build` (hipcc, gfx950). "
                "The ring-zk MI355X backend has no CPU fallback.")
        # PyTorch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Device memory and streams
        # are shared with torch, so torch's HIP runtime must be the one this process uses: load it first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO)
        # a stale or variant library with another ABI must fail here, not read shifted arguments later
        try:
            L.rzk_abi_version.restype = C.c_uint32
            have = int(L.rzk_abi_version())
        except AttributeError:
            have = None
        if have != ABI_VERSION:
            raise RuntimeError(f"{SO}: C ABI version {have}, this binding needs {ABI_VERSION} (include/rzk.h); rebuild with "
                               "`python -m ring_zk_amd.build`")
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
