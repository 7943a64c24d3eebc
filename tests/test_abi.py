"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/rzk.h declares, and
refuses to create a context without a HIP device (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from ring_zk_amd import build

    return build.build_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rzk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rzk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(built):
    lib = C.CDLL(built)
    names = declared_symbols()
    assert len(names) >= 50
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/rzk.h but not exported"


def test_binding_table_matches_header(built):
    from ring_zk_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.lib()


def test_pure_host_queries(built):
    from ring_zk_amd import _lib

    L = _lib.lib()
    primes = [L.rzk_ntt_prime(i) for i in range(3)]
    assert primes == [1073692673, 1073668097, 1073651713]
    assert L.rzk_ntt_prime(3) == 0
    for N in (512, 1024, 2048):
        idx = sorted(L.rzk_ntt_layout_index(N, j) for j in range(N))
        assert idx == list(range(N))
        for i, p in enumerate(primes):
            psi = L.rzk_ntt_psi(i, N)
            assert pow(psi, N, p) == p - 1 and pow(psi, 2 * N, p) == 1


def test_no_cpu_fallback_without_device(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; the no-device behaviour cannot be exercised")
    from ring_zk_amd import Context, RzkError

    with pytest.raises(RzkError):
        Context(1024)


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under ring_zk_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ring_zk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "rzk_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_abi_version_matches_header():
    """A stale or variant .so must fail at load time: the binding checks rzk_abi_version()."""
    import re

    from ring_zk_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "rzk.h")).read()
    m = re.search(r"#define RZK_ABI_VERSION (\d+)u", hdr)
    assert m and int(m.group(1)) == _lib.ABI_VERSION
    assert int(_lib.lib().rzk_abi_version()) == _lib.ABI_VERSION
