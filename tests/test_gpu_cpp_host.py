"""Builds and runs tests/cpp/test_ring_zk.cpp: the C++ mirror of the reference's integration tests
(tests/test.rs) over ring_zk_amd/host/ring_zk.hpp -> the C ABI -> the HIP kernels."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ring_n", [16, 512])   # 16 is the reference's own test size (tests/test.rs:8)
def test_cpp_host_mirror_runs_reference_style_tests(tmp_path, ring_n):
    from ring_zk_amd import build

    so = build.build_library()
    libdir = os.path.dirname(so)
    exe = str(tmp_path / f"test_ring_zk_{ring_n}")
    subprocess.check_call(["g++", "-O2", "-std=c++17", f"-DTEST_N={ring_n}",
                           os.path.join(ROOT, "tests", "cpp", "test_ring_zk.cpp"),
                           "-L" + libdir, "-lrzk_hip", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe, "100"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all ok" in out.stdout
