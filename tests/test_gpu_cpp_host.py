"""Builds and runs tests/cpp/test_ring_zk.cpp: the C++ mirror of the reference's integration tests
(tests/test.rs) over ring_zk_amd/host/ring_zk.hpp -> the C ABI -> the HIP kernels."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# 16 is the reference's own test size (tests/test.rs:8); the larger shapes drive the row-group, shared-operand and
# row-block kernels through the same flows
@pytest.mark.parametrize("ring_n,shape,iters", [(4, "mat-only", 1), (16, "", 100), (512, "", 100), (512, "2,5,2", 10),
                                                (2048, "2,5,2", 2)])
def test_cpp_host_mirror_runs_reference_style_tests(tmp_path, ring_n, shape, iters):
    from ring_zk_amd import build

    so = build.build_library()
    libdir = os.path.dirname(so)
    exe = str(tmp_path / f"test_ring_zk_{ring_n}")
    subprocess.check_call(["g++", "-O2", "-std=c++17", f"-DTEST_N={ring_n}",
                           os.path.join(ROOT, "tests", "cpp", "test_ring_zk.cpp"),
                           "-L" + libdir, "-lrzk_hip", "-Wl,-rpath," + libdir, "-o", exe])
    env = dict(os.environ)
    if shape == "mat-only":      # the reference's Mat unit tests run at ring degree 4 (src/mat.rs:241)
        env["TEST_ONLY"] = "mat"
    elif shape:
        env["TEST_SHAPE"] = shape
    out = subprocess.run([exe, str(iters)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all ok" in out.stdout
