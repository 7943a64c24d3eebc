#!/bin/bash
# Host-side AddressSanitizer run of the C ABI (GPU box): the library's host code and the C++ mirror test are built
# with -fsanitize=address (device code untouched), then tests/cpp/test_ring_zk.cpp runs at N = 16 and N = 512.
set -e
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/asan
mkdir -p $out
cd $root
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -std=c++17 -fPIC -shared -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
  -o $out/librzk_hip_asan.so ring_zk_amd/csrc/rzk_kernels.hip ring_zk_amd/csrc/rzk_api.cpp ring_zk_amd/csrc/rzk_wire.cpp
for n in 16 512; do
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -fno-omit-frame-pointer -DTEST_N=$n tests/cpp/test_ring_zk.cpp \
    -L$out -lrzk_hip_asan -Wl,-rpath,$out -o $out/test_ring_zk_$n
  ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 timeout -k 10 300 $out/test_ring_zk_$n 20 2>&1 | tail -8
done
