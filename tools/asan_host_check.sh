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
for n in 16 512 1024 2048; do
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -fno-omit-frame-pointer -DTEST_N=$n tests/cpp/test_ring_zk.cpp \
    -L$out -lrzk_hip_asan -Wl,-rpath,$out -o $out/test_ring_zk_$n
done
run() { echo "== N=$1 shape=${2:-default}"; TEST_SHAPE=$2 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 timeout -k 10 300 $out/test_ring_zk_$1 $3 2>&1 | tail -3; }
run 16 "" 20
run 512 "" 20
run 512 2,5,2 5      # row groups, two-step relation
run 1024 4,9,4 3     # row groups (4 rows), shared-operand sums
run 2048 2,5,2 2     # row blocks
