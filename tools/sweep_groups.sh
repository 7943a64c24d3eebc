#!/bin/bash
cd "$(dirname "$0")/.."
for w in 1 3; do
python - <<PY
from ring_zk_amd import build
build.build_library(force=True, out="/tmp/librzk_gw$w.so", defines=["RZK_GROUP_MIN_WAVES=$w"])
PY
echo "== RZK_GROUP_MIN_WAVES=$w"
RZK_LIB=/tmp/librzk_gw$w.so python tools/bench_commit_shape.py 1024 4 9 4 16384
RZK_LIB=/tmp/librzk_gw$w.so python tools/bench_commit_shape.py 2048 8 17 8 4096
RZK_LIB=/tmp/librzk_gw$w.so python tools/bench_commit_shape.py 512 2 5 2 16384
done
