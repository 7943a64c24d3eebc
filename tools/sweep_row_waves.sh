#!/bin/bash
# Build the library with different register budgets for the row kernel and bench each (GPU box).
set -e
cd "$(dirname "$0")/.."
for w in 1 3 4 5; do
  python - <<PY
from ring_zk_amd import build
build.build_library(force=True, out="/tmp/librzk_w$w.so", defines=["RZK_ROW_MIN_WAVES=$w"])
PY
  echo "== RZK_ROW_MIN_WAVES=$w"
  RZK_LIB=/tmp/librzk_w$w.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('value %.0f proofs/s  ms/step %.3f  row avg %.1f us  frac %.3f  ntt %.0f GB/s'%(j['value'],j['ms_per_step'],j['roofline']['avg_launch_us'],j['roofline']['frac'],j['ntt_roofline']['achieved']))"
done
