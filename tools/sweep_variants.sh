#!/bin/bash
# Build the library with sets of -D defines and bench each on the GPU box.
# usage: tools/sweep_variants.sh "NAME1:DEF_A=1,DEF_B=2" "NAME2:" ...
set -e
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  python - <<PY
from ring_zk_amd import build
defs=[d for d in "$defs".split(",") if d]
build.build_library(force=True, out="/tmp/librzk_$name.so", defines=defs)
PY
  for i in 1 2; do
  RZK_LIB=/tmp/librzk_$name.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$name [$defs]: %.0f proofs/s  ms/step %.3f  row avg %.1f us  frac %.3f  ntt %.0f GB/s  phases %s'%(j['value'],j['ms_per_step'],j['roofline']['avg_launch_us'],j['roofline']['frac'],j['ntt_roofline']['achieved'],{k:round(v,1) for k,v in j['roofline']['phase_us'].items()}))"
  done
done
