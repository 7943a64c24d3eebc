#!/bin/bash
# A/B prebuilt library variants: tools/ab_libs.sh name=path ...
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name="${spec%%=*}"; lib="${spec#*=}"
  for i in 1 2; do
  RZK_LIB=$lib python bench.py --steps 200 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$name: %.0f proofs/s  ms/step %.3f  row avg %.1f us  phases %s'%(j['value'],j['ms_per_step'],j['roofline']['avg_launch_us'],{k:round(v,1) for k,v in j['roofline']['phase_us'].items()}))"
  done
done
