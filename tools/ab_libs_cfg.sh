#!/bin/bash
# A/B prebuilt libraries on one bench configuration: BENCH_ARGS="--workload sum ..." tools/ab_libs_cfg.sh name=path ...
cd "$(dirname "$0")/.."
for rep in 1 2; do
for spec in "$@"; do
  name="${spec%%=*}"; lib="${spec#*=}"
  RZK_LIB=$lib python bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$name: %.0f proofs/s  ms/step %.3f  phases %s'%(j['value'],j['ms_per_step'],{k:round(v,1) for k,v in j['roofline']['phase_us'].items()}))"
done
done
