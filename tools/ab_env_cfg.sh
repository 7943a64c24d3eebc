#!/bin/bash
# A/B environment knobs on one bench configuration: BENCH_ARGS="--workload sum ..." tools/ab_env_cfg.sh "VAR=1" "VAR=2" -
cd "$(dirname "$0")/.."
for rep in 1 2; do
for kv in "$@"; do
  if [ "$kv" = "-" ]; then envs=""; else envs="$kv"; fi
  env $envs python bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$kv: %.0f proofs/s  ms/step %.3f  phases %s'%(j['value'],j['ms_per_step'],{k:round(v,1) for k,v in j['roofline']['phase_us'].items()}))"
done
done
