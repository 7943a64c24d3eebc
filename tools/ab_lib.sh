#!/bin/bash
# A/B of library variants (tools/ab_build.py) on the GPU box, alternating runs:  tools/ab_lib.sh "<bench args>" <repeats> default rr1 rr2 ...
args=$1; rep=$2; shift 2
# (DIAG=RZK_BENCH_DIAG=1 in the environment for diagnostic builds whose results are wrong on purpose)
cd "$(dirname "$0")/.."
for i in $(seq $rep); do
  for v in "$@"; do
    lib=""; [ "$v" != default ] && lib="RZK_LIB=$PWD/ring_zk_amd/variants/lib_$v.so"
    env $DIAG $lib python bench.py $args --no-cpu-baseline --extra-steps 0 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-12s %12.0f /s  %s' % ('$v', j['value'], {k: round(v,1) for k,v in j['roofline']['phase_us'].items()}))"
  done
done
