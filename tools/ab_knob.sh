#!/bin/bash
# A/B of one environment knob on the GPU box, alternating runs:  tools/ab_knob.sh "<bench args>" VAR=a VAR=b [repeats]
args=$1; A=$2; B=$3; rep=${4:-2}
cd "$(dirname "$0")/.."
for i in $(seq $rep); do
  for kv in "$A" "$B"; do
    env $kv python bench.py $args --no-cpu-baseline --extra-steps 0 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-22s %12.0f /s  %s' % ('$kv', j['value'], {k: round(v,1) for k,v in j['roofline']['phase_us'].items()}))"
  done
done
