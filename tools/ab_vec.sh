#!/bin/bash
# A/B of an environment knob on the Linear / Sum configurations: tools/ab_vec.sh "VAR=0" "VAR=1"
cd "$(dirname "$0")/.."
show='import json,sys
j=json.loads(sys.stdin.read())
print("   %-60s %10.0f proofs/s  phases %s" % (j["config"]["workload"][:60], j["value"], {k: round(v,1) for k,v in j["roofline"]["phase_us"].items()}))'
for kv in "$@"; do
  echo "== $kv"
  env $kv python bench.py --workload linear --batch 8192 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$show"
  env $kv python bench.py --workload sum --shape 4,9,4 --summands 8 --batch 4096 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$show"
  env $kv python bench.py --workload sum --N 2048 --shape 8,17,8 --summands 32 --batch 256 --steps 2 --warmup 1 --ramp 0 --no-cpu-baseline 2>/dev/null | python -c "$show"
done
