#!/bin/bash
# A/B against the round-2 tree: same box, interleaved runs.  The tree is not tracked; recreate it with
#   git worktree add .ab_r02 7622940 && (cd .ab_r02 && python -m ring_zk_amd.build)     (7622940 = end of round 2)
cd "$(dirname "$0")/.."
sumline='import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    j=json.loads(line); r=j["roofline"]
    print("%-8s %12.0f /s  %s" % (sys.argv[1], j["value"], {k: round(v,1) for k,v in r["phase_us"].items()}))'
for cfg in "--workload linear --batch 8192 --steps 50 --warmup 10" "--workload sum --shape 4,9,4 --summands 8 --batch 4096 --steps 5 --warmup 2" "--N 2048 --steps 100 --warmup 20"; do
  echo "== $cfg"
  for rep in 1 2; do
    (cd .ab_r02 && python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c "$sumline" r02)
    python bench.py $cfg --no-cpu-baseline --extra-steps 0 2>/dev/null | python -c "$sumline" new
  done
done
