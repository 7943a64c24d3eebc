#!/bin/bash
# Config 5 (Sum, N = 2048, (8,17,8), V = 32, 4096 proofs per GPU) at several chunk sizes (GPU box):
#   tools/chunk_sweep.sh <tag> [chunks...]  ->  gpurun_out/<tag>_chunks.jsonl + one summary line per chunk size
tag=${1:-c5}; shift
chunks=${@:-256 512 1024}
cd "$(dirname "$0")/.."
out=gpurun_out/${tag}_chunks.jsonl
: > $out
for c in $chunks; do
  python bench.py --workload sum --N 2048 --shape 8,17,8 --summands 32 --batch 4096 --chunk $c --steps 1 --warmup 0 --ramp 0 \
    --no-cpu-baseline --extra-steps 0 >> $out 2>/dev/null || echo "chunk $c failed"
done
python - <<PY
import json
for line in open("$out"):
    if not line.startswith("{"): continue
    j=json.loads(line); r=j["roofline"]
    print(j["config"]["workload"][-40:], round(j["value"]), round(j["ms_per_step"],1), {k: round(v) for k,v in r["phase_us"].items()})
PY
