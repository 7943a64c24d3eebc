#!/bin/bash
# A/B the same library under different environment knobs, interleaved: ./tools/ab_env.sh "RZK_SHIFT=0" "RZK_SHIFT=1"
# (runs on the GPU box; each entry is a space-separated list of VAR=VALUE settings, "-" = none)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ab_env.jsonl
: > $out
for rep in 1 2 3; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    line=$(env $envs python bench.py --steps 200 --warmup 100 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1)
    echo "{\"cfg\": \"$cfg\", \"rep\": $rep, \"bench\": $line}" >> $out
  done
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/ab_env.jsonl")]
by={}
for r in rows: by.setdefault(r["cfg"],[]).append(r["bench"])
for k,v in by.items():
    vals=[b["value"] for b in v]
    ph=v[-1]["roofline"].get("phase_us")
    print(f"{k:30s} best {max(vals)/1e6:8.3f} M  median {sorted(vals)[len(vals)//2]/1e6:8.3f} M  phases {ph}")
PY
