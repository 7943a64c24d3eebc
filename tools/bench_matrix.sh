#!/bin/bash
# Quick matrix of the BASELINE configurations (GPU box): kernels only, no CPU baseline, no secondary regions.
#   tools/bench_matrix.sh <tag> [chunks...]   ->  gpurun_out/<tag>_matrix.jsonl + a one-line summary per configuration
tag=${1:-m}; shift
chunks=${@:-512}
cd "$(dirname "$0")/.."
out=gpurun_out/${tag}_matrix.jsonl
: > $out
X="--no-cpu-baseline --extra-steps 0"
python bench.py --N 512 --steps 200 --warmup 50 $X >> $out 2>/dev/null
python bench.py --steps 200 --warmup 50 $X >> $out 2>/dev/null
python bench.py --N 2048 --steps 100 --warmup 20 $X >> $out 2>/dev/null
python bench.py --workload linear --batch 8192 --steps 50 --warmup 10 $X >> $out 2>/dev/null
python bench.py --workload sum --shape 4,9,4 --summands 8 --batch 4096 --steps 5 --warmup 2 $X >> $out 2>/dev/null
for c in $chunks; do
python bench.py --workload sum --N 2048 --shape 8,17,8 --summands 32 --batch 4096 --chunk $c --steps 1 --warmup 0 --ramp 0 $X >> $out 2>/dev/null
done
python - <<PY
import json
for line in open("$out"):
    line=line.strip()
    if not line.startswith("{"): continue
    j=json.loads(line)
    r=j["roofline"]
    ks=" | ".join("%s %.0fus x%.1f %.2f"%(k.replace("rzk::",""), v["avg_launch_us"], v["launches_per_step"], v["frac_of_hbm_peak"]) for k,v in list(r["kernels"].items())[:4])
    print("%-100s %12.0f /s  phases %s  cycle frac %.3f\n      %s" % (j["config"]["workload"][:100], j["value"], {k: round(v,1) for k,v in r["phase_us"].items()}, r["cycle"]["frac"], ks))
PY
