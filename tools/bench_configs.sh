#!/bin/bash
# The BASELINE configurations on one GPU (GPU box): tools/bench_configs.sh <tag>  ->  gpurun_out/<tag>_bench_configs.jsonl
# config 2: Open N=512 B=4096; metric config: Open N=1024 B=4096; config 3: Sum N=1024 (4,9,4) V=8 B=4096;
# config 4: Linear N=1024 (1,3,1), 65536 / 8 = 8192 proofs per GPU; config 5: Sum N=2048 (8,17,8) V=32, 32768 / 8 = 4096
# proofs per GPU in chunks of 256 (the whole batch does not fit in HBM).
tag=${1:-cfg}
cd "$(dirname "$0")/.."
out=gpurun_out/${tag}_bench_configs.jsonl
: > $out
python bench.py --N 512 --steps 200 --warmup 50 --cpu-seconds 8 >> $out 2>/dev/null
python bench.py --steps 200 --warmup 50 --cpu-seconds 8 >> $out 2>/dev/null
python bench.py --N 2048 --batch 4096 --steps 100 --warmup 20 --no-cpu-baseline >> $out 2>/dev/null
python bench.py --workload linear --batch 8192 --steps 50 --warmup 10 --cpu-seconds 8 >> $out 2>/dev/null
python bench.py --workload sum --shape 4,9,4 --summands 8 --batch 4096 --steps 5 --warmup 2 --cpu-seconds 8 >> $out 2>/dev/null
python bench.py --workload sum --N 2048 --shape 8,17,8 --summands 32 --batch 4096 --chunk 256 --steps 1 --warmup 0 --ramp 0 --cpu-seconds 20 >> $out 2>/dev/null
python - <<PY
import json
for line in open("$out"):
    line=line.strip()
    if not line.startswith("{"): continue
    j=json.loads(line)
    r=j["roofline"]; u=j["units"]; c=j.get("cpu_baseline") or {}
    print("%-90s %12.0f proofs/s  phases %s  dom frac %.3f cycle frac %.3f  bfly frac %.3f  cpu %s" % (
        j["config"]["workload"], j["value"], {k: round(v,1) for k,v in r["phase_us"].items()}, r["frac"], r["cycle"]["frac"],
        u["frac_of_butterfly_peak"], round(c.get("value",0),1)))
PY
