#!/bin/bash
# Bench the BASELINE.json configurations that fit one GPU (GPU box). Output: one JSON line per config.
cd "$(dirname "$0")/.."
run() { echo "## $*"; python bench.py --no-cpu-baseline "$@" 2>/dev/null; }
if [ "$1" != "big" ]; then
run --workload open --N 512 --batch 4096
run --workload open --N 1024 --batch 4096
run --workload open --N 2048 --batch 4096 --steps 200 --warmup 50 --ramp 50
fi
run --workload linear --N 1024 --batch 8192 --steps 100 --warmup 20 --ramp 30
run --workload sum --N 1024 --shape 4,9,4 --summands 8 --batch 4096 --steps 10 --warmup 2 --ramp 4
run --workload sum --N 2048 --shape 8,17,8 --summands 32 --batch 256 --steps 5 --warmup 1 --ramp 2
