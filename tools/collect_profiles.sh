#!/bin/bash
# Round-end evidence (GPU box): kernel trace + stats of the default bench, PMC traffic passes, bench JSON.
# usage: tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*   (copy what should be judged into profiles/)
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o t -- python3 $root/bench.py --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_fetch -o f -- python3 $root/bench.py --steps 3 --warmup 1 --ramp 0 --no-cpu-baseline > /dev/null 2> $out/${tag}_pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_write -o w -- python3 $root/bench.py --steps 3 --warmup 1 --ramp 0 --no-cpu-baseline > /dev/null 2> $out/${tag}_pmc_write.log
cd $root
f=$(find $out/${tag}_pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find $out/${tag}_pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$f" "$w" $out/${tag}_pmc_row_kernel.json > /dev/null
grep -E "rzk::" $(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1) | head -8
python3 -c "import json; j=json.load(open('$out/${tag}_pmc_row_kernel.json')); print('traffic/alg', j['traffic_over_algorithmic'], 'bytes/launch', j['hbm_bytes_per_launch'])"
grep -E "row_kernel|unit_kernel" $(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1) > /dev/null && cp $(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_bench_open1024.csv
