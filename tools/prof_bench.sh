#!/bin/bash
# rocprofv3 kernel trace of bench.py on the GPU box: ./tools/prof_bench.sh <tag> [bench args...]
# writes gpurun_out/prof_<tag>/..._kernel_stats.csv (copy the summary to profiles/ to keep it)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --steps 200 --warmup 100 --no-cpu-baseline "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(11), r['Percentage'])
PY
