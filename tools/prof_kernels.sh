#!/bin/bash
# rocprofv3 kernel stats of one bench.py command (GPU box):  tools/prof_kernels.sh <tag> <bench args...>
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o p -- python3 $root/bench.py --no-cpu-baseline "$@" > $root/gpurun_out/prof_$tag.log 2>&1
f=$(find $root/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(11), ("%.1f ms" % (float(r['TotalDurationNs'])/1e6)).rjust(10), r['Percentage'])
PY
