#!/bin/bash
# rocprofv3 kernel stats of the same configuration on this tree and on the round-2 tree (.ab_r02/): per-kernel A/B.
#   tools/prof_ab.sh <tag> <bench args...>
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for tree in new r02; do
  dir=$root; [ $tree = r02 ] && dir=$root/.ab_r02
  extra="--extra-steps 0"; [ $tree = r02 ] && extra=""
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_$tree -o p -- python3 $dir/bench.py --no-cpu-baseline $extra "$@" > $root/gpurun_out/prof_${tag}_$tree.log 2>&1
  f=$(find $root/gpurun_out/prof_${tag}_$tree -name "*kernel_stats.csv" | head -1)
  echo "== $tree"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(11), ("%.1f ms" % (float(r['TotalDurationNs'])/1e6)).rjust(10), r['Percentage'])
PY
done
