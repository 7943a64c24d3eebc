#!/bin/bash
# VALU instructions per wavefront of every rzk kernel of one bench configuration (GPU box):
#   tools/valu_per_kernel.sh <tag> [bench args...]   -> gpurun_out/<tag>_valu.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $root/gpurun_out/valu_$tag -o v -- python3 $root/bench.py --steps 1 --warmup 0 --ramp 0 --no-cpu-baseline "$@" > /dev/null 2> $root/gpurun_out/valu_$tag.log
cd $root
python3 - "$(find gpurun_out/valu_$tag -name '*counter_collection.csv' | head -1)" > gpurun_out/${tag}_valu.txt <<'PY'
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "rzk::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "") + " grid=" + r["Grid_Size"]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        cnt[name] += 1
for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    w = max(c.get("SQ_WAVES", 1), 1)
    print("%-60s launches %3d waves/launch %8.0f VALU/wave %8.0f SALU/wave %7.0f LDS/wave %6.0f VALU-active share of wave life %.2f" % (
        name[:60], cnt[name], w / max(cnt[name], 1), c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w,
        c.get("SQ_INSTS_LDS", 0) / w, c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
PY
cat gpurun_out/${tag}_valu.txt
