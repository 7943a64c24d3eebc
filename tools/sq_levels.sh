#!/bin/bash
# average latencies (LEVEL / INSTS) of VMEM, SMEM, LDS and instruction fetch for one entry point: tools/sq_levels.sh <tag> <lib|default> <entry> [B]
tag=$1; lib=$2; what=$3; B=${4:-8192}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
[ "$lib" != default ] && export RZK_LIB=$root/$lib
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_lv$i -o s -- python3 $root/tools/run_one.py $what $B > /dev/null 2> $out/${tag}_lv$i.log
done
cd $root
python3 tools/sq_counters.py $out/${tag}_lv1 $out/${tag}_lv2 $out/${tag}_lv3 > $out/${tag}_lv.json
python3 - <<PY
import json
j=json.load(open("$out/${tag}_lv.json"))
for k,v in j.items():
    if "row_kernel" in k or "unit_kernel" in k:
        w=v.get("SQ_WAVES",1)
        g=lambda c: v.get(c,0)
        print("$tag", k[:40], "| VMEM n/wave %d lat %.0f | SMEM n %d lat %.0f | LDS n %d lat %.0f | IFETCH n %d lat %.0f | SALU cyc %d SMEM cyc %d branches %d waitLDS %d life %d" % (
          g("SQ_INSTS_VMEM")/w, g("SQ_INST_LEVEL_VMEM")/max(g("SQ_INSTS_VMEM"),1), g("SQ_INSTS_SMEM")/w, g("SQ_INST_LEVEL_SMEM")/max(g("SQ_INSTS_SMEM"),1),
          g("SQ_INSTS_LDS")/w, g("SQ_INST_LEVEL_LDS")/max(g("SQ_INSTS_LDS"),1), g("SQ_IFETCH")/w, g("SQ_IFETCH_LEVEL")/max(g("SQ_IFETCH"),1),
          g("SQ_INST_CYCLES_SALU")/w, g("SQ_INST_CYCLES_SMEM")/w, g("SQ_INSTS_BRANCH")/w, g("SQ_WAIT_INST_LDS")/w, g("SQ_WAVE_CYCLES")/w))
PY
