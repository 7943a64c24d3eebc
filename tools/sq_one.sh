#!/bin/bash
# SQ counters of one entry point for a library variant: tools/sq_one.sh <tag> <lib|default> <entry> [B]
tag=$1; lib=$2; what=$3; B=${4:-8192}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
[ "$lib" != default ] && export RZK_LIB=$root/$lib
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_sq$i -o s -- python3 $root/tools/run_one.py $what $B > /dev/null 2> $out/${tag}_sq$i.log
done
cd $root
python3 tools/sq_counters.py $out/${tag}_sq1 $out/${tag}_sq2 > $out/${tag}_sq.json
python3 - <<PY
import json
j=json.load(open("$out/${tag}_sq.json"))
for k,v in j.items():
    if "row_kernel" in k or "unit_kernel" in k:
        w=v.get("SQ_WAVES",1)
        print("$tag", k[:44], "waves", int(w), " per wave: VALU %d SALU %d LDS %d VMEM_RD %d VMEM_WR %d SMEM %d | cycles/4: life %d active %d wait %d wait_inst %d | dur %.1f us" % tuple(
            [v.get(c,0)/w for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_INSTS_SMEM","SQ_WAVE_CYCLES","SQ_ACTIVE_INST_ANY","SQ_WAIT_ANY","SQ_WAIT_INST_ANY")] + [v.get("dur_us",0)]))
PY
