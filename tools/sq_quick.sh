#!/bin/bash
# SQ counters of one bench configuration, summary per kernel: tools/sq_quick.sh <tag> [bench args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
small="--steps 3 --warmup 1 --ramp 0 --no-cpu-baseline --extra-steps 0"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_sq$i -o s -- python3 $root/bench.py "$@" $small > /dev/null 2> $out/${tag}_sq$i.log
done
cd $root
python3 tools/sq_counters.py $out/${tag}_sq1 $out/${tag}_sq2 > $out/${tag}_sq_counters.json
python3 - <<PY
import json
s=json.load(open("$out/${tag}_sq_counters.json"))
for k,v in s.items():
    if v.get("SQ_WAVES") and ("unit_" in k or "row_" in k or "shift" in k):
        w=v["SQ_WAVES"]; wc=v.get("SQ_WAVE_CYCLES",0)
        print("%-58s VALU/w %6.0f LDS/w %5.0f VMEMrd/w %4.0f SALU/w %5.0f | share of wave life: VALU %.2f LDS %.3f wait_any %.2f wait_inst %.2f | bank confl/LDS active %.2f dur %.1f" % (
          k[:58], v.get("SQ_INSTS_VALU",0)/w, v.get("SQ_INSTS_LDS",0)/w, v.get("SQ_INSTS_VMEM_RD",0)/w, v.get("SQ_INSTS_SALU",0)/w,
          v.get("SQ_ACTIVE_INST_VALU",0)/max(wc,1), v.get("SQ_ACTIVE_INST_LDS",0)/max(wc,1), v.get("SQ_WAIT_ANY",0)/max(wc,1), v.get("SQ_WAIT_INST_ANY",0)/max(wc,1),
          v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_LDS_IDX_ACTIVE",1),1), v.get("dur_us",0)))
PY
