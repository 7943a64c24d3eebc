#!/bin/bash
# SQ counters of the bench kernels (GPU box): three separate --pmc passes (counter slots), csv output, then
# tools/sq_counters.py aggregates per kernel.  usage: tools/sq_counters.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_COUNT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_sq$i -o s -- python3 $root/bench.py --steps 3 --warmup 1 --ramp 0 --no-cpu-baseline > /dev/null 2> $out/${tag}_sq$i.log
done
cd $root
python3 tools/sq_counters.py $out/${tag}_sq1 $out/${tag}_sq2 $out/${tag}_sq3 > $out/${tag}_sq_counters.json
python3 - <<PY
import json
j=json.load(open("$out/${tag}_sq_counters.json"))
for k,v in j.items():
    if "unit_kernel" in k or "row_" in k:
        busy=v.get("SQ_ACTIVE_INST_VALU",0)*4/max(v.get("SQ_BUSY_CYCLES",1),1)
        print(k, "VALU insts/wave", round(v.get("SQ_INSTS_VALU",0)/max(v.get("SQ_WAVES",1),1)), "dur_us", v.get("dur_us"))
PY
