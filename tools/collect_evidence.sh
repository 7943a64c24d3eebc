#!/bin/bash
# Evidence of one BASELINE configuration (GPU box): bench JSON, rocprofv3 kernel stats of the same command, FETCH / WRITE
# passes -> HBM bytes per kernel, SQ counter passes.  Everything lands in gpurun_out/<tag>_<name>_*; copy what should
# be judged into profiles/.
#   tools/collect_evidence.sh <tag> <name> <workload-key> [bench args...]
tag=$1; name=$2; key=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
pre=$out/${tag}_${name}
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --no-cpu-baseline "$@" > ${pre}_bench.json 2> ${pre}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d ${pre}_trace -o t -- python3 $root/bench.py --no-cpu-baseline --extra-steps 0 "$@" > ${pre}_bench_under_rocprof.json 2> ${pre}_trace.log
small="--steps 3 --warmup 1 --ramp 0 --no-cpu-baseline --extra-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${pre}_pmc_fetch -o f -- python3 $root/bench.py "$@" $small > /dev/null 2> ${pre}_pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${pre}_pmc_write -o w -- python3 $root/bench.py "$@" $small > /dev/null 2> ${pre}_pmc_write.log
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_COUNT GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${pre}_sq$i -o s -- python3 $root/bench.py "$@" $small > /dev/null 2> ${pre}_sq$i.log
done
cd $root
cp "$(find ${pre}_trace -name '*kernel_stats.csv' | head -1)" ${pre}_kernel_stats.csv
python3 tools/pmc_kernels.py ${pre}_pmc_fetch ${pre}_pmc_write ${pre}_bench.json "$key" ${pre}_pmc_traffic.json > /dev/null
python3 tools/sq_counters.py ${pre}_sq1 ${pre}_sq2 ${pre}_sq3 > ${pre}_sq_counters.json
python3 - <<PY
import json,csv
j=[json.loads(l) for l in open("${pre}_bench.json") if l.startswith("{")][-1]
print("${name}:", round(j["value"]), {k: round(v) for k, v in j.items() if k.startswith("value_")}, {k: round(v,1) for k,v in j["roofline"]["phase_us"].items()})
t=json.load(open("${pre}_pmc_traffic.json"))
for k,v in list(t["kernels"].items())[:6]:
    print("   %-52s traffic/alg %s  HBM-side %.2f TB/s" % (k, round(v.get("traffic_over_algorithmic", 0), 2), v.get("hbm_side_TBps", 0)))
rows=list(csv.DictReader(open("${pre}_kernel_stats.csv")))
for r in rows[:6]:
    print("   rocprof %-70s %6s x %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
s=json.load(open("${pre}_sq_counters.json"))
for k,v in s.items():
    if v.get("SQ_WAVES") and ("unit_" in k or "row_" in k):
        w=v["SQ_WAVES"]; wc=v.get("SQ_WAVE_CYCLES",0)
        print("   SQ %-60s VALU/wave %6.0f  VALU share of wave life %.2f  wait_any %.2f" % (k[:60], v.get("SQ_INSTS_VALU",0)/w, v.get("SQ_ACTIVE_INST_VALU",0)/max(wc,1), v.get("SQ_WAIT_ANY",0)/max(wc,1)))
PY
