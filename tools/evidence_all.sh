#!/bin/bash
# Evidence of every BASELINE configuration (GPU box), one tools/collect_evidence.sh run each:  tools/evidence_all.sh <tag> [which...]
tag=${1:-r03}; shift
which=${@:-open1024 open512 open2048 linear sum494 sum8178}
cd "$(dirname "$0")/.."
for w in $which; do
  case $w in
    open1024) tools/collect_evidence.sh $tag open1024 open-1024-1,3,1 ;;
    open512)  tools/collect_evidence.sh $tag open512 open-512-1,3,1 --N 512 --steps 300 --warmup 50 ;;
    open2048) tools/collect_evidence.sh $tag open2048 open-2048-1,3,1 --N 2048 --steps 100 --warmup 20 ;;
    linear)   tools/collect_evidence.sh $tag linear linear-1024-1,3,1 --workload linear --batch 8192 --steps 50 --warmup 10 ;;
    sum494)   tools/collect_evidence.sh $tag sum494 sum-1024-4,9,4 --workload sum --shape 4,9,4 --summands 8 --batch 4096 --steps 5 --warmup 2 ;;
    sum8178)  tools/collect_evidence.sh $tag sum8178 sum-2048-8,17,8 --workload sum --N 2048 --shape 8,17,8 --summands 32 --batch 4096 --chunk 512 --steps 1 --warmup 0 --ramp 0 --extra-steps 0 ;;
  esac
done
