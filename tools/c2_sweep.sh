#!/bin/bash
# Diagnostic: config 2 (chain) under variations of the planner's cost-model constants (JTP_COST_*), one line each.
OUT=${1:-gpurun_out/c2sweep.txt}
: > $OUT
for e in "X=0" "JTP_COST_RED_FIX=1" "JTP_COST_RED_FIX=2" "JTP_COST_RED_FIX=8" "JTP_COST_ITER_C=1.0 JTP_COST_ITER_D=1.2" "JTP_COST_ITER_C=0.2 JTP_COST_ITER_D=0.25" \
         "JTP_COST_ITER_C=1.0 JTP_COST_ITER_D=1.2 JTP_COST_RED_FIX=2" "JTP_COST_EPI=0.1" "JTP_COST_EPI=1.5" "JTP_COST_WG=0.5" "JTP_COST_WG=4" "JTP_COST_STAGE_FIX=2" "JTP_COST_STAGE_FIX=10" \
         "JTP_COST_LANE=0.02" "JTP_COST_LANE=0.4" "JTP_COST_WAVE=0.5" "JTP_COST_WAVE=4" "JTP_REDUCE_MIN=2" "JTP_REDUCE_MIN=64"; do
    env $e timeout -k 10 100 python3 bench.py --config c2 --cpu-sample 0 --steps 30 > /tmp/c2s.json 2>/dev/null || exit 1
    echo "$e : $(python3 tools/bsum.py /tmp/c2s.json | tr '\n' ' ' | tr -s ' ')" >> $OUT
done
