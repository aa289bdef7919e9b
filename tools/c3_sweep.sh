#!/bin/bash
# Diagnostic: config 3 under variations of the planner's cost-model constants (JTP_COST_*), one line each.
OUT=${1:-gpurun_out/c3sweep.txt}
: > $OUT
for e in "JTP_DUMMY=0" "JTP_COST_MAX_CU=3" "JTP_COST_EPI=0.25" "JTP_COST_EPI=1.0" "JTP_COST_STAGE_FIX=2.5" "JTP_COST_STAGE_FIX=8" "JTP_COST_BW=3.5e6" "JTP_COST_STAGE_BW=2048" "JTP_COST_STAGE_BW=8192" \
         "JTP_COST_FLUSH_BW=8192" "JTP_COST_ITER_D=0.8" "JTP_COST_ITER_D=1.3 JTP_COST_ITER_C=0.33" "JTP_COST_WG=3" "JTP_COST_RED_BW=1.5e6" "JTP_COST_RED_BW=6e6" "JTP_COST_OVERLAP=0.25" "JTP_COST_OVERLAP=1.0" \
         "JTP_COST_WAVE=0.75" "JTP_COST_WAVE=3" "JTP_COST_LANE=0.2" "JTP_MAX_BLOCK_LOG2_D=16" "JTP_REDUCE_MIN=16" "JTP_REDUCE_MIN=8"; do
    echo -n "$e : " >> $OUT
    env $e timeout -k 10 150 python3 tools/c3_time.py >> $OUT 2>&1 || exit 1
done
