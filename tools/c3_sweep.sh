#!/bin/bash
# Diagnostic: config 3 under variations of the planner's cost-model constants (JTP_COST_*), one line each.
OUT=${1:-gpurun_out/c3sweep.txt}
: > $OUT
B="JTP_COST_MAX_CU=3 JTP_COST_STAGE_BW=4096 JTP_COST_LANE=0.1"
for e in "$B" "$B JTP_COST_EPI=0.5" "$B JTP_COST_STAGE_FIX=2" "$B JTP_COST_BW=3e6" "$B JTP_COST_STAGE_BW=2048" "$B JTP_COST_FLUSH_BW=4096" "$B JTP_COST_MAX_CU=2" \
         "$B JTP_COST_ITER_D=0.8 JTP_COST_ITER_C=0.6" "$B JTP_COST_EPI=0.5 JTP_COST_BW=3e6 JTP_COST_FLUSH_BW=4096"; do
    echo -n "$e : " >> $OUT
    env $e timeout -k 10 150 python3 tools/c3_time.py >> $OUT 2>&1 || exit 1
done
