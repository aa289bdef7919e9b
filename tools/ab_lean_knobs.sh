#!/bin/bash
# Planner knobs under the lean unit pass on config 3 inside ONE gpurun call (diagnostic)
O=gpurun_out/ab_lean_knobs.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 200 python3 tools/c3_time.py >> $O 2>&1; }
run A=default
run JTP_TARGET_BLOCKS=512
run JTP_TARGET_BLOCKS=768
run JTP_TARGET_BLOCKS=1024
run JTP_TARGET_BLOCKS=1536
run JTP_TARGET_BLOCKS=3072
run JTP_COST_ITER_C=0.1
run JTP_COST_ITER_C=0.2
run JTP_COST_ITER_C=0.1 JTP_COST_STAGE_FIX=5
run JTP_COST_ITER_C=0.1 JTP_COST_STAGE_FIX=5 JTP_TARGET_BLOCKS=1024
run JTP_COST_ITER_C=0.1 JTP_COST_STAGE_FIX=5 JTP_TARGET_BLOCKS=768
run JTP_COST_ITER_C=0.1 JTP_COST_EPI=0.25
run JTP_COST_ITER_C=0.05 JTP_COST_STAGE_FIX=6 JTP_COST_WG=3
run JTP_COST_MAX_CU=5
run JTP_REDUCE_MIN=8
run A=default
cat $O
