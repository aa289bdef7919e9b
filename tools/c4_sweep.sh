#!/bin/bash
# Diagnostic: config 4 (bench.py), one line per variation:  tools/c4_sweep.sh OUT [consts|sizes]
#   consts: the planner's cost-model constants (JTP_COST_*);  sizes: the workgroup-size knobs
OUT=${1:-gpurun_out/c4sweep.txt}
MODE=${2:-consts}
: > $OUT
run() {
    env $1 timeout -k 10 100 python3 bench.py --cpu-sample 0 --steps 100 > /tmp/c4s.json 2>/dev/null || exit 1
    echo "$1 : $(python3 tools/bsum.py /tmp/c4s.json | tr '\n' ' ' | tr -s ' ')" >> $OUT
}
if [ "$MODE" = consts ]; then
for e in "X=0" "JTP_COST_LANE=0.03" "JTP_COST_LANE=0.3" "JTP_COST_EPI=0.25" "JTP_COST_EPI=1.0" "JTP_COST_STAGE_FIX=8" "JTP_COST_STAGE_FIX=2" \
         "JTP_COST_ITER_D=0.8" "JTP_COST_ITER_D=0.3" "JTP_COST_WAVE=3" "JTP_COST_WAVE=0.5" "JTP_COST_MAX_CU=2" "JTP_COST_MAX_CU=4" "JTP_COST_BW=3e6" "JTP_COST_BW=7e6" \
         "JTP_COST_OVERLAP=1.0" "JTP_COST_OVERLAP=0.0" "JTP_COST_RED_FIX=10" "JTP_COST_RED_FIX=1" "JTP_COST_WG=4" "JTP_COST_STAGE_BW=16384" "JTP_COST_STAGE_BW=1024" \
         "JTP_COST_FLUSH_FIX=3" "JTP_COST_FLUSH_BW=4096" "JTP_COST_ITER_C=0.8" "JTP_COST_ITER_C=0.25" "JTP_SEARCH_ALL=0"; do run "$e"; done
else
for e in "X=0" "JTP_TARGET_BLOCKS=512" "JTP_TARGET_BLOCKS=2048" "JTP_TARGET_BLOCKS=4096" "JTP_TARGET_BLOCKS_D=512" "JTP_TARGET_BLOCKS_D=2048" "JTP_TARGET_BLOCKS_D=4096" \
         "JTP_MAX_BLOCK_LOG2=15" "JTP_MAX_BLOCK_LOG2=14" "JTP_MIN_BLOCK_LOG2=12" "JTP_MIN_BLOCK_LOG2=14" "JTP_TINY_LEVEL_ELEMS=100000" "JTP_TINY_LEVEL_ELEMS=8000000" "X=1"; do run "$e"; done
fi
