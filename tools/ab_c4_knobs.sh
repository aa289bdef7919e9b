#!/bin/bash
# Planner knobs of the headline workload (config 4) on the final kernels, one box (profiles/r06_ab_c4_knobs.txt)
O=gpurun_out/ab_c4_knobs.txt; echo "# library build: $(cat junction-tree_amd/junctiontree_amd/lib/BUILD_ID | tr '\n' ' ')" > $O
run() { echo "== $*" >> $O; for i in 1 2; do env "$@" timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --cpu-sample 0 --no-profile --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 ms_per_step', round(d['ms_per_step'], 4))" >> $O 2>&1; done; }
run A=default
run JTP_TARGET_BLOCKS=768
run JTP_TARGET_BLOCKS=1536
run JTP_TARGET_BLOCKS=2048
run JTP_TARGET_BLOCKS_D=2048
run JTP_MAX_BLOCK_LOG2_D=14
run JTP_MAX_BLOCK_LOG2_D=16
run JTP_MAX_BLOCK_LOG2=15
run JTP_TOP_SHARE=0.06
run JTP_TOP_SHARE=0.25
run JTP_TOP_ROWS2=1024
run JTP_TOP_ROWS2=4096
run JTP_TOP_MIN_LOOP=2
run JTP_TOP_MIN_LOOP=4
run JTP_SETTLE_LEVEL_ELEMS=4194304
run JTP_SETTLE_LEVEL_ELEMS=33554432
run JTP_KEEP_ROWS_MB=64
run JTP_KEEP_ROWS_MB=256
run JTP_REDUCE_MIN=4
run JTP_LANE_LOW=1
run JTP_MERGE_PHASES=0
run A=default
cat $O
