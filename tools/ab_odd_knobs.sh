#!/bin/bash
# Odd-cardinality trees: planner knobs (workgroups per level, rows) inside ONE gpurun call (diagnostic)
O=gpurun_out/ab_odd_knobs.txt; : > $O
run() { echo "== $*" >> $O; for a in "3 13 6 63 f32" "3 12 6 63 f64" "5 9 4 63 f32" "6 8 4 63 f32"; do env "$@" timeout -k 10 120 python3 tools/odd_time.py $a 2>&1 | grep "mixed" | cut -c40-120 >> $O; done; }
run A=default
run JTP_TARGET_BLOCKS=512
run JTP_TARGET_BLOCKS=2048
run JTP_TARGET_BLOCKS=4096
run JTP_REDUCE_MIN=8
run JTP_MIN_BLOCK_LOG2=14
run JTP_MIN_BLOCK_LOG2=12
run JTP_TOP_ROWS2=0
run A=default
cat $O
