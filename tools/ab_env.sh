#!/bin/bash
# Planner-knob sweep on config 3 inside ONE gpurun call (diagnostic)
O=gpurun_out/ab_env.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1; }
run A=default
run JTP_LONGEST_FIRST=0
run JTP_LANE_LOW=0
run JTP_LANE_LOW=3
run JTP_SETTLE_LEVEL_ELEMS=1e12
run JTP_SETTLE_LEVEL_ELEMS=0
run JTP_REDUCE_MIN=4
run JTP_REDUCE_MIN=64
run JTP_UNIT_RATIO=2
run JTP_UNIT_RATIO=16
run JTP_KEEP_ROWS_MB=0
run A=default
cat $O
