#!/bin/bash
# Planner-knob sweep on config 3 inside ONE gpurun call (diagnostic)
O=gpurun_out/ab_env.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1; }
run A=default
run JTP_MERGE_PHASES=0
for b in 16384 20480 22528; do
run JTP_TARGET_BLOCKS=1024 C3_LDS_BUDGET=$b
run JTP_TARGET_BLOCKS=512 C3_LDS_BUDGET=$b
done
run JTP_TARGET_BLOCKS=1024 JTP_TARGET_BLOCKS_D=2048 C3_LDS_BUDGET=20480
run JTP_TARGET_BLOCKS=2048 JTP_TARGET_BLOCKS_D=1024 C3_LDS_BUDGET=20480
run A=default
cat $O
