#!/bin/bash
# Occupancy, second try: builds for 4 (product) and 5 (-DJT_FLOW_WAVES=5: 96 registers, the lean loops fit) waves per SIMD x the planner's
# per-kind LDS caps for plans of mostly unit cliques, config 3, inside ONE gpurun call (diagnostic).  prev = before the scratch was dropped.
L=$PWD/junction-tree_amd/junctiontree_amd/lib
O=gpurun_out/ab_occ2.txt; : > $O
run() { echo "== $V $*" >> $O; env "$@" timeout -k 10 200 python3 tools/c3_time.py >> $O 2>&1; }
for V in prev product w5 product w5; do
  if [ $V = product ]; then unset JTPROP_LIB; else export JTPROP_LIB=$L/libjtprop_$V.so; fi
  run A=default
  if [ $V != prev ]; then
    run JTP_UD_TABLE_LDS=4096
    run JTP_UD_TABLE_LDS=4096 JTP_UD_UNIT_LDS=16384
    run JTP_UD_TABLE_LDS=4096 C3_SWEEP=1
  fi
done
cat $O
