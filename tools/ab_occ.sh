#!/bin/bash
# Occupancy of the dataflow kernels under the lean unit pass (builds with -DJT_FLOW_WAVES=5/6/8) x LDS budget of the planner, config 3,
# inside ONE gpurun call (diagnostic).  bash tools/ab_occ.sh w5 w6 w8
L=$PWD/junction-tree_amd/junctiontree_amd/lib
O=gpurun_out/ab_occ.txt; : > $O
for v in product "$@" product; do
  if [ $v = product ]; then unset JTPROP_LIB; else export JTPROP_LIB=$L/libjtprop_$v.so; fi
  for b in 0 24576 20480 16384; do
    echo "== $v lds_budget=$b" >> $O
    C3_LDS_BUDGET=$b timeout -k 10 200 python3 tools/c3_time.py >> $O 2>&1
  done
  echo "== $v sweep tree" >> $O
  C3_SWEEP=1 timeout -k 10 200 python3 tools/c3_time.py >> $O 2>&1
done
cat $O
