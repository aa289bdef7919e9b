#!/bin/bash
# A/B of library variants and planner settings on config 3 inside ONE gpurun call (diagnostic; results of the JTP_FLOW_DEBUG runs are wrong by design)
L=$PWD/junction-tree_amd/junctiontree_amd/lib
O=gpurun_out/ab_c3.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1; }
run A=product
for v in w5 w6; do
  [ -f $L/libjtprop_$v.so ] || continue
  for b in 0 24576 20480 16384; do run JTPROP_LIB=$L/libjtprop_$v.so C3_LDS_BUDGET=$b; done
done
for b in 24576 20480 16384; do run C3_LDS_BUDGET=$b; done
if [ -f $L/libjtprop_exp.so ]; then
  for d in 0 16 32 48; do run JTPROP_LIB=$L/libjtprop_exp.so JTP_FLOW_DEBUG=$d; done
fi
run A=product_again
cat $O
