#!/bin/bash
# Odd-cardinality trees: library variants and planner knobs inside ONE gpurun call (diagnostic)
L=$PWD/junction-tree_amd/junctiontree_amd/lib
O=gpurun_out/ab_odd.txt; : > $O
run() { echo "== $*" >> $O; for a in "3 13 6 63 f32" "3 12 6 63 f64" "5 9 4 63 f32" "6 8 4 63 f32" "7 7 3 63 f32"; do env "$@" timeout -k 10 120 python3 tools/odd_time.py $a 2>&1 | grep "mixed\|padded" >> $O; done; }
for rep in 1 2; do
run A=default
run JTP_KEEP_INVALID=1
[ -f $L/libjtprop_mw4.so ] && run JTPROP_LIB=$L/libjtprop_mw4.so
done
cat $O
