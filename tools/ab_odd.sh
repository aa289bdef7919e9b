#!/bin/bash
# Odd-cardinality trees: planner knobs inside ONE gpurun call (diagnostic)
O=gpurun_out/ab_odd.txt; : > $O
run() { echo "== $*" >> $O; for a in "3 13 6 63 f32" "3 12 6 63 f64" "5 9 4 63 f32" "6 8 4 63 f32"; do env "$@" timeout -k 10 120 python3 tools/odd_time.py $a 2>&1 | grep "mixed" >> $O; done; }
for rep in 1 2; do
run A=default
run JTP_NO_VGROUPS=1
done
cat $O
