L=$PWD/junction-tree_amd/junctiontree_amd/lib
O=gpurun_out/ab_sweep.txt; : > $O
for v in prev w5 w6 prev w5 w6; do
  echo "== $v" >> $O
  JTPROP_LIB=$L/libjtprop_$v.so C3_SWEEP=1 timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1
done
cat $O
