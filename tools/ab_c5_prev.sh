#!/bin/bash
# Multi-set plans: the product library against lib/libjtprop_prev.so, 8 / 64 / 512 sets, inside ONE gpurun call (diagnostic)
L=$PWD/junction-tree_amd/junctiontree_amd/lib
O=gpurun_out/ab_c5_prev.txt; : > $O
for rep in 1 2; do
for v in prev product; do
  if [ $v = product ]; then unset JTPROP_LIB; else export JTPROP_LIB=$L/libjtprop_$v.so; fi
  for n in 8 64; do
    echo "== $v $n sets" >> $O
    timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --batch $n --multiset 2>>$O | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O 2>&1
  done
done
done
unset JTPROP_LIB
echo "== product 512 sets" >> $O
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-profile --batch 512 --multiset 2>>$O | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O 2>&1
cat $O
