#!/bin/bash
# A/B of two library builds inside ONE gpurun call: config 3 (tools/c3_time.py), config 4 / 2 / 5 through bench.py  (diagnostic)
#   bash tools/ab_libs.sh prev          -> lib/libjtprop_prev.so against the product library, A B A B
L=$PWD/junction-tree_amd/junctiontree_amd/lib
O=gpurun_out/ab_libs.txt; : > $O
for rep in 1 2; do
for v in "$@" product; do
  if [ $v = product ]; then unset JTPROP_LIB; else export JTPROP_LIB=$L/libjtprop_$v.so; fi
  echo "== $v" >> $O
  timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --cpu-sample 0 --no-profile --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 ms_per_step', d['ms_per_step'])" >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --batch 64 --multiset 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5x64 ms_per_step', d['ms_per_step'])" >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --config c2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 ms_per_step', d['ms_per_step'])" >> $O 2>&1
done
done
cat $O
