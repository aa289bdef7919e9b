#!/bin/bash
# A/B of the product library against lib/libjtprop_prev.so on configs 2, 3 (generic pass forced: JTP_NO_LEAN=1, and lean), 4, 5 and a rank share,
# inside ONE gpurun call (diagnostic)
L=$PWD/junction-tree_amd/junctiontree_amd/lib
O=gpurun_out/ab_prev.txt; : > $O
for rep in 1 2; do
for v in prev product; do
  if [ $v = product ]; then unset JTPROP_LIB; else export JTPROP_LIB=$L/libjtprop_$v.so; fi
  echo "== $v" >> $O
  timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1
  JTP_NO_LEAN=1 timeout -k 10 120 python3 tools/c3_time.py | sed 's/^/no-lean /' >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --cpu-sample 0 --no-profile --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 ms_per_step', d['ms_per_step'])" >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --batch 64 --multiset 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5x64 ms_per_step', d['ms_per_step'])" >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --config c2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 ms_per_step', d['ms_per_step'])" >> $O 2>&1
  JTP_FAKE_COMM=1 timeout -k 10 200 python3 tools/rank_time.py 8 2>/dev/null | tail -3 >> $O 2>&1
done
done
cat $O
