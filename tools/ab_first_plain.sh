#!/bin/bash
# Experiment: the first staging attempt of a dataflow workgroup with ordinary (L2-cacheable) loads (JTP_FLOW_DEBUG=64) against through-to-memory
# loads from the start, configs 2, 3 (both trees), 4, 5, a rank share - inside ONE gpurun call (diagnostic)
O=gpurun_out/ab_first_plain.txt; : > $O
for rep in 1 2; do
for v in 0 64; do
  export JTP_FLOW_DEBUG=$v
  echo "== JTP_FLOW_DEBUG=$v" >> $O
  timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1
  C3_SWEEP=1 timeout -k 10 120 python3 tools/c3_time.py | sed 's/^/sweep /' >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --cpu-sample 0 --no-profile --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 ms_per_step', d['ms_per_step'], 'Zerr', d['config'].get('Z_rel_err'))" >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --batch 64 --multiset 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5x64 ms_per_step', d['ms_per_step'])" >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --config c2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 ms_per_step', d['ms_per_step'])" >> $O 2>&1
  JTP_FAKE_COMM=1 timeout -k 10 200 python3 tools/rank_time.py 8 2>/dev/null | tail -2 | cut -c1-150 >> $O 2>&1
done
done
unset JTP_FLOW_DEBUG
cat $O
