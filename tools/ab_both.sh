#!/bin/bash
# JTP_FLOW_BOTH experiment inside ONE gpurun call (diagnostic)
O=gpurun_out/ab_both.txt; : > $O
for rep in 1 2; do
for v in 0 1; do
  echo "== JTP_FLOW_BOTH=$v" >> $O
  JTP_FLOW_BOTH=$v JTP_MERGE_PHASES=0 timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1
  JTP_FLOW_BOTH=$v JTP_MERGE_PHASES=0 C3_SWEEP=1 timeout -k 10 120 python3 tools/c3_time.py >> $O 2>&1
  JTP_FLOW_BOTH=$v timeout -k 10 200 python3 tools/rank_time.py 8 30 2>&1 | tail -3 >> $O
done
done
cat $O
