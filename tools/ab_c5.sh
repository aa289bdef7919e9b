#!/bin/bash
# Multi-set plans: active lists per task (default from 8 groups on) against fixed groups without an evidence-free set (JTP_NO_EF_SHARE=1), inside ONE gpurun call
O=gpurun_out/ab_c5.txt; : > $O
run() { echo "== $*" >> $O; env "${@:2}" timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --batch $1 --multiset 2>>$O | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'Zerr', d.get('config',{}).get('Z_rel_err'))" >> $O 2>&1; }
run 64 A=lists
run 64 JTP_NO_EF_SHARE=1
run 64 A=lists
run 16 A=default
run 16 JTP_EF_SHARE=1
run 8 A=default
run 8 JTP_EF_SHARE=1
run 512 A=lists
cat $O
