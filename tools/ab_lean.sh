#!/bin/bash
# The lean unit pass against the generic one (JTP_NO_LEAN=1) on config 3, both trees, inside ONE gpurun call (diagnostic)
O=gpurun_out/ab_lean.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 200 python3 tools/c3_time.py >> $O 2>&1; }
run A=lean
run JTP_NO_LEAN=1
run A=lean
run JTP_NO_LEAN=1
run C3_SWEEP=1
run C3_SWEEP=1 JTP_NO_LEAN=1
cat $O
