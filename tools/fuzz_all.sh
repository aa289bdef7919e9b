#!/bin/bash
# The GPU fuzz batteries with fresh seeds, one gpurun call (profiles/rNN_gpu_fuzz.txt is made from its output)
O=gpurun_out/gpu_fuzz.txt
ID="# library build: $(cat junction-tree_amd/junctiontree_amd/lib/BUILD_ID 2>/dev/null | tr '\n' ' ')"
echo "$ID" > $O
run() { echo "$*:" >> $O; env "$@" 2>&1 | tail -2 | cut -c1-900 >> $O; }
run timeout -k 10 300 python3 tools/gpu_fuzz.py 800 70000
run timeout -k 10 300 python3 tools/gpu_fuzz_evidence.py 300 71000
run timeout -k 10 400 python3 tools/gpu_fuzz_api.py 1500 72000
run FUZZ_BIG=1 timeout -k 10 300 python3 tools/gpu_fuzz.py 300 73000
run FUZZ_WIDE=1 timeout -k 10 400 python3 tools/gpu_fuzz_api.py 800 74000
run timeout -k 10 300 python3 tools/gpu_fuzz_wide.py 200 75000
run timeout -k 10 600 python3 tools/gpu_fuzz_compact.py 200 90000
cat $O
