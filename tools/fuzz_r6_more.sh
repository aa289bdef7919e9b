#!/bin/bash
# Round 6, a second and larger run of the GPU fuzz batteries on the final build, other seeds (appended to profiles/r06_gpu_fuzz.txt)
O=gpurun_out/gpu_fuzz_r6_more.txt
echo "# second run, other seeds; library build: $(cat junction-tree_amd/junctiontree_amd/lib/BUILD_ID 2>/dev/null | tr '\n' ' ')" > $O
run() { echo "$*:" >> $O; env "$@" 2>&1 | tail -2 | cut -c1-900 >> $O; }
run timeout -k 10 300 python3 tools/gpu_fuzz_fold.py 4000 300000
run timeout -k 10 200 python3 tools/gpu_fuzz.py 2000 310000
run timeout -k 10 200 python3 tools/gpu_fuzz_evidence.py 1000 320000
run JTP_EF_SHARE=1 timeout -k 10 200 python3 tools/gpu_fuzz_evidence.py 1000 330000
run timeout -k 10 300 python3 tools/gpu_fuzz_api.py 4000 340000
run JTP_FOLD=1 JTP_TINY_LEVEL_ELEMS=0 timeout -k 10 300 python3 tools/gpu_fuzz_api.py 4000 350000
run JTP_FOLD=1 JTP_TINY_LEVEL_ELEMS=0 FUZZ_WIDE=1 timeout -k 10 300 python3 tools/gpu_fuzz_api.py 2000 360000
run timeout -k 10 300 python3 tools/gpu_fuzz_wide.py 400 370000
run timeout -k 10 600 python3 tools/gpu_fuzz_compact.py 400 380000
cat $O
