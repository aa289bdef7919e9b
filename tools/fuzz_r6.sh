#!/bin/bash
# Round 6: the GPU fuzz batteries on the lean unit pass (public API: covers -> unit cliques) and on the active lists of multi-set plans
# (JTP_EF_SHARE=1: an evidence-free set whatever the number of sets), fresh seeds, one gpurun call
O=gpurun_out/gpu_fuzz_r6.txt
echo "# library build: $(cat junction-tree_amd/junctiontree_amd/lib/BUILD_ID 2>/dev/null | tr '\n' ' ')" > $O
run() { echo "$*:" >> $O; env "$@" 2>&1 | tail -2 | cut -c1-900 >> $O; }
run timeout -k 10 300 python3 tools/gpu_fuzz.py 800 170000
run timeout -k 10 300 python3 tools/gpu_fuzz_evidence.py 300 171000
run JTP_EF_SHARE=1 timeout -k 10 300 python3 tools/gpu_fuzz_evidence.py 400 176000
run timeout -k 10 400 python3 tools/gpu_fuzz_api.py 1500 172000
run JTP_EF_SHARE=1 timeout -k 10 400 python3 tools/gpu_fuzz_api.py 1500 177000
run FUZZ_BIG=1 timeout -k 10 300 python3 tools/gpu_fuzz.py 300 173000
run FUZZ_WIDE=1 timeout -k 10 400 python3 tools/gpu_fuzz_api.py 800 174000
run JTP_EF_SHARE=1 FUZZ_WIDE=1 timeout -k 10 400 python3 tools/gpu_fuzz_api.py 800 178000
run timeout -k 10 300 python3 tools/gpu_fuzz_wide.py 200 175000
run timeout -k 10 600 python3 tools/gpu_fuzz_compact.py 200 190000
# the marginal tasks folded into the propagate (plans of latency-bound levels are built without them: planned here as large trees are)
run JTP_FOLD=1 JTP_TINY_LEVEL_ELEMS=0 timeout -k 10 400 python3 tools/gpu_fuzz_api.py 1500 179000
run JTP_FOLD=1 JTP_TINY_LEVEL_ELEMS=0 FUZZ_WIDE=1 timeout -k 10 400 python3 tools/gpu_fuzz_api.py 800 180000
run timeout -k 10 900 python3 tools/gpu_fuzz_fold.py 2000 210000
cat $O
