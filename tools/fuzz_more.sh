O=gpurun_out/gpu_fuzz_more.txt
echo "# library build: $(cat junction-tree_amd/junctiontree_amd/lib/BUILD_ID | tr '\n' ' ') - a second, larger run with other seeds" > $O
run() { echo "$*:" >> $O; env "$@" 2>&1 | tail -1 | cut -c1-700 >> $O; }
run timeout -k 10 500 python3 tools/gpu_fuzz.py 3000 110000
run timeout -k 10 300 python3 tools/gpu_fuzz_evidence.py 1000 120000
run timeout -k 10 500 python3 tools/gpu_fuzz_api.py 4000 130000
run FUZZ_BIG=1 timeout -k 10 300 python3 tools/gpu_fuzz.py 1000 140000
run timeout -k 10 500 python3 tools/gpu_fuzz_compact.py 600 150000
cat $O
