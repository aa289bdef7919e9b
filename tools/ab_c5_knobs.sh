O=gpurun_out/ab_c5_knobs.txt; echo "# library build: $(cat junction-tree_amd/junctiontree_amd/lib/BUILD_ID | tr '\n' ' ')" > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --no-configs --batch 64 --multiset 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5x64 ms_per_step', round(d['ms_per_step'], 4))" >> $O 2>&1; }
run A=default
run JTP_MULTI_MIN_BLOCK_LOG2=15
run JTP_MULTI_MIN_BLOCK_LOG2=15 JTP_TARGET_BLOCKS=256
run JTP_MULTI_MIN_BLOCK_LOG2=15 JTP_TARGET_BLOCKS=2048
run JTP_MULTI_MIN_BLOCK_LOG2=14 JTP_TARGET_BLOCKS=2048
run JTP_LONGEST_FIRST=0
run JTP_REDUCE_MIN=4
run JTP_REDUCE_MIN=16
run JTP_TOP_MIN_LOOP=2
run JTP_TOP_MIN_LOOP=4
run JTP_LANE_LOW=1
run JTP_LANE_LOW=3
run A=default
cat $O
