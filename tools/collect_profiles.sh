#!/bin/bash
# Run on the GPU box (through gpurun): bench lines, rocprofv3 kernel trace + stats, two PMC passes for
# HBM traffic.  Everything lands under gpurun_out/prof/; tools/summarize_profiles.py turns it into the
# committed files under profiles/.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
timeout 300 python3 bench.py --steps 30 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python3 bench.py --steps 30 --warmup 3 --cpu-sample 0 --level-launches --per-launch > $OUT/bench_level.json 2> $OUT/per_launch.txt
timeout 300 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --batch 4 > $OUT/bench_batch4.json 2>/dev/null
timeout 300 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --batch 16 --share > $OUT/bench_c5_shared.json 2>/dev/null
timeout 600 python3 bench.py --steps 5 --warmup 1 --config c2 > $OUT/bench_c2.json 2>/dev/null
JTP_DEBUG=2 timeout 300 python3 tools/stamps.py > $OUT/stage_times.txt 2>&1
timeout 300 python3 tools/rank_time.py 8 20 > $OUT/rank_time_8.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > $OUT/kt_bench.json 2> $OUT/kt.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-profile > /dev/null 2> $OUT/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-profile > /dev/null 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
# keep what travels back small: the per-dispatch traces are reduced here
python3 tools/summarize_profiles.py $OUT
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
