#!/bin/bash
# Run on the GPU box (through gpurun): bench lines, rocprofv3 kernel trace + stats, PMC passes for HBM traffic
# (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, never combined with a trace domain).  Everything lands under
# gpurun_out/prof/; tools/summarize_profiles.py turns it into the files committed under profiles/.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py"
timeout 300 $B --cpu-all-cores > $OUT/bench.json 2> $OUT/bench.err
timeout 300 $B --steps 30 --warmup 3 --cpu-sample 0 --level-launches --per-launch > $OUT/bench_level.json 2> $OUT/per_launch.txt
timeout 300 $B --steps 20 --warmup 2 --cpu-sample 0 --no-profile --batch 4 > $OUT/bench_batch4.json 2>/dev/null
timeout 300 $B --steps 10 --warmup 2 --cpu-sample 0 --batch 16 --share > $OUT/bench_c5_share16.json 2>/dev/null
for n in 8 16 64; do timeout 300 $B --steps 10 --warmup 2 --cpu-sample 0 --batch $n --multiset > $OUT/bench_c5_multiset$n.json 2>/dev/null; done
timeout 600 $B --steps 20 --warmup 3 --config c2 > $OUT/bench_c2.json 2>/dev/null
timeout 600 python3 tools/run_c3.py > $OUT/c3.txt 2>&1
JTP_DEBUG=2 timeout 300 python3 tools/stamps.py > $OUT/stage_times.txt 2>&1
JTP_DEBUG=2 timeout 300 python3 tools/stamps.py multi 8 > $OUT/stage_times_multiset8.txt 2>&1
timeout 300 python3 tools/rank_time.py 8 20 > $OUT/rank_time_8.txt 2>&1
echo "benches done" > $OUT/progress.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > $OUT/kt_bench.json 2> $OUT/kt.err
echo "kernel trace done" >> $OUT/progress.txt
# HBM traffic: case name -> bench arguments
pmc() {   # $1 = case, rest = bench arguments
  local c=$1; shift
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$c -o p -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-profile "$@" > /dev/null 2> $OUT/fetch_$c.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$c -o p -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-profile "$@" > /dev/null 2> $OUT/write_$c.err
  echo "pmc $c done" >> $OUT/progress.txt
}
pmc single
pmc share16 --batch 16 --share
pmc multiset16 --batch 16 --multiset
pmc multiset64 --batch 64 --multiset
timeout 600 rocprofv3 --pmc VALUBusy --output-format csv -d $OUT/valu_multiset16 -o p -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-profile --batch 16 --multiset > /dev/null 2> $OUT/valu.err
# keep what travels back small: the per-dispatch traces are reduced here
python3 tools/summarize_profiles.py $OUT
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
