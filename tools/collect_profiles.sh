#!/bin/bash
# Run on the GPU box (through gpurun): bench lines, rocprofv3 kernel trace + stats, PMC passes for HBM traffic
# (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, never combined with a trace domain).  Everything lands under
# gpurun_out/prof/; tools/summarize_profiles.py turns it into gpurun_out/prof/summary/, whose files are committed
# under profiles/ as r06_* (tools/publish_profiles.py).  Every text file starts with the id of the library build it was measured on (the JSON
# bench lines carry it in config.library; hbm_traffic*.json in "source_id"): bench.py quotes roofline.traffic only
# when that id is the running library's.
#   bash tools/collect_profiles.sh            (needs lib/libjtprop_stamps.so = build.py --out ... -DJT_STAMPS of the same sources)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py"
L=junction-tree_amd/junctiontree_amd/lib
ID="# library build: $(cat $L/BUILD_ID 2>/dev/null | tr '\n' ' ')"
hdr() { echo "$ID" > $1; }          # start a text file with the build id
# the driver's form of the command first: the headline workload AND the sub-results for the other BASELINE configs in one line
timeout 600 $B --cpu-all-cores > $OUT/bench.json 2> $OUT/bench.err
timeout 600 $B --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
hdr $OUT/per_launch.txt
timeout 300 $B --steps 30 --warmup 3 --cpu-sample 0 --level-launches --per-launch > $OUT/bench_level.json 2>> $OUT/per_launch.txt
timeout 300 $B --steps 20 --warmup 2 --cpu-sample 0 --no-profile --batch 4 > $OUT/bench_batch4.json 2>/dev/null
timeout 300 $B --steps 10 --warmup 2 --cpu-sample 0 --batch 16 --share > $OUT/bench_c5_share16.json 2>/dev/null
for n in 8 16 64; do timeout 300 $B --steps 10 --warmup 2 --cpu-sample 0 --batch $n --multiset > $OUT/bench_c5_multiset$n.json 2>/dev/null; done
# configs[4] at its stated count on ONE device: 512 evidence sets, 64 groups of eight per pass over a table
timeout 600 $B --steps 3 --warmup 1 --cpu-sample 0 --batch 512 --multiset > $OUT/bench_c5_multiset512.json 2>/dev/null
# the API call on config 3, stage by stage, on the min-fill tree and on the column-sweep tree of SURVEY.md 8d
hdr $OUT/c3_api.txt; timeout 600 python3 tools/run_c3.py >> $OUT/c3_api.txt 2>&1
hdr $OUT/c3_api_column_sweep.txt; timeout 600 python3 tools/run_c3.py 167 8 sweep >> $OUT/c3_api_column_sweep.txt 2>&1
timeout 600 $B --steps 20 --warmup 3 --config c2 > $OUT/bench_c2.json 2>/dev/null
timeout 900 $B --steps 10 --warmup 3 --config c3 > $OUT/bench_c3.json 2>/dev/null
# a second, idle plan on the device (round 2: ticket order for everybody, +10 %; round 3: tickets only while
# another plan's propagate is in flight)
for k in 0 1 0 1; do timeout 300 $B --steps 40 --warmup 10 --cpu-sample 0 --idle-plans $k >> $OUT/idle_plans_ab.jsonl 2>/dev/null; done
hdr $OUT/rank_time_8.txt; timeout 300 python3 tools/rank_time.py 8 30 >> $OUT/rank_time_8.txt 2>&1
hdr $OUT/odd_cardinalities.txt
for a in "3 13 6 63 f32" "3 12 6 63 f64" "5 9 4 63 f32" "6 8 4 63 f32" "7 7 3 63 f32"; do timeout 300 python3 tools/odd_time.py $a >> $OUT/odd_cardinalities.txt 2>&1; done
# ... the same trees with the chunks that do not exist back in the block lists (rounds 2-4), and one tree level by level
echo "# JTP_KEEP_INVALID=1 (every chunk a workgroup of every propagate, as in rounds 2-4):" >> $OUT/odd_cardinalities.txt
for a in "3 13 6 63 f32" "3 12 6 63 f64" "5 9 4 63 f32" "6 8 4 63 f32" "7 7 3 63 f32"; do JTP_KEEP_INVALID=1 timeout 300 python3 tools/odd_time.py $a >> $OUT/odd_cardinalities.txt 2>&1; done
echo "# ODD_LEVELS=1 (one launch per level, every launch timed), cardinality 3 width 13 f32:" >> $OUT/odd_cardinalities.txt
ODD_LEVELS=1 timeout 300 python3 tools/odd_time.py 3 13 6 63 f32 >> $OUT/odd_cardinalities.txt 2>&1
if [ -f $L/libjtprop_stamps.so ]; then
  export JTPROP_LIB=$L/libjtprop_stamps.so JTP_DEBUG=2
  hdr $OUT/stage_times.txt; timeout 300 python3 tools/stamps.py >> $OUT/stage_times.txt 2>&1
  hdr $OUT/stage_times_c2.txt; STAMPS_SUMMARY=1 timeout 300 python3 tools/stamps.py c2 1000 >> $OUT/stage_times_c2.txt 2>&1
  hdr $OUT/stage_times_multiset8.txt; timeout 300 python3 tools/stamps.py multi 8 >> $OUT/stage_times_multiset8.txt 2>&1
  hdr $OUT/stage_times_rank0_of_8.txt; STAMPS_SUMMARY=1 timeout 300 python3 tools/stamps.py ranks 8 0 >> $OUT/stage_times_rank0_of_8.txt 2>&1
  unset JTPROP_LIB JTP_DEBUG
fi
echo "benches done" > $OUT/progress.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > $OUT/kt_bench.json 2> $OUT/kt.err
# ... and of the whole API call on config 3 (jt_eval_batch, the message-passing kernels, jt_marginals)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c3api -o kt -- python3 tools/run_c3.py > $OUT/kt_c3api.txt 2> $OUT/kt_c3api.err
echo "kernel trace done" >> $OUT/progress.txt
# HBM traffic: case name -> bench arguments
pmc() {   # $1 = case, rest = bench arguments
  local c=$1; shift
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$c -o p -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-profile "$@" > /dev/null 2> $OUT/fetch_$c.err
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$c -o p -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-profile "$@" > /dev/null 2> $OUT/write_$c.err
  echo "pmc $c done" >> $OUT/progress.txt
}
pmc single
pmc multiset64 --batch 64 --multiset
pmc c2 --config c2
pmc c3 --config c3
JTP_BENCH_C3_SWEEP=1 pmc c3_sweep --config c3
JTP_BENCH_NO_COVER=1 pmc c3_full_tables --config c3
# what bounds jt_multi_flow: busy cycles of the vector and LDS pipes against the kernel's cycles (a few counters per pass)
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "VALUBusy" "MemUnitStalled"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/valu_$n -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-profile --batch 64 --multiset > /dev/null 2> $OUT/valu_$n.err
done
# ... and of the message-passing kernel(s) on config 3 (tools/pmc_c3.sh: the same counters over tools/c3_time.py)
hdr $OUT/counters_c3.txt
echo "# bash tools/pmc_c3.sh: rocprofv3 --pmc <a few counters per pass> -- python3 tools/c3_time.py; means over the kernel's launches" >> $OUT/counters_c3.txt
bash tools/pmc_c3.sh > /dev/null 2>&1
grep "flow" gpurun_out/pmc_c3/summary.txt >> $OUT/counters_c3.txt
# keep what travels back small: the per-dispatch traces are reduced here
python3 tools/summarize_profiles.py $OUT
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
