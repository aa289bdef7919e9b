#!/bin/bash
# Diagnostic: config 2 (chain) without the four-iteration rule for tiny levels, under variations of the cost model.
OUT=${1:-gpurun_out/c2sweep.txt}
: > $OUT
T="JTP_TINY_LEVEL_ELEMS=100000"
for e in "$T" "$T JTP_COST_RED_FIX=1" "$T JTP_COST_RED_FIX=8" "$T JTP_COST_ITER_C=0.2 JTP_COST_ITER_D=0.25" "$T JTP_COST_ITER_C=1.0" "$T JTP_COST_ITER_D=1.2" "$T JTP_COST_ITER_D=0.25" \
         "$T JTP_COST_STAGE_FIX=2" "$T JTP_COST_STAGE_FIX=10" "$T JTP_COST_EPI=0.1" "$T JTP_COST_EPI=1.5" "$T JTP_COST_WG=0.5" "$T JTP_COST_WG=5" "$T JTP_COST_FLUSH_FIX=0.2" "$T JTP_COST_FLUSH_FIX=4" \
         "$T JTP_COST_STAGE_BW=16384" "$T JTP_COST_LANE=0.03" "$T JTP_COST_WAVE=0.5" "$T JTP_COST_WAVE=4"; do
    env $e timeout -k 10 100 python3 bench.py --config c2 --cpu-sample 0 --steps 30 > /tmp/c2s.json 2>/dev/null || exit 1
    echo "$e : $(python3 tools/bsum.py /tmp/c2s.json | tr '\n' ' ' | tr -s ' ')" >> $OUT
done
for i in 1 2; do
env $T timeout -k 10 100 python3 bench.py --cpu-sample 0 --steps 100 > /tmp/c4s.json 2>/dev/null || exit 1
echo "C4 $T : $(python3 tools/bsum.py /tmp/c4s.json | tr '\n' ' ' | tr -s ' ')" >> $OUT
timeout -k 10 100 python3 bench.py --cpu-sample 0 --steps 100 > /tmp/c4s.json 2>/dev/null || exit 1
echo "C4 default : $(python3 tools/bsum.py /tmp/c4s.json | tr '\n' ' ' | tr -s ' ')" >> $OUT
done
