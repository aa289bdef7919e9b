#!/bin/bash
# A/B of library builds on ONE box (boxes of the pool differ by +-2.5 %: never compare across gpurun calls).
#   bash tools/ab.sh OUTDIR "bench args" lib1 lib2 ...      (libN = a name under junctiontree_amd/lib/: libjtprop_NAME.so, or "cur")
# Runs every build REPS times (default 2), interleaved, and prints ms/step per run.
OUT=$1; ARGS=$2; shift 2
REPS=${REPS:-2}
mkdir -p $OUT
L=junction-tree_amd/junctiontree_amd/lib
for i in $(seq $REPS); do
  for n in "$@"; do
    if [ "$n" = cur ]; then env -u JTPROP_LIB python bench.py $ARGS --cpu-sample 0 > $OUT/${n}_$i.json 2>&1
    else JTPROP_LIB=$L/libjtprop_$n.so python bench.py $ARGS --cpu-sample 0 > $OUT/${n}_$i.json 2>&1; fi
  done
done
python tools/bsum.py $OUT/*.json | grep "ms/step\|unread"
