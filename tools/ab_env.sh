#!/bin/bash
# A/B of ENVIRONMENT settings (planner knobs) on ONE box with the current library.
#   bash tools/ab_env.sh OUTDIR "bench args" name1 "ENV1=.. ENV2=.." name2 "..." ...     (use "-" for no variables)
OUT=$1; ARGS=$2; shift 2
REPS=${REPS:-2}
mkdir -p $OUT
names=(); envs=()
while [ $# -gt 1 ]; do names+=("$1"); envs+=("$2"); shift 2; done
for i in $(seq $REPS); do
  for j in "${!names[@]}"; do
    e="${envs[$j]}"; [ "$e" = "-" ] && e=""
    env $e python bench.py $ARGS --cpu-sample 0 > $OUT/${names[$j]}_$i.json 2>&1
  done
done
python tools/bsum.py $OUT/*.json | grep "ms/step\|unread"
