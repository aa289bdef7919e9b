#!/bin/bash
# A/B (round 6): did the folded-marginal code in the dataflow kernels move the HOT path (plans without folded tasks)?  The product against
# the tree of the commit before the fold with ITS library (build 879098159e11), exported to .ab_prev/ (git archive 30b1db4 + the saved .so),
# alternating, inside ONE gpurun call
R=$PWD
O=$R/gpurun_out/ab_fold_hot.txt; echo "# product: $(cat junction-tree_amd/junctiontree_amd/lib/BUILD_ID | tr '\n' ' ') against .ab_prev = build 879098159e11 (commit 30b1db4)" > $O
for rep in 1 2 3; do
for v in prev product; do
  if [ $v = product ]; then cd $R; else cd $R/.ab_prev; fi
  echo "== $v" >> $O
  timeout -k 10 120 python3 tools/c3_time.py | sed 's/^/c3 min-fill: /' >> $O 2>&1
  C3_SWEEP=1 timeout -k 10 120 python3 tools/c3_time.py | sed 's/^/c3 column sweep: /' >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --cpu-sample 0 --no-profile --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 ms_per_step', d['ms_per_step'], d['config']['library'])" >> $O 2>&1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-profile --config c2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 ms_per_step', d['ms_per_step'])" >> $O 2>&1
done
done
cd $R; cat $O
