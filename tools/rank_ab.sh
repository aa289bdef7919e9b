#!/bin/bash
# rank-share timing (tools/rank_time.py) under several environment settings on one box: bash tools/rank_ab.sh OUT name1 "ENV.." name2 "ENV.." ...
OUT=$1; shift; mkdir -p $OUT
while [ $# -gt 1 ]; do
  e="$2"; [ "$e" = "-" ] && e=""
  env JTP_FAKE_COMM=1 $e python tools/rank_time.py 8 30 > $OUT/rank_$1.txt 2>&1
  echo "== $1: $(grep -h "^rank" $OUT/rank_$1.txt | sed "s/.*groups *\([0-9.]*\) us.*/\1/" | awk "{s+=\$1; n++} END {printf \"mean %.1f us over %d ranks\", s/n, n}")"
  shift 2
done
