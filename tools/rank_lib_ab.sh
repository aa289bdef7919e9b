OUT=$1; shift; mkdir -p $OUT
L=junction-tree_amd/junctiontree_amd/lib
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = cur ]; then env -u JTPROP_LIB JTP_FAKE_COMM=1 python tools/rank_time.py 8 30 > $OUT/rank_${n}_$rep.txt 2>&1
  else JTPROP_LIB=$L/libjtprop_$n.so JTP_FAKE_COMM=1 python tools/rank_time.py 8 30 > $OUT/rank_${n}_$rep.txt 2>&1; fi
  echo "== $n $rep: $(grep -h "^rank" $OUT/rank_${n}_$rep.txt | sed "s/.*groups *\([0-9.]*\) us.*/\1/" | awk "{s+=\$1; n++} END {printf \"mean %.1f us over %d ranks\", s/n, n}")"
done
done
