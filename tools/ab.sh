#!/bin/bash
# A/B of two builds of libjtprop.so on one box: tools/ab.sh other.so  (configs 4, 2, 3; alternating)
ALT=$1
for i in 1 2 3; do for lib in "" "$ALT"; do
  export JTPROP_LIB=${lib:-$PWD/junction-tree_amd/junctiontree_amd/lib/libjtprop.so}
  timeout -k 10 100 python3 bench.py --cpu-sample 0 --steps 100 > /tmp/ab.json 2>/dev/null || exit 1
  echo "C4 ${lib:+alt} : $(python3 tools/bsum.py /tmp/ab.json | tr '\n' ' ' | tr -s ' ')"
done; done
for lib in "" "$ALT"; do
  export JTPROP_LIB=${lib:-$PWD/junction-tree_amd/junctiontree_amd/lib/libjtprop.so}
  timeout -k 10 100 python3 bench.py --config c2 --cpu-sample 0 --steps 30 > /tmp/ab.json 2>/dev/null || exit 1
  echo "C2 ${lib:+alt} : $(python3 tools/bsum.py /tmp/ab.json | tr '\n' ' ' | tr -s ' ')"
  echo "C3 ${lib:+alt} : $(timeout -k 10 150 python3 tools/c3_time.py)"
done
