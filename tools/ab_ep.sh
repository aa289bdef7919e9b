O=gpurun_out/ab_ep.txt; : > $O
run() { echo "== $*" >> $O; for a in "3 13 6 63 f32" "5 9 4 63 f32" "6 8 4 63 f32"; do env "$@" timeout -k 10 120 python3 tools/odd_time.py $a 2>&1 | grep "mixed" >> $O; done; }
run JTP_FORCE_LEVEL_LAUNCHES=1
run JTP_DEBUG=1
run JTP_FORCE_LEVEL_LAUNCHES=1 JTP_NO_VGROUPS=1
run JTP_DEBUG=1 JTP_NO_VGROUPS=1
cat $O
