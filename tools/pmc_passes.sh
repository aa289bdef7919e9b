#!/bin/bash
# Derived-counter passes over the C4 bench (one or two counters per pass).  Usage (GPU box): bash tools/pmc_passes.sh
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
for c in "LdsBankConflict" "LdsUtil" "VALUBusy" "SALUBusy" "MemUnitStalled" "MemUnitBusy" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQC_ICACHE_MISSES SQC_ICACHE_REQ" "MeanOccupancyPerCU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "L2CacheHit"; do
  n=$(echo $c | tr " " "_")
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$n -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-profile > /dev/null 2> $OUT/$n.err
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ.get("PWD", "."), "gpurun_out", "pmc")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        k = r["Kernel_Name"].split("(")[0]
        if "flow" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s mean %14.4f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
