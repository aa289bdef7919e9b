#!/bin/bash
# Instruction-cache counters of one bench command:   bash tools/pmc_icache.sh OUTDIR "bench args"
export TMPDIR=/tmp
OUT=$PWD/$1; ARGS=$2
rm -rf $OUT; mkdir -p $OUT
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  n=$(echo $c | tr " " "_" | cut -c1-60)
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/$n -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-profile $ARGS > /dev/null 2> $OUT/$n.err
done
python3 - <<PY
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join("$OUT", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        k = r["Kernel_Name"].split("(")[0]
        if "jt_" in k and "flow" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join("$OUT", "summary.txt"), "w") as fh:
    for k, cs in sorted(acc.items()):
        print(k, file=fh)
        for c, v in sorted(cs.items()):
            print("   %-26s mean %18.1f  (n=%d)" % (c, sum(v) / len(v), len(v)), file=fh)
print(open(os.path.join("$OUT", "summary.txt")).read())
PY
for f in $OUT/*.err; do tail -n 2 $f; done | head -30
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
