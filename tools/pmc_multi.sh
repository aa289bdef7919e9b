#!/bin/bash
# Counter passes over one bench command (one or a few counters per pass; never combined with a trace domain).
#   bash tools/pmc_multi.sh OUTDIR "bench args"          (LDS_ONLY=1: only the LDS counters)
export TMPDIR=/tmp
OUT=$PWD/$1; ARGS=$2
rm -rf $OUT; mkdir -p $OUT
[ -n "$LDS_ONLY" ] && set -- "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" || set -- "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "LdsBankConflict" "VALUBusy" "MemUnitStalled" "FETCH_SIZE" "WRITE_SIZE"
for c in "$@"; do
  n=$(echo $c | tr " " "_" | cut -c1-60)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$n -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-profile $ARGS > /dev/null 2> $OUT/$n.err
done
python3 - <<PY
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join("$OUT", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        k = r["Kernel_Name"].split("(")[0]
        if "jt_" in k and ("flow" in k):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join("$OUT", "summary.txt"), "w") as fh:
    for k, cs in sorted(acc.items()):
        print(k, file=fh)
        for c, v in sorted(cs.items()):
            print("   %-26s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)), file=fh)
print(open(os.path.join("$OUT", "summary.txt")).read())
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
