#!/bin/bash
# Diagnostic: issue / LDS counters of the message-passing kernels on config 3 (a few counters per pass, no trace domain)
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_c3; rm -rf $O; mkdir -p $O
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM" "VALUBusy" "MemUnitStalled"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$n -o p -- python3 tools/c3_time.py > $O/$n.out 2> $O/$n.err
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.getcwd(), "gpurun_out", "pmc_c3")
acc = collections.defaultdict(list)
for path in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        if "jt_" in r.get("Kernel_Name", ""):
            acc[(r["Kernel_Name"].split("(")[0].split("<")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(O + "/summary.txt", "w") as fh:
    for k in sorted(acc):
        v = acc[k]
        fh.write("%-28s %-24s mean %.4g over %d launches\n" % (k[0], k[1], sum(v) / len(v), len(v)))
print(open(O + "/summary.txt").read())
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
