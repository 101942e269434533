#!/bin/bash
# Dynamic VALU instruction mix of one kernel: scratch/pmc_valu_mix.sh <python script> <kernel name substring>
R=$PWD
SCRIPT=$1; KERN=$2
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_IOPS" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_VSKIPPED" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F32 SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcv_$i -o x --output-format csv -- python3 $R/$SCRIPT > /tmp/pmcv_$i.log 2>&1 || tail -3 /tmp/pmcv_$i.log
done
python3 - "$KERN" <<'PY'
import glob, csv, collections, sys
kern = sys.argv[1]
tot = {}
for f in sorted(glob.glob("/tmp/pmcv_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in agg.items():
        tot[k] = v[0] / v[1]
w = tot.get("SQ_WAVES", 1.0)
for k in sorted(tot):
    print("%-28s %14.0f  per wave %10.1f" % (k, tot[k], tot[k] / w))
PY
