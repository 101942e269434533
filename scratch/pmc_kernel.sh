#!/bin/bash
# SQ counters of one kernel: scratch/pmc_kernel.sh <python script> <kernel name substring>
R=$PWD
SCRIPT=$1; KERN=$2
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmck_$i -o x --output-format csv -- python3 $R/$SCRIPT > /tmp/pmck_$i.log 2>&1 || tail -3 /tmp/pmck_$i.log
done
python3 - "$KERN" <<'EOF'
import glob, csv, collections, sys
kern = sys.argv[1]
for f in sorted(glob.glob("/tmp/pmck_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in agg.items():
        print(k, v[0] / v[1], v[1])
EOF
