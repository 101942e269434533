#!/bin/bash
# usage: pmc_kernel2.sh <python script> <kernel name substring>  -- SQ counter sets for one kernel (averages per launch)
R=$PWD; S=$1; K=$2
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  PYTHONPATH=$R rocprofv3 --pmc $set --kernel-trace -d /tmp/pk_$i -o x --output-format csv -- python3 $R/$S > /tmp/pk_$i.log 2>&1 || tail -3 /tmp/pk_$i.log
done
python3 - "$K" <<'PY'
import glob, csv, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pk_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, v in sorted(agg.items()):
    v = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]      # drop warm-up outliers
    print("%-32s %16.0f  (launches %d)" % (n, sum(v) / len(v), len(v)))
PY
