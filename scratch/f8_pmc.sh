#!/bin/bash
# SQ / SQC counters of the fused E-step kernel in the stand-alone harness: scratch/f8_pmc.sh <variant>:<tile> ...
R=$PWD
OUT=$R/gpurun_out/f8_pmc.log
: > $OUT
cd /tmp; export TMPDIR=/tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES"
      "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH"
      "SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
      "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES")
for v in "$@"; do
  name=${v%%:*}; tile=${v#*:}
  rm -rf /tmp/f8pmc_*
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d /tmp/f8pmc_$i -o x --output-format csv -- $R/scratch/libs/f8_bench $R/scratch/libs/libpm_$name.so $tile 196608 3 > /tmp/f8pmc_$i.log 2>&1 || tail -3 /tmp/f8pmc_$i.log >> $OUT
  done
  echo "== $name tile $tile" >> $OUT
  python3 - >> $OUT <<'PY'
import glob, csv, collections
tot = {}
for f in sorted(glob.glob("/tmp/f8pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f)):
        if "bsc_estep_fused" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in agg.items():
        tot[k] = v[0] / v[1]
w = tot.get("SQ_WAVES", 1.0)
for k in sorted(tot):
    print("%-28s %16.0f  per wave %12.1f" % (k, tot[k], tot[k] / w))
PY
done
cat $OUT
