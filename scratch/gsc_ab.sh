#!/bin/bash
# A/B of the GSC E-step kernel (variant library in scratch/libA) + SQ counters of the shipped one
R=$PWD
cp prosper_amd/libprosper_hip.so /tmp/base.so
if [ -f scratch/libA/libprosper_hip.so ]; then
  cp scratch/libA/libprosper_hip.so prosper_amd/libprosper_hip.so
  echo variantA; python scratch/bench_gsc.py 2>&1 | grep -E "estep|Error|error" | head -3
  cp /tmp/base.so prosper_amd/libprosper_hip.so
fi
echo base; python scratch/bench_gsc.py 2>&1 | grep -E "estep|Error|error" | head -3
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$i -o x --output-format csv -- python3 $R/scratch/bench_gsc.py > /tmp/pmc_$i.log 2>&1 || tail -3 /tmp/pmc_$i.log
done
python3 $R/scratch/gsc_ab_sum.py
