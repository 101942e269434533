#!/bin/bash
# SQ instruction counters of the fused E-step kernel and of the two-kernel path's row kernel (config 2, one pass each)
R=$PWD
cd /tmp; export TMPDIR=/tmp
export ITERS=3
i=0
for set in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  for fused in 1 0; do
    PM_FUSED=$fused rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcf_${fused}_$i -o x --output-format csv -- python3 $R/scratch/fused_probe.py > /tmp/pmcf_${fused}_$i.log 2>&1 || tail -3 /tmp/pmcf_${fused}_$i.log
  done
done
python3 - <<'EOF2'
import glob, csv, collections
for f in sorted(glob.glob("/tmp/pmcf_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f)):
        for kern in ("bsc_estep_fused", "bsc_select_estep16", "gemm_nt_f64_dma_kernel<false>"):
            if kern in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
                a = agg[(kern, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in sorted(agg.items()):
        print("%-32s %-28s %16.0f  (%d launches)" % (k[0], k[1], v[0] / v[1], v[1]))
EOF2
