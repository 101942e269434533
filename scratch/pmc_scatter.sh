#!/bin/bash
# SQ / LDS / L2 counters of mca_defer_scatter_kernel (scratch/mca_T_sweep.py): what the 0.46 ms are spent on
R=/root/repo
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/ps_$i
  rocprofv3 --pmc $set --kernel-trace -d /tmp/ps_$i -o x --output-format csv -- python3 $R/scratch/mca_T_sweep.py > /tmp/ps_$i.log 2>&1 || tail -2 /tmp/ps_$i.log
done
python3 - <<'PY'
import glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/ps_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "mca_defer_scatter" in n or "mca_defer_q1" in n:
            agg[n.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for name, v in sorted(c.items()):
        print("   %-24s %16.0f  (%d launches)" % (name, sum(v) / len(v), len(v)))
PY
