#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVES / SQ_BUSY_CYCLES of the MCA statistics pass with the log/exp-free power of rho = 21
# (pm_pow_m20_21, shipped) and with the table-driven general power (-DPM_MCA_NO_ROOT21)
R=$PWD
cp prosper_amd/libprosper_hip.so /tmp/lib.keep
for f in "" "-DPM_MCA_NO_ROOT21"; do
  touch prosper_amd/csrc/mca_kernels.hip
  PM_EXTRA_FLAGS="$f" bash prosper_amd/csrc/build.sh > /dev/null 2>&1
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/mp
  PYTHONPATH=$R rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d /tmp/mp -o x --output-format csv -- python3 $R/scratch/mca_em_time.py > /tmp/mp.log 2>&1 || tail -3 /tmp/mp.log
  cd $R
  python3 - "$f" <<'PY'
import glob, csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/mp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mca_estep_fused" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print("flags '%s'" % sys.argv[1], k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(c["SQ_WAVES"]))
PY
done
cp /tmp/lib.keep prosper_amd/libprosper_hip.so
touch prosper_amd/csrc/mca_kernels.hip
