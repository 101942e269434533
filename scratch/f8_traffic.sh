#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the fused kernel in the harness: scratch/f8_traffic.sh <variant> ...
R=$PWD
cd /tmp; export TMPDIR=/tmp
for name in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/f8t
    rocprofv3 --pmc $c --kernel-trace -d /tmp/f8t -o x --output-format csv -- $R/scratch/libs/f8_bench $R/scratch/libs/libpm_$name.so 8 196608 3 > /tmp/f8t.log 2>&1
    python3 - "$name" $c <<'PY'
import glob, csv, sys
tot=0; n=0
for f in glob.glob("/tmp/f8t/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bsc_estep_fused8s" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print(sys.argv[1], sys.argv[2], "%.1f MB per launch (raw KiB counter x 1024%s)" % (tot/max(n,1)*1024/1e6*(2 if sys.argv[2]=="FETCH_SIZE" else 1), ", x2 gfx950 correction" if sys.argv[2]=="FETCH_SIZE" else ""))
PY
  done
done
