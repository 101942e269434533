#!/bin/bash
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o x --output-format csv -- python3 $R/scratch/bench_gsc.py > /tmp/kt.log 2>&1 || tail -3 /tmp/kt.log
python3 - <<'EOF'
import glob, csv
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
EOF
