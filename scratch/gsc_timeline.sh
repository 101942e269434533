#!/bin/bash
# device timeline of one steady EM iteration of GSC config 4
R=$PWD
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tlg
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tlg -o tl -- python3 $R/scratch/gsc_em_time.py > /tmp/tlg.log 2>&1 || tail -3 /tmp/tlg.log
python3 - <<'EOF'
import csv, glob
ev = []
for f in glob.glob("/tmp/tlg/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
ev.sort()
idx = [i for i, e in enumerate(ev) if "gsc_estep_kernel" in e[2]]
lo, hi = idx[-12], idx[-11]
t0 = ev[lo][0]; prev = None; small = 0; gaps = 0.0
for s, e, n in ev[lo:hi]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%8.1f +%7.1f gap %5.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev = max(prev or e, e)
print("period %.3f ms" % ((ev[hi][0] - ev[lo][0]) / 1e6))
EOF
