#!/bin/bash
# device timeline (kernels + copies, gaps) of the LAST steady EM iteration of a script:  bash scratch/timeline.sh <script.py> <marker kernel substring>
R=$PWD; S=$1; K=$2
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tlx
PYTHONPATH=$R rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tlx -o tl -- python3 $R/$S > /tmp/tlx.log 2>&1 || tail -3 /tmp/tlx.log
python3 - "$K" <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob("/tmp/tlx/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:56]))
for f in glob.glob("/tmp/tlx/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
idx = [i for i, e in enumerate(ev) if sys.argv[1] in e[2]]
lo, hi = idx[-2], idx[-1]
t0 = ev[lo][0]; prev = None; busy = 0.0
for s, e, n in ev[lo:hi]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%8.1f +%7.1f gap %6.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    busy += (e - max(s, prev or s)) / 1e3 if (prev is None or e > prev) else 0.0
    prev = max(prev or e, e)
print("period %.3f ms, device busy %.3f ms" % ((ev[hi][0] - ev[lo][0]) / 1e6, busy / 1e3))
PY
