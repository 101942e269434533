"""One EM iteration's kernel timeline from a rocprofv3 kernel trace: timeline.py <dir> <substring of the kernel that starts an iteration> [which]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
es = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(es) // 2
i0, i1 = es[k], es[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
print("iteration length %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
for r in rows[i0:i1 + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print("%8.1f %8.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), n))
