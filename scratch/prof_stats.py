"""Print the kernel-stats CSV of a rocprofv3 --kernel-trace --stats run: name, calls, average / total duration."""
import csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
files = glob.glob(pat + "/**/*kernel_stats.csv", recursive=True)
if not files:
    print("no kernel_stats.csv under", pat, glob.glob(pat + "/**", recursive=True)[:20])
    sys.exit(0)
rows = list(csv.DictReader(open(files[0])))
for r in rows[:top]:
    print("%-100s %6s %12.1f us  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r.get("Percentage", "")[:5]))
