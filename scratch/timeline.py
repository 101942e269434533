"""Print the device timeline (kernels + copies) of the last EM iteration from a rocprofv3 csv directory."""
import csv, glob, sys
root = sys.argv[1]
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# iterations are delimited by the scores GEMM (largest grid dma kernel); take the last complete one
idx = [i for i, e in enumerate(ev) if ("bsc_estep_fused" in e[2] or "gemm_nt_f64_dma_kernel" in e[2]) and (e[1] - e[0]) > 1_000_000]
lo, hi = idx[-2], idx[-1]
# walk back to include the uploads preceding the GEMM
t0 = ev[lo][0]
prev_end = None
for s, e, n in ev[lo - 12:hi]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%9.1f us  +%8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev_end = max(prev_end or e, e)
print("iteration period: %.3f ms" % ((ev[hi][0] - ev[lo][0]) / 1e6))
