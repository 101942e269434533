#!/bin/bash
# round-6 evidence, one call: bench.py plain, under rocprofv3 --kernel-trace --stats, the FETCH_SIZE / WRITE_SIZE passes
# (profiles/summarize_pmc.py) and the SQ_INSTS_VALU / SQ_WAVES / SQ_BUSY_CYCLES counts of every model's row kernels.
# usage: bash scratch/profile_r06.sh [tag]      -> gpurun_out/<tag>_*; copy what is to be judged into profiles/
TAG=${1:-r06_v1}
R=$PWD
mkdir -p $R/gpurun_out
SHA=$(sha256sum $R/prosper_amd/libprosper_hip.so | cut -c1-16)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG /tmp/pmc_fetch /tmp/pmc_write /tmp/pv /tmp/pvd
if [ -z "$SKIP_PMC" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --em-steps 3 --prewarm-ms 0 --data device --no-cpu-baseline --no-other-models --no-other-shapes > /tmp/pmc_f.log 2>&1 || tail -3 /tmp/pmc_f.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --em-steps 3 --prewarm-ms 0 --data device --no-cpu-baseline --no-other-models --no-other-shapes > /tmp/pmc_w.log 2>&1 || tail -3 /tmp/pmc_w.log
python3 $R/profiles/summarize_pmc.py /tmp/pmc_fetch /tmp/pmc_write $R/gpurun_out/${TAG}_pmc_traffic.json > /dev/null
# VALU wave-instruction counts (+ waves, busy cycles) per launch of the row kernels of every model
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pv -o x --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --em-steps 2 --prewarm-ms 0 --no-cpu-baseline --no-other-shapes > /tmp/pv.log 2>&1 || tail -3 /tmp/pv.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pvd -o x --output-format csv -- python3 $R/scratch/bench_dsc.py > /tmp/pvd.log 2>&1 || tail -3 /tmp/pvd.log
python3 - $R/gpurun_out/${TAG}_valu_counts.json $SHA <<'PY'
import glob, csv, collections, json, sys, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d, tag in (("/tmp/pv", ""), ("/tmp/pvd", "dsc_only:")):
  for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|void |pm_fused8::", "", r["Kernel_Name"]).split("(")[0]
        if tag and not name.startswith("dsc_"):
            continue
        agg[(tag + name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for (name, grid), c in sorted(agg.items()):
    v = {k: sum(x) / len(x) for k, x in c.items()}
    if v.get("SQ_INSTS_VALU", 0) < 1e6:
        continue
    out["%s@grid%d" % (name, grid)] = {"launches": len(c["SQ_INSTS_VALU"]), "valu_insts": v.get("SQ_INSTS_VALU"),
                                        "mfma_insts": v.get("SQ_INSTS_MFMA"), "waves": v.get("SQ_WAVES"),
                                        "busy_cycles": v.get("SQ_BUSY_CYCLES")}
json.dump({"note": "wave-instructions per launch (SQ_INSTS_VALU includes MFMAs), SQ_WAVES and SQ_BUSY_CYCLES (summed over the 8 XCDs), "
                   "rocprofv3 --pmc, bench.py --steps 2 --warmup 1 --em-steps 2 --prewarm-ms 0 --no-cpu-baseline (+ scratch/bench_dsc.py "
                   "for the DSC-only keys)", "library_sha16": sys.argv[2], "kernels": out}, open(sys.argv[1], "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
fi
# the counts of THIS build are what the bench lines below quote (roofline.traffic, the *_roofline objects of other_models)
if [ -z "$SKIP_PMC" ]; then
cp $R/gpurun_out/${TAG}_pmc_traffic.json $R/profiles/r06_pmc_traffic.json
cp $R/gpurun_out/${TAG}_valu_counts.json $R/profiles/r06_valu_counts.json
fi
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $R/gpurun_out/${TAG}_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_$TAG.log 2>/dev/null
tail -1 /tmp/prof_$TAG.log > $R/gpurun_out/${TAG}_bench_under_rocprof.json
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
# steady-state view of the dominant kernel: its launches from the kernel trace, without the first 40 % (clock ramp)
python3 - /tmp/prof_$TAG $R/gpurun_out/${TAG}_dominant_steady.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bsc_estep_fused8s_kernel<4, 8, 4, true, false, false>" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
d = [x[1] / 1e6 for x in rows]
tail = d[int(len(d) * 0.4):]
open(sys.argv[2], "w").write(
    "bsc_estep_fused8s_kernel<4, 8, 4, true, false, false> (the E-step pass's main launch, 196608 datapoints), rocprofv3 --kernel-trace of\n"
    "`python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline`: %d launches, average %.4f ms, minimum %.4f ms;\n"
    "steady state (the last 60 %% of the launches, past the clock ramp of the first passes): average %.4f ms = %.1f TFLOP/s = %.3f of 78.6\n"
    % (len(d), sum(d) / len(d), min(d), sum(tail) / len(tail), 2 * 196608 * 1024 * 256 / (sum(tail) / len(tail) * 1e-3) / 1e12,
       2 * 196608 * 1024 * 256 / (sum(tail) / len(tail) * 1e-3) / 1e12 / 78.6))
PY
head -c 600 $R/gpurun_out/${TAG}_bench.json; echo; cat $R/gpurun_out/${TAG}_dominant_steady.txt
