#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVES / SQ_INSTS_VALU_MFMA-free counts per launch of every model's row kernels (bench.py's side models):
# profiles/r03_valu_counts.json feeds the VALU-issue rooflines bench.py reports for them.
R=$PWD
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES --kernel-trace -d /tmp/pv -o x --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --em-steps 2 --prewarm-ms 0 --no-cpu-baseline > /tmp/pv.log 2>&1 || tail -3 /tmp/pv.log
# DSC and TSC run the same kernels at the same grid: a DSC-only pass tells them apart (keys "dsc_only:...")
rm -rf /tmp/pvd
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES --kernel-trace -d /tmp/pvd -o x --output-format csv -- python3 $R/scratch/bench_dsc.py > /tmp/pvd.log 2>&1 || tail -3 /tmp/pvd.log
python3 - $R/gpurun_out/r03_valu_counts.json <<'PY'
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
                                        "mfma_insts": v.get("SQ_INSTS_MFMA"), "waves": v.get("SQ_WAVES")}
json.dump({"note": "wave-instructions per launch (SQ_INSTS_VALU includes MFMAs), rocprofv3 --pmc, bench.py --steps 2 "
                   "--warmup 1 --em-steps 2 --prewarm-ms 0 --no-cpu-baseline", "kernels": out}, open(sys.argv[1], "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
