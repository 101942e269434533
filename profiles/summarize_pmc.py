#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into profiles/<round>_pmc_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python profiles/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units and corrections follow MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled
for the GEMM kernels (16-B global loads / LDS-DMA).  Other access widths are "uncalibrated" per the guide,
so the row kernels were calibrated on known byte counts:
  * bsc_select_estep16 (8 B/lane, 128-B aligned row segments): select-only pass over a 196608 x 256 f64
    score matrix (402.65 MB) reads FETCH_SIZE = 400.0 MB raw (scratch/calib_rows.py) -> factor 1;
  * bsc_mstep_rows16 (8 B/lane, rows offset by one double): the 200000 x 416 f64 logpj + candidates + lse
    (673.6 MB) read FETCH_SIZE = 342.8 MB raw -> factor 2 (the 128-B-request behaviour of the guide).
WRITE_SIZE is exact for 16-B-per-lane stores and f64 atomics are counted as written bytes."""
import collections
import csv
import glob
import json
import sys

import re

# round 3: the 8-/16-wavefront kernels of bsc_fused8.hip, template arguments <STAGES, H', gamma, FULL, MSTATS, TAIL[, W16]>
_F8 = re.compile(r"bsc_estep_fused8s_kernel<\d+, \d+, \d+, (?:true|false), (true|false), (true|false)")


def _fused8_label(name):
    m = _F8.search(name)
    if not m:
        return None
    if m.group(2) == "true":
        return "estep_fused_tail_mstats" if m.group(1) == "true" else "estep_fused_tail"
    return "estep_fused_mstats" if m.group(1) == "true" else "estep_fused"


KERNELS = {
    # (order matters: first match wins)
    "bsc_estep_fused_kernel<16, 4, true, true>": "estep_fused_mstats",   # E-step + M-step statistics (inside EM steps)
    "bsc_estep_fused_kernel": "estep_fused",                              # scores GEMM + select + E-step in one launch
    "gemm_nt_f64_dma_kernel<false>": "scores_gemm",
    "gemm_nt_f64_dma_kernel<true>": "scores_gemm_splitk",
    "gemm_tn_f64": "stats_gemm",
    "bsc_wp_sparse_kernel": "stats_sparse",                              # Wp from the non-zero lists of E[s] (round 3)
    "bsc_select_estep16_kernel": "select_estep",
    "bsc_mstep_rows16_kernel": "mstep_rows",
}
FETCH_FACTOR = {"estep_fused": 2.0, "estep_fused_mstats": 2.0,    # reads are the LDS-DMA stream of Y (16 B per lane)
                "estep_fused_tail": 2.0, "estep_fused_tail_mstats": 2.0,
                "scores_gemm": 2.0, "scores_gemm_splitk": 2.0, "stats_gemm": 2.0, "select_estep": 1.0, "mstep_rows": 2.0,
                # 8 B/lane non-temporal loads of 512-byte row segments: calibrated on the bytes the kernel must read -- N D 8 =
                # 1.638 GB of Y at N = 200000 against a raw counter of 0.948 GB -> factor 2 (1.90 GB: Y + the lists' L2 misses)
                "stats_sparse": 2.0}


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        lab8 = _fused8_label(r["Kernel_Name"])
        if lab8:
            agg[(lab8, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
            continue
        for pat, lab in KERNELS.items():
            if pat in r["Kernel_Name"]:
                agg[(lab, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
                break
    return agg


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for (lab, grid), v in sorted(fetch.items()):
        w = write.get((lab, grid), [0.0])
        f_b = FETCH_FACTOR[lab] * 1024.0 * sum(v) / len(v)
        w_b = 1024.0 * sum(w) / len(w)
        res["%s@grid%d" % (lab, grid)] = {"launches": len(v), "fetch_factor": FETCH_FACTOR[lab], "fetch_bytes": f_b,
                                         "write_bytes": w_b, "hbm_bytes": f_b + w_b}
        if lab == "scores_gemm" and f_b + w_b > 1e9:   # the headline launch: all 200000 datapoints of the shard (whole
            res["%s@grid%d" % (lab, grid)]["datapoints_per_launch"] = 200000   # rounds + the fused ragged round)
        if lab.startswith("estep_fused_tail"):         # 512-thread workgroups of 16 datapoints (upper bound: ragged last one)
            res["%s@grid%d" % (lab, grid)]["datapoints_per_launch"] = grid // 512 * 16
        elif lab.startswith("estep_fused"):            # 1024-thread workgroups of 128 datapoints (round 3; round 2: 256 / 64)
            res["%s@grid%d" % (lab, grid)]["datapoints_per_launch"] = grid // 1024 * 128 if grid % 1024 == 0 and grid >= 1024 * 256 else grid // 256 * 64
    json.dump({"note": "per-launch HBM-side bytes (FETCH_SIZE x per-kernel gfx950 factor, see summarize_pmc.py, "
                       "+ WRITE_SIZE), rocprofv3 --pmc, separate passes; bench.py --steps 5 --warmup 2 --em-steps 3 "
                       "--prewarm-ms 0", "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
