#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into profiles/<round>_pmc_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python profiles/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units and corrections follow MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming read -- the access
pattern of every kernel listed here -- so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores
and f64 atomics are counted as written bytes."""
import collections
import csv
import glob
import json
import sys

KERNELS = {
    "gemm_nt_f64_dma_kernel<false>": "scores_gemm",
    "gemm_nt_f64_dma_kernel<true>": "scores_gemm_splitk",
    "gemm_tn_f64_kernel": "stats_gemm",
    "bsc_select_estep16_kernel": "select_estep",
    "bsc_mstep_rows16_kernel": "mstep_rows",
}


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for pat, lab in KERNELS.items():
            if pat in r["Kernel_Name"]:
                agg[(lab, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for (lab, grid), v in sorted(fetch.items()):
        w = write.get((lab, grid), [0.0])
        f_b = 2.0 * 1024.0 * sum(v) / len(v)
        w_b = 1024.0 * sum(w) / len(w)
        res["%s@grid%d" % (lab, grid)] = {"launches": len(v), "fetch_bytes": f_b, "write_bytes": w_b,
                                         "hbm_bytes": f_b + w_b}
    json.dump({"note": "per-launch HBM-side bytes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), rocprofv3 --pmc, "
                       "separate passes", "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
