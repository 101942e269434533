"""ORACLE leg of bench.py: time the per-datapoint reference algorithm on the host cores.

Test/measurement infrastructure, not product code.  Runs the faithful restatement
(oracle/bsc_oracle.py *_loop functions == prosper/em/camodels/bsc_et.py:98-115, 119-192,
195-438 including their redundant per-datapoint work) on a BOUNDED sample of the bench
workload: every worker process (one per host core, OPENBLAS_NUM_THREADS=1 -- the stand-in
for `mpirun -np P`) draws its own rows of the config-2 generator and processes rows until its
time budget is spent.  Prints one JSON object.

    python oracle/cpu_baseline.py --D 1024 --H 256 --Hprime 8 --gamma 4 --budget 8
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bsc_oracle as O  # noqa: E402


def _worker(args):
    rank, D, H, Hp, gamma, budget, full_budget, chunk = args
    rng = np.random.RandomState(0)
    W_gt = rng.normal(size=(D, H))                       # same ground truth on every worker
    rng = np.random.RandomState(1000 + rank)
    pi_gt = 4.0 / H
    model = O.make_model(D, H, Hp, gamma)
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": pi_gt, "sigma": 1.0}
    an = O.Anneal(T=1.0)
    mu = np.zeros(D)
    done, t_e = 0, 0.0
    last = None
    while t_e < budget:
        y, _ = O.generate_bsc_data(W_gt, pi_gt, 1.0, chunk, rng)
        t0 = time.perf_counter()
        cand = O.select_hprimes_loop(params["W"], y, Hp)
        logpj = O.e_step_loop(an, params["W"], params["pi"], params["sigma"], mu, y, cand,
                              model["SM"], model["state_abs"])
        t_e += time.perf_counter() - t0
        done += chunk
        last = (y, cand, logpj)
    # M-step statistics loops on the last chunk(s), separately budgeted
    m_done, t_m = 0, 0.0
    while t_m < full_budget and last is not None:
        y, cand, logpj = last
        t0 = time.perf_counter()
        O.m_step_stats_loop(params["W"], mu, y, cand, logpj, model["SM"])
        t_m += time.perf_counter() - t0
        m_done += chunk
    return done, t_e, m_done, t_m


def _vectorised(D, H, Hp, gamma, budget, rows=4096):
    """The "fair CPU" figure: the oracle's GEMM + Gram-matrix E-step (select_hprimes_vec +
    e_step_vec) on chunks of `rows` datapoints with multi-threaded BLAS, in THIS process."""
    rng = np.random.RandomState(0)
    W_gt = rng.normal(size=(D, H))
    rng = np.random.RandomState(2000)
    pi_gt = 4.0 / H
    model = O.make_model(D, H, Hp, gamma)
    W = W_gt + 0.1 * rng.normal(size=(D, H))
    an = O.Anneal(T=1.0)
    mu = np.zeros(D)
    y, _ = O.generate_bsc_data(W_gt, pi_gt, 1.0, rows, rng)
    O.select_hprimes_vec(W, y[:64], Hp)            # first-call overheads (thread pool) out of the clock
    done, t = 0, 0.0
    while t < budget:
        t0 = time.perf_counter()
        cand = O.select_hprimes_vec(W, y, Hp)
        O.e_step_vec(an, W, pi_gt, 1.0, mu, y, cand, model["SM"], model["state_abs"])
        t += time.perf_counter() - t0
        done += rows
    return done / t, done


def parity_case(path, D, H, Hp, gamma, N=1536, steps=3):
    """Reference answer for bench.py's parity report: `steps` EM steps of the (vectorised, golden-pinned)
    oracle on a seeded sample at the bench's dimensions.  Written as plain arrays to `path`; bench.py runs
    the same inputs through the HIP path and reports the differences."""
    rng = np.random.RandomState(7)
    W_gt = rng.normal(size=(D, H))
    y, _ = O.generate_bsc_data(W_gt, 4.0 / H, 1.0, N, rng)
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
    model = O.make_model(D, H, Hp, gamma)
    an = O.Anneal(T=1.0)
    out = {"y": y, "W0": params["W"], "pi0": params["pi"], "sigma0": params["sigma"], "steps": steps}
    p = dict(params)
    for k in range(steps):
        p, log = O.em_step(an, model, p, y, stats_fn=O.m_step_stats_vec, vec=True)
        p["mu"] = np.zeros(D)
        if k == 0:
            out["candidates1"] = log["candidates"]
        out["L%d" % (k + 1)] = log["L"]
    out.update(W=p["W"], pi=p["pi"], sigma=p["sigma"])
    np.savez(path, **out)


def _physical_cores():
    """Worker processes = PHYSICAL cores this process may run on (one BLAS-free NumPy loop per core): the logical count
    (SMT siblings) oversubscribes the cores' memory system and makes the baseline a straw man."""
    logical = len(os.sched_getaffinity(0))
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or logical
        total = psutil.cpu_count(logical=True) or logical
        # the affinity mask may cover a part of the machine: scale the physical count with it
        phys = max(1, min(logical, int(round(phys * logical / float(total)))))
        return phys
    except Exception:
        return logical


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--D", type=int, default=1024)
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--Hprime", type=int, default=8)
    ap.add_argument("--gamma", type=int, default=4)
    ap.add_argument("--budget", type=float, default=8.0, help="seconds of E-step work per core")
    ap.add_argument("--full-budget", type=float, default=6.0, help="seconds of M-step work per core")
    ap.add_argument("--chunk", type=int, default=32)
    ap.add_argument("--cores", type=int, default=0)
    ap.add_argument("--parity-out", default="", help="also write the oracle's answer for bench.py's parity report")
    ap.add_argument("--solo-budget", type=float, default=3.0, help="seconds of E-step work of ONE uncontended process")
    ap.add_argument("--vec-budget", type=float, default=3.0, help="seconds of the vectorised multi-threaded leg")
    ap.add_argument("--vec-only", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.vec_only:      # child of this script, started with a multi-threaded BLAS environment
        rate, rows = _vectorised(a.D, a.H, a.Hprime, a.gamma, a.vec_budget)
        print(json.dumps({"value": rate, "rows": rows}))
        return
    if a.parity_out:
        parity_case(a.parity_out, a.D, a.H, a.Hprime, a.gamma)
    cores = a.cores or _physical_cores()
    # one process alone on the box: the per-core rate without the memory-system contention of `cores` workers
    solo = None
    if a.solo_budget > 0:
        d, t, _, _ = _worker((997, a.D, a.H, a.Hprime, a.gamma, a.solo_budget, 0.0, a.chunk))
        solo = d / t
    vec = None
    if a.vec_budget > 0:
        import subprocess
        threads = min(cores, 64)
        env = dict(os.environ, OPENBLAS_NUM_THREADS=str(threads), OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--vec-only", "--D", str(a.D), "--H", str(a.H),
                                  "--Hprime", str(a.Hprime), "--gamma", str(a.gamma), "--vec-budget", str(a.vec_budget)],
                                 env=env, check=True, capture_output=True, text=True, timeout=120).stdout.strip().splitlines()
            v = json.loads(out[-1])
            vec = {"value": v["value"], "unit": "datapoints/s", "blas_threads": threads, "rows": v["rows"],
                   "what": "oracle select_hprimes_vec + e_step_vec (one GEMM + Gram-matrix algebra, NumPy/OpenBLAS), "
                           "chunks of 4096 datapoints"}
        except Exception as e:   # the faithful figure is still worth reporting
            vec = {"value": None, "error": repr(e)}
    ctx = mp.get_context("spawn")
    jobs = [(r, a.D, a.H, a.Hprime, a.gamma, a.budget, a.full_budget, a.chunk) for r in range(cores)]
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_worker, jobs)
    wall = time.perf_counter() - t0
    rows = sum(r[0] for r in res)
    t_e = max(r[1] for r in res)
    e_rate = rows / t_e                                    # all cores together, slowest worker's clock
    per_core_e = np.mean([r[0] / r[1] for r in res])
    per_core_m = np.mean([r[2] / r[3] for r in res if r[3] > 0]) if any(r[3] > 0 for r in res) else float("nan")
    full = 1.0 / (1.0 / per_core_e + 1.0 / per_core_m) * cores if per_core_m == per_core_m else None
    print(json.dumps({
        "value": e_rate, "unit": "datapoints/s", "cores": cores, "kind": "port",
        "sample": "%d rows of the config-2 generator (D=%d H=%d H'=%d gamma=%d), %d rows/chunk, %.0f s E-step budget "
                  "per core; faithful per-datapoint NumPy loops, one process per PHYSICAL core, 1 BLAS thread each" % (
                      rows, a.D, a.H, a.Hprime, a.gamma, a.chunk, a.budget),
        "per_core_estep": per_core_e, "per_core_mstep": per_core_m, "full_step_value": full,
        "uncontended_per_core": solo, "vectorised": vec,
        "wall_s": wall,
    }))


if __name__ == "__main__":
    main()
