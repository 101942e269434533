#!/usr/bin/env python3
"""The reference's bars test (examples/barstests/bars-learning.py + param-bars-*.py) on the MI355X path.

    python examples/bars_learning.py [bsc|mca|mmca|dsc|tsc|gsc] [--steps 50] [--N 2000] [--h5]

Generates bars data from ground-truth parameters, runs the annealed EM loop through the drop-in classes and
reports how well the learned dictionary matches the bars (mean absolute error after the best permutation).
Under `torchrun --nproc-per-node N` every rank takes its `stride_data` share and the statistics are
all-reduced once per EM step over RCCL.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from prosper_amd.em import EM                                  # noqa: E402
from prosper_amd.em.annealing import LinearAnnealing           # noqa: E402
from prosper_amd.utils import parallel                         # noqa: E402
from prosper_amd.utils.barstest import generate_bars_dict, find_permutation   # noqa: E402
from prosper_amd.utils.datalog import dlog, StoreInMemory      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="bsc", choices=["bsc", "mca", "mmca", "dsc", "tsc", "gsc"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--N", type=int, default=2000)
    ap.add_argument("--size", type=int, default=5)
    ap.add_argument("--h5", action="store_true", help="store every parameter and the objective per EM step in "
                    "output/<script>.<date>/result.h5, as the reference's bars-learning.py does")
    a = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    comm = parallel.Comm()
    np.random.seed(1 + comm.rank)

    size = a.size
    H, D = 2 * size, size ** 2
    Hprime, gamma = 5, 3
    bars = 10 * generate_bars_dict(H)
    if a.model == "bsc":
        from prosper_amd.em.camodels.bsc_et import BSC_ET as Model
        model = Model(D, H, Hprime, gamma, comm=comm)
        gt = {'W': bars, 'pi': 2. / H, 'sigma': 1.0}
    elif a.model == "mca":
        from prosper_amd.em.camodels.mca_et import MCA_ET as Model
        model = Model(D, H, Hprime, gamma, comm=comm)
        gt = {'W': bars, 'pi': 2. / H, 'sigma': 1.0}
    elif a.model == "mmca":
        from prosper_amd.em.camodels.mmca_et import MMCA_ET as Model
        model = Model(D, H, Hprime, gamma, comm=comm)
        gt = {'W': 10 * generate_bars_dict(H, neg_bars=True), 'pi': 2. / H, 'sigma': 1.0}
    elif a.model == "dsc":
        from prosper_amd.em.camodels.dsc_et import DSC_ET as Model
        model = Model(D, H, Hprime, gamma, states=np.array([-1., 0., 1.]), comm=comm)
        gt = {'W': bars, 'pi': np.array([1. / H, 1 - 2. / H, 1. / H]), 'sigma': 1.0}
    elif a.model == "tsc":
        from prosper_amd.em.camodels.tsc_et import TSC_ET as Model
        model = Model(D, H, Hprime, gamma, comm=comm)
        gt = {'W': bars, 'pi': 2. / H, 'sigma': 1.0}
    else:
        from prosper_amd.em.camodels.gsc_et import GSC as Model
        model = Model(D, H, Hprime, gamma, 'scalar', comm=comm)
        gt = {'W': bars / 10., 'pi': np.full(H, 2. / H), 'mu': np.full(H, 5.0), 'psi_sq': np.eye(H),
              'sigma_sq': 1.0}

    first, last = parallel.stride_data(a.N, comm=comm)
    my_data = model.generate_data(gt, last - first)
    init = model.standard_init(my_data)

    anneal = LinearAnnealing(a.steps)
    anneal['T'] = [(0, 2.), (.7, 1.)]
    anneal['Ncut_factor'] = [(0, 0.), (2. / 3, 1.)]
    anneal['anneal_prior'] = False
    log = dlog.set_handler(('L', 'Q'), StoreInMemory)
    store = None
    if a.h5:
        from prosper_amd.utils import create_output_path
        from prosper_amd.utils.datalog import StoreToH5
        out_dir = create_output_path("bars_learning_" + a.model, comm=comm)
        store = dlog.set_handler(('W', 'pi', 'sigma', 'sigma_sq', 'mu', 'psi_sq', 'L', 'Q', 'N_use', 'T'), StoreToH5,
                                 out_dir + "result.h5")
    em = EM(model=model, anneal=anneal, data={'y': my_data['y']}, lparams=init)
    em.run()
    if store is not None:
        store.close()
        if comm.rank == 0:
            print("parameters of every EM step in %sresult.h5" % out_dir)
    W = np.asarray(em.lparams['W'])
    W_gt = np.asarray(gt['W'])
    if a.model == "gsc":      # the scale of a column trades against the scale of its latent: compare shapes
        W, W_gt = W / np.abs(W).max(axis=0, keepdims=True) * 10, W_gt / np.abs(W_gt).max(axis=0, keepdims=True) * 10
    _, mae = find_permutation(np.abs(W) if a.model in ("mmca", "dsc", "tsc", "gsc") else W, np.abs(W_gt))
    if comm.rank == 0:
        trace = log.tables.get('L', log.tables.get('Q', [])) if log is not None else []
        print("%s on %d bars datapoints (%d ranks), %d EM steps: bars recovered with mean abs error %.3f%s" % (
            a.model.upper(), a.N, comm.size, a.steps, mae,
            "; objective %.3f -> %.3f" % (float(trace[0]), float(trace[-1])) if len(trace) else ""))


if __name__ == "__main__":
    main()
