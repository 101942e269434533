"""Soak: repeated annealed EM runs (all models) and a long deterministic GSC loop -- device memory must not grow, results stay finite."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
from schedule_inputs import schedule_inputs, DSC_STATES
from prosper_amd.em import EM
from prosper_amd.em.annealing import LinearAnnealing
def sched(steps=50):
    an = LinearAnnealing(steps); an["T"] = [(0, 2.), (.7, 1.)]; an["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]; an["anneal_prior"] = False
    return an
def mk(kind, D, H, Hp, g):
    if kind == "bsc":
        from prosper_amd.em.camodels.bsc_et import BSC_ET; return BSC_ET(D, H, Hp, g)
    if kind == "mca":
        from prosper_amd.em.camodels.mca_et import MCA_ET; return MCA_ET(D, H, Hp, g)
    if kind == "mmca":
        from prosper_amd.em.camodels.mmca_et import MMCA_ET; return MMCA_ET(D, H, Hp, g)
    if kind == "dsc":
        from prosper_amd.em.camodels.dsc_et import DSC_ET; return DSC_ET(D, H, Hp, g, states=DSC_STATES.copy())
    if kind == "tsc":
        from prosper_amd.em.camodels.tsc_et import TSC_ET; return TSC_ET(D, H, Hp, g)
    from prosper_amd.em.camodels.gsc_et import GSC; return GSC(D, H, Hp, g, sigma_sq_type="scalar")
for kind, shape in (("bsc", (256, 160, 8, 3, 20000)), ("gsc", (256, 128, 6, 3, 20000)), ("mca", (128, 64, 6, 3, 8000)),
                    ("mmca", (128, 64, 6, 3, 8000)), ("dsc", (128, 64, 6, 3, 12000)), ("tsc", (128, 64, 6, 3, 12000))):
    D, H, Hp, g, N = shape
    y, p0 = schedule_inputs(kind, D, H, N, 900)
    yd = torch.from_numpy(y).cuda()
    mem = []
    for det in (False, True):
        m = mk(kind, D, H, Hp, g); m.deterministic = det
        for rep in range(6):
            em = EM(model=m, anneal=sched(), data={"y": yd}, lparams={k: np.array(v, copy=True) for k, v in p0.items()})
            em.run()
            torch.cuda.synchronize(); gc.collect()
            mem.append(torch.cuda.memory_allocated() / 2**20)
            assert all(np.isfinite(np.asarray(v, dtype=np.float64)).all() for k, v in em.lparams.items() if k != "Q"), (kind, det, rep)
        del m
    print("%-5s 2 x 6 runs of 50 annealed steps: device memory after each run (MiB) %s" % (kind, ["%.0f" % x for x in mem]), flush=True)
    del yd
    torch.cuda.empty_cache()
