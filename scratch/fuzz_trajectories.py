"""Exploratory: short EM trajectories (6 steps of a moving schedule: T ramp + data truncation) at random shapes, HIP loop with all
pipeline features on against the oracle loop -- speculation, deferred statistics, warm inverses, lists at shapes no test names."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.annealing import LinearAnnealing
class _OA(dict):
    def __missing__(s, k): return 0.0
def sched(steps):
    an = LinearAnnealing(steps); an["T"] = [(0, 1.6), (.7, 1.)]; an["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]; an["anneal_prior"] = False
    return an
if os.environ.get("FUZZ_DET") == "1":          # every model in deterministic mode (libprosper_hip_det.so)
    from prosper_amd.em.camodels import _device
    _orig_init = _device.DeviceCAModel.__init__
    def _init(self, *a, **k):
        _orig_init(self, *a, **k)
        self.deterministic = True
    _device.DeviceCAModel.__init__ = _init
rng = np.random.RandomState(21)
fails = 0
STEPS = 6
def close(tag, k, g, r, tol):
    global fails
    g, r = np.asarray(g, dtype=np.float64), np.asarray(r, dtype=np.float64)
    if not np.allclose(g, r, rtol=tol, atol=tol * max(1.0, float(np.abs(r).max()))):
        print("DEVIATION:", tag, k, float(np.abs(g - r).max() / max(1.0, np.abs(r).max())), flush=True); fails += 1
for trial in range(int(os.environ.get("FUZZ_TRIALS", "40"))):
    kind = ["bsc", "gsc", "dsc", "tsc", "mca"][trial % 5]
    H = int(rng.randint(3, 60)); Hp = int(rng.randint(2, min(H, 8) + 1)); gamma = int(rng.randint(1, min(Hp, 4) + 1))
    D = int(rng.randint(8, 150)); N = int(rng.randint(20 * H, 40 * H))
    tag = "%s D=%d H=%d H'=%d g=%d N=%d" % (kind, D, H, Hp, gamma, N)
    try:
        if kind == "bsc":
            from oracle import bsc_oracle as O
            from prosper_amd.em.camodels.bsc_et import BSC_ET
            W = rng.normal(size=(D, H)) * 2; y = (rng.random_sample((N, H)) < 2.0 / H) @ W.T + rng.normal(size=(N, D))
            p = {"W": W + 0.2 * rng.normal(size=(D, H)), "pi": 2.5 / H, "sigma": 1.2, "mu": np.zeros(D)}
            m, om = BSC_ET(D, H, Hp, gamma), O.make_model(D, H, Hp, gamma)
            ostep = lambda a, q: O.em_step(O.Anneal(T=a["T"], Ncut_factor=a["Ncut_factor"], anneal_prior=False), om, q, y, stats_fn=O.m_step_stats_vec, vec=True)[0]
            keys, tol = ("W", "pi", "sigma"), 1e-6
        elif kind == "gsc":
            from oracle import gsc_oracle as O
            from prosper_amd.em.camodels.gsc_et import GSC
            W = rng.normal(size=(D, H)); y = ((rng.random_sample((N, H)) < 2.0 / H) * (1.5 + rng.normal(size=(N, H)))) @ W.T + rng.normal(size=(N, D))
            p = {"W": W + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.2 / H), "mu": 1.4 + 0.1 * rng.normal(size=H), "psi_sq": np.diag(rng.uniform(0.8, 1.3, size=H)), "sigma_sq": 1.2}
            m, om = GSC(D, H, Hp, gamma, 'scalar'), O.make_model(D, H, Hp, gamma)
            ostep = lambda a, q: O.em_step(O.Anneal(T=a["T"]), om, q, y)[0]
            keys, tol = ("W", "pi", "mu", "psi_sq", "sigma_sq"), 1e-5
        elif kind in ("dsc", "tsc"):
            states = np.array([-1., 0., 1.])
            W = rng.normal(size=(D, H)) * 2; pig = np.array([1.0 / H, 1 - 2.0 / H, 1.0 / H])
            y = rng.choice(states, size=(N, H), p=pig) @ W.T + rng.normal(size=(N, D))
            if kind == "dsc":
                from oracle import dsc_oracle as O
                from prosper_amd.em.camodels.dsc_et import DSC_ET
                m, om, pi = DSC_ET(D, H, Hp, gamma, states=states), O.make_model(D, H, Hp, gamma, states), pig * np.array([1.2, 1.0, 0.8]) / (pig * np.array([1.2, 1.0, 0.8])).sum()
            else:
                from oracle import tsc_oracle as O
                from prosper_amd.em.camodels.tsc_et import TSC_ET
                m, om, pi = TSC_ET(D, H, Hp, gamma), O.make_model(D, H, Hp, gamma), 2.4 / H
            p = {"W": W + 0.2 * rng.normal(size=(D, H)), "pi": pi, "sigma": 1.2}
            ostep = lambda a, q: O.em_step(O.Anneal(T=a["T"], Ncut_factor=a["Ncut_factor"], anneal_prior=False), om, q, y, vec=True)[0]
            keys, tol = ("W", "pi", "sigma"), 1e-6
        else:
            from oracle import mca_oracle as O
            from prosper_amd.em.camodels.mca_et import MCA_ET
            W = np.abs(rng.normal(size=(D, H))) * 2 + 0.1
            y = np.where((rng.random_sample((N, H)) < 2.0 / H)[:, None, :], W[None], 0.0).max(axis=2) + rng.normal(size=(N, D))
            m, om = MCA_ET(D, H, Hp, gamma), O.make_model(D, H, Hp, gamma)
            p = m.check_params({"W": W * (1 + 0.05 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.2 / H, "sigma": 1.1})
            ostep = lambda a, q: O.em_step(O.Anneal(T=a["T"], Ncut_factor=a["Ncut_factor"]), om, q, y, vec=True)[0]
            keys, tol = ("W", "pi", "sigma"), 1e-6
        cp = lambda q: {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in q.items()}
        an, pg = sched(STEPS), cp(p)
        yd = torch.from_numpy(y).cuda()
        while not an.finished:
            pg = m.step(an, pg, {"y": yd}); an.next()
        an2, po = sched(STEPS), cp(p)
        while not an2.finished:
            po = ostep(an2, po); an2.next()
        for k in keys:
            close(tag, k, pg[k], po[k], tol)
    except Exception as e:
        print("EXCEPTION:", tag, type(e).__name__, str(e)[:300], flush=True); fails += 1
print("deviations / exceptions:", fails)
