"""Exploratory: every non-empty subset of ``to_learn`` per model, one EM step against the oracle (which takes the same
argument); parameters not learned must come back unchanged."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
def subsets(keys):
    for r in range(1, len(keys) + 1):
        for c in itertools.combinations(keys, r):
            yield list(c)
fails = 0
def check(tag, got, ref, keys, p, learn, rtol=1e-8):
    global fails
    for k in keys:
        g, r = np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64)
        if k not in learn and not np.array_equal(g, np.asarray(p[k], dtype=np.float64)):
            print("CHANGED although not learned:", tag, k, flush=True); fails += 1
        if not np.allclose(g, r, rtol=rtol, atol=1e-9 * max(1.0, float(np.abs(r).max()))):
            print("DEVIATION:", tag, k, float(np.abs(g - r).max()), flush=True); fails += 1
rng = np.random.RandomState(0)
D, H, Hp, gamma, N = 40, 16, 5, 3, 900
# BSC
from oracle import bsc_oracle as BO
from prosper_amd.em.camodels.bsc_et import BSC_ET
W = rng.normal(size=(D, H)); mu_gt = rng.normal(size=D)
y = (rng.random_sample((N, H)) < 0.15) @ W.T + mu_gt + rng.normal(size=(N, D))
p = {"W": W + 0.1 * rng.normal(size=(D, H)), "pi": 0.15, "sigma": 1.1, "mu": mu_gt + 0.1 * rng.normal(size=D)}
om = BO.make_model(D, H, Hp, gamma)
for learn in subsets(["W", "pi", "sigma", "mu"]):
    for ncut in (0.0, 0.6):
        m = BSC_ET(D, H, Hp, gamma, to_learn=list(learn))
        got = m.step(An(T=1.2, Ncut_factor=ncut), {k: np.array(v, copy=True) for k, v in p.items()}, {"y": y})
        oan = BO.Anneal(T=1.2, Ncut_factor=ncut, anneal_prior=False)
        cand = BO.select_hprimes_vec(p["W"], y, Hp)
        lp = BO.e_step_vec(oan, p["W"], p["pi"], p["sigma"], p["mu"], y, cand, om["SM"], om["state_abs"])
        ref, _ = BO.m_step(oan, om, p["W"], p["pi"], p["sigma"], p["mu"], y, cand, lp, to_learn=tuple(learn), stats_fn=BO.m_step_stats_vec)
        check("bsc %s ncut=%.1f" % (learn, ncut), got, ref, ("W", "pi", "sigma", "mu"), p, learn)
print("bsc done", flush=True)
# MCA / MMCA
for kind in ("mca", "mmca"):
    if kind == "mca":
        from oracle import mca_oracle as O
        from prosper_amd.em.camodels.mca_et import MCA_ET as cls
        Wm = np.abs(rng.normal(size=(D, H))) * 2 + 0.1
        ym = np.where((rng.random_sample((N, H)) < 0.15)[:, None, :], Wm[None], 0.0).max(axis=2) + rng.normal(size=(N, D))
    else:
        from oracle import mmca_oracle as O
        from prosper_amd.em.camodels.mmca_et import MMCA_ET as cls
        Wm = rng.normal(size=(D, H)) * 3
        ym = O.generate_from_hidden(Wm, rng.random_sample((N, H)) < 0.15) + rng.normal(size=(N, D))
    omm = O.make_model(D, H, Hp, gamma)
    for learn in subsets(["W", "pi", "sigma"]):
        for ncut in (0.0, 0.6):
            m = cls(D, H, Hp, gamma, to_learn=list(learn))
            pm = m.check_params({"W": Wm * (1 + 0.05 * rng.uniform(-1, 1, size=(D, H))), "pi": 0.15, "sigma": 1.1})
            got = m.step(An(T=1.2, Ncut_factor=ncut), {k: (np.array(v, copy=True) if hasattr(v, "copy") else v) for k, v in pm.items()}, {"y": ym})
            oan = O.Anneal(T=1.2, Ncut_factor=ncut)
            cand = np.asarray(m.select_Hprimes(pm, {"y": ym})["candidates"]).astype(np.int64)
            lp = O.e_step_vec(oan, pm["W"], pm["pi"], pm["sigma"], ym, cand, omm["SM"], omm["state_abs"])
            ref, _ = O.m_step(oan, omm, pm["W"], pm["pi"], pm["sigma"], ym, cand, lp, to_learn=tuple(learn), vec=True)
            check("%s %s ncut=%.1f" % (kind, learn, ncut), got, ref, ("W", "pi", "sigma"), pm, learn)
    print(kind, "done", flush=True)
# DSC / TSC
for kind in ("dsc", "tsc"):
    states = np.array([-1., 0., 1.])
    Wd = rng.normal(size=(D, H)) * 2
    yd = rng.choice(states, size=(N, H), p=[0.07, 0.86, 0.07]) @ Wd.T + rng.normal(size=(N, D))
    for learn in subsets(["W", "pi", "sigma"]):
        for ncut in (0.0, 0.6):
            if kind == "dsc":
                from oracle import dsc_oracle as O
                from prosper_amd.em.camodels.dsc_et import DSC_ET
                m, omd, pi = DSC_ET(D, H, Hp, gamma, states=states, to_learn=list(learn)), O.make_model(D, H, Hp, gamma, states), np.array([0.08, 0.84, 0.08])
            else:
                from oracle import tsc_oracle as O
                from prosper_amd.em.camodels.tsc_et import TSC_ET
                m, omd, pi = TSC_ET(D, H, Hp, gamma, to_learn=list(learn)), O.make_model(D, H, Hp, gamma), 0.16
            pd_ = {"W": Wd + 0.1 * rng.normal(size=(D, H)), "pi": pi, "sigma": 1.1}
            got = m.step(An(T=1.2, Ncut_factor=ncut), {k: np.array(v, copy=True) for k, v in pd_.items()}, {"y": yd})
            oan = O.Anneal(T=1.2, Ncut_factor=ncut, anneal_prior=False)
            cand = np.asarray(m.select_Hprimes(pd_, {"y": yd})["candidates"])
            lp = O.e_step_vec(oan, omd, pd_["W"], pi, 1.1, yd, cand)
            ref, _ = O.m_step(oan, omd, pd_["W"], pi, 1.1, yd, cand, lp, to_learn=tuple(learn), vec=True)
            check("%s %s ncut=%.1f" % (kind, learn, ncut), got, ref, ("W", "pi", "sigma"), pd_, learn)
    print(kind, "done", flush=True)
# GSC
from oracle import gsc_oracle as GO
from prosper_amd.em.camodels.gsc_et import GSC
Wg = rng.normal(size=(D, H))
yg = ((rng.random_sample((N, H)) < 0.15) * (1.5 + rng.normal(size=(N, H)))) @ Wg.T + rng.normal(size=(N, D))
pg = {"W": Wg + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 0.15), "mu": 1.4 + 0.1 * rng.normal(size=H), "psi_sq": np.diag(rng.uniform(0.8, 1.3, size=H)), "sigma_sq": 1.2}
omg = GO.make_model(D, H, Hp, gamma)
cp = lambda q: {k: np.array(v, copy=True) for k, v in q.items()}
for learn in subsets(["W", "pi", "mu", "sigma_sq", "psi_sq"]):
    m = GSC(D, H, Hp, gamma, 'scalar', to_learn=list(learn))
    got = m.step(An(T=1.2), cp(pg), {"y": yg})
    cand = GO.select_hprimes(pg, yg, Hp)
    suff = GO.e_step(GO.Anneal(T=1.2), omg, pg, yg, cand)
    ref = GO.m_step(omg, cp(pg), suff, yg, to_learn=tuple(learn))
    check("gsc %s" % (learn,), got, ref, ("W", "pi", "mu", "sigma_sq", "psi_sq"), pg, learn, rtol=1e-7)
print("gsc done; deviations:", fails, flush=True)
