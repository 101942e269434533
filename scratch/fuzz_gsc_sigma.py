"""Exploratory: GSC with diagonal / full noise covariance at random shapes -- E-step moments and one EM step against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import gsc_oracle as O
from prosper_amd.em.camodels.gsc_et import GSC
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
rng = np.random.RandomState(11)
fails = 0
cp = lambda q: {k: np.array(v, copy=True) for k, v in q.items()}
for trial in range(60):
    H = int(rng.randint(2, 40)); Hp = int(rng.randint(1, min(H, 7) + 1)); gamma = int(rng.randint(1, min(Hp, 4) + 1))
    D = int(rng.randint(2, 90)); N = int(rng.choice([H + 5, 3 * H, 200, 333])); T = float(rng.choice([1.0, 1.4]))
    kind = str(rng.choice(["diagonal", "full"]))
    W = rng.normal(size=(D, H))
    y = ((rng.random_sample((N, H)) < min(0.4, 2.0 / H)) * (1.5 + rng.normal(size=(N, H)))) @ W.T + rng.normal(size=(N, D)) * rng.uniform(0.7, 1.4, size=D)
    if kind == "diagonal":
        sig = rng.uniform(0.8, 1.6, size=D)
    else:
        A = rng.normal(size=(D, D)) * 0.15
        sig = np.eye(D) * 1.2 + A @ A.T
    p = {"W": W + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, min(0.4, 2.2 / H)), "mu": 1.4 + 0.1 * rng.normal(size=H),
         "psi_sq": np.diag(rng.uniform(0.8, 1.3, size=H)), "sigma_sq": sig}
    tag = "gsc %s D=%d H=%d H'=%d g=%d N=%d T=%.1f" % (kind, D, H, Hp, gamma, N, T)
    try:
        m = GSC(D, H, Hp, gamma, kind); om = O.make_model(D, H, Hp, gamma)
        d = m.select_Hprimes(cp(p), {"y": y}); ss = m.E_step(An(T=T), cp(p), d)
        cand = np.asarray(d["candidates"]).astype(np.int64)
        suff = O.e_step(O.Anneal(T=T), om, p, y, cand)
        w = 0.0
        for k in ("xpt_s", "xpt_sz"):
            w = max(w, float(np.max(np.abs(np.asarray(ss[k]) - suff[k]) / (1e-12 + 1e-9 * np.abs(suff[k])))))
        ref = O.m_step(om, cp(p), suff, y)
        got = m.M_step(An(T=T), cp(p), ss, d)
        tol = max(1e-8, 50 * np.linalg.cond(suff["xpt_szsz"].sum(0)) * np.finfo(float).eps)
        for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
            g, r = np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64)
            if not np.allclose(g, r, rtol=10 * tol, atol=tol * max(1.0, float(np.abs(r).max()))):
                print("DEVIATION:", tag, k, float(np.abs(g - r).max()), "tol", tol, flush=True); fails += 1
        if w > 1.0:
            print("E-STEP DEVIATION %.3g x tol:" % w, tag, flush=True); fails += 1
    except Exception as e:
        print("EXCEPTION:", tag, type(e).__name__, str(e)[:200], flush=True); fails += 1
print("deviations / exceptions:", fails)
