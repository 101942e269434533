#!/usr/bin/env python3 -B
"""Mint golden vectors for the truncated-EM hot path from the REFERENCE itself.

Runs only in the build container: imports ml-uol/prosper from /root/reference through the
single-rank mpi4py / tables shims in ./_shims and the NumPy-2 alias patch (SURVEY appendix
A), feeds it seeded inputs and stores inputs + outputs as small .npz files next to this
script.  The fixtures (data) are committed; the reference (code) is not and never travels.

    python -B tests/golden/make_golden.py            # regenerates every *.npz here

Fixtures (all float64, bit-for-bit what the reference returned):
  bsc_step_<case>.npz   one select_Hprimes -> E_step -> M_step of BSC_ET on seeded data,
                        incl. L / N / N_use captured from the reference's dlog
  bsc_traj_c1.npz       BASELINE config 1: bars data D=25 H=10 H'=5 gamma=3 N=2000,
                        20 EM.run steps, parameters after every step
  bsc_init_c1.npz       generate_data + standard_init for fixed seeds (RNG stream order)
  anneal_tracks.npz     LinearAnnealing values per step for the bars-learning schedule
  schedule_<name>.npz   the reference's own 50-step schedule at config-2 / 4 / 5 dimensions (+ MMCA, DSC, TSC): per-step
                        L, N_use, parameters (tests/golden/schedule_inputs.py holds the seeded inputs).  Slow: the
                        reference needs 9 minutes each for bsc_c2 and gsc_c4 -- `make_golden.py schedule` mints only
                        these, `make_golden.py schedule mca_c5,dsc` a subset
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(1, "/root/reference")

import numpy as np

for _n, _t in (("int", int), ("bool", bool), ("str", str), ("object", object), ("float", float)):
    if _n not in np.__dict__:
        setattr(np, _n, _t)

from prosper.utils.datalog import dlog, DataHandler          # noqa: E402
from prosper.em import EM                                    # noqa: E402
from prosper.em.annealing import LinearAnnealing             # noqa: E402
from prosper.em.camodels.bsc_et import BSC_ET                # noqa: E402
from prosper.em.camodels.mca_et import MCA_ET                # noqa: E402
from prosper.em.camodels.gsc_et import GSC                   # noqa: E402
from prosper.em.camodels.mmca_et import MMCA_ET              # noqa: E402
from prosper.em.camodels.dsc_et import DSC_ET                # noqa: E402
from prosper.em.camodels import tsc_et as _tsc               # noqa: E402
from prosper.em import Model as _Model                       # noqa: E402
from prosper.utils.barstest import generate_bars_dict        # noqa: E402


class Capture(DataHandler):
    rows = {}

    def append(self, tblname, value):
        Capture.rows.setdefault(tblname, []).append(np.array(value, copy=True))


dlog.set_handler(("L", "N", "N_use", "prior_mass"), Capture)


def gsc_step_case(name, D, H, Hp, gamma, N, seed, T, full_psi=False, sigma_type="scalar", presteps=0):
    """select_Hprimes -> E_step -> M_step (+ compute_lpj) of GSC with scalar sigma_sq.  The reference
    returns its statistics in candidate-bucket order; they are mapped back to datapoint order here.
    ``presteps``: the fixture's INPUT parameters are what that many reference EM steps leave behind -- from the first
    M-step on psi_sq is NOT symmetric (gsc_et.py:660-675: the term -2 outer(mu * xpt_s, xpt_sz)), and with it Lambda,
    its inverse and sum xpt_szsz (which gsc_et.py:625 inverts as it is)."""
    rng = np.random.RandomState(seed)
    W_gt = rng.normal(size=(D, H))
    pi_gt = np.full(H, min(0.4, 2.0 / H))
    psi_gt = np.eye(H)
    mu_gt = np.ones(H) * 1.5
    s = rng.random_sample((N, H)) <= pi_gt
    z = np.where(s, mu_gt[None, :] + rng.normal(size=(N, H)), 0.0)
    y = z @ W_gt.T + rng.normal(size=(N, D))
    psi0 = np.diag(rng.uniform(0.6, 1.6, size=H))
    if full_psi:
        Q = 0.08 * rng.normal(size=(H, H))
        psi0 = psi0 + Q @ Q.T
    params = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": np.clip(pi_gt * rng.uniform(0.7, 1.4, size=H), 0.02, 0.9),
              "mu": mu_gt + 0.2 * rng.normal(size=H), "psi_sq": psi0, "sigma_sq": 1.3}
    if sigma_type == "diagonal":
        params["sigma_sq"] = rng.uniform(0.8, 1.8, size=D)
    elif sigma_type == "full":
        Qs = 0.15 * rng.normal(size=(D, D))
        params["sigma_sq"] = np.diag(rng.uniform(0.8, 1.8, size=D)) + Qs @ Qs.T
    model = GSC(D, H, Hp, gamma, sigma_type)
    anneal = FixedAnneal(T=T)
    for _ in range(presteps):
        d0 = model.select_Hprimes(params, {"y": y.copy()})
        params = model.M_step(anneal, params, model.E_step(anneal, params, d0), d0)
        params = {k: np.array(params[k], copy=True) for k in ("W", "pi", "mu", "psi_sq", "sigma_sq")}
    if presteps:
        assert np.abs(params["psi_sq"] - params["psi_sq"].T).max() > 0
    inp = {k: np.array(v, copy=True) for k, v in params.items()}
    logpj, cands = model.compute_lpj(anneal, {k: np.array(v, copy=True) for k, v in inp.items()}, {"y": y.copy()})
    data = model.select_Hprimes(params, {"y": y.copy()})
    order = np.concatenate([np.array(c["ind"]) for c in data["data_clusters"].values()])
    suff = model.E_step(anneal, params, data)
    assert np.array_equal(data["y"], y[order])
    inv = np.argsort(order)
    new = model.M_step(anneal, params, suff, data)
    out = dict(D=D, H=H, Hprime=model.Hprime, gamma=model.gamma, T=T, y=y, sigma_type=sigma_type,
               candidates=cands.astype(np.int64),
               logpj=logpj, xpt_s=suff["xpt_s"][inv], xpt_sz=suff["xpt_sz"][inv],
               sum_xpt_ss=suff["xpt_ss"].sum(axis=0), sum_xpt_szsz=suff["xpt_szsz"].sum(axis=0),
               state_matrix=model.state_matrix)
    for k, v in inp.items():
        out[k] = v
    for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
        out[k + "_new"] = np.asarray(new[k])
    assert all(np.isfinite(out[k + "_new"]).all() for k in ("W", "pi", "mu", "psi_sq", "sigma_sq")), name
    np.savez_compressed(os.path.join(HERE, "gsc_step_%s.npz" % name), **out)
    s2n = np.asarray(new["sigma_sq"])
    print("gsc_step_%s: N=%d K=%d clusters=%d mean sigma_sq_new=%.6f" % (
        name, N, logpj.shape[1], len(data["data_clusters"]), float(np.diag(s2n).mean() if s2n.ndim == 2 else s2n.mean())))


def bsc_inference_case():
    """CAModel.inference (camodels/__init__.py:256-375) of BSC_ET: top-K states, marginals, adaptive H'/gamma."""
    D, H, Hp, gamma, N = 25, 10, 4, 2, 60
    rng = np.random.RandomState(41)
    W = 10 * generate_bars_dict(H) + 0.3 * rng.normal(size=(D, H))
    params = {"W": W, "pi": 0.25, "sigma": 1.5}
    s = rng.random_sample((N, H)) < 0.25
    y = s.astype(float) @ (10 * generate_bars_dict(H)).T + rng.normal(scale=1.5, size=(N, D))
    model = BSC_ET(D, H, Hp, gamma)
    anneal = LinearAnnealing(1)
    anneal["T"] = [(0, 1.)]
    anneal["anneal_prior"] = False
    out = {}
    for tag, kw in (("plain", dict(topK=5, adaptive=False)), ("adaptive", dict(topK=4, adaptive=True)),
                    ("capped", dict(topK=3, adaptive=True, Hprime_max=5, gamma_max=3, logprob=True))):
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            res = model.inference(anneal, {k: np.array(v, copy=True) for k, v in params.items()}, {"y": y.copy()}, **kw)
        for k, v in res.items():
            out["%s_%s" % (tag, k)] = v
    assert (model.Hprime, model.gamma) == (Hp, gamma)
    np.savez_compressed(os.path.join(HERE, "bsc_inference.npz"), D=D, H=H, Hprime=Hp, gamma=gamma, y=y, W=W,
                        pi=params["pi"], sigma=params["sigma"], **out)
    print("bsc_inference: adaptive gamma max %d, Hprime max %d" % (out["adaptive_gamma"].max(), out["adaptive_Hprime"].max()))


def _inference_runs(model, anneal, params, y, runs):
    import io, contextlib
    out = {}
    for tag, kw in runs:
        with contextlib.redirect_stdout(io.StringIO()):
            res = model.inference(anneal, {k: np.array(v, copy=True) for k, v in params.items()}, {"y": y.copy()}, **kw)
        for k, v in res.items():
            out["%s_%s" % (tag, k)] = v
    return out


_INFERENCE_RUNS = (("plain", dict(topK=5, adaptive=False)), ("adaptive", dict(topK=4, adaptive=True)),
                   ("capped", dict(topK=3, adaptive=True, Hprime_max=5, gamma_max=3, logprob=True)))


def inference_big_case(kind, D, H, Hp, gamma, N, seed):
    """CAModel.inference on thousands of datapoints (inputs: schedule_inputs.py): plain top-K and the adaptive H' / gamma
    growth with caps -- ties, repeated growth rounds and every exit of the loop get exercised, which 60 datapoints do not."""
    from schedule_inputs import schedule_inputs
    y, p0 = schedule_inputs(kind, D, H, N, seed)
    if kind == "gsc":
        model = GSC(D, H, Hp, gamma, sigma_sq_type="scalar")
    else:
        model = {"bsc": BSC_ET, "mca": MCA_ET}[kind](D, H, Hp, gamma)
    anneal = LinearAnnealing(1)
    anneal["T"] = [(0, 1.)]
    anneal["anneal_prior"] = False
    out = _inference_runs(model, anneal, p0, y, (("plain", dict(topK=4, adaptive=False)),
                                                   ("capped", dict(topK=3, adaptive=True, Hprime_max=Hp + 1, gamma_max=gamma + 1))))
    np.savez_compressed(os.path.join(HERE, "%s_inference_big.npz" % kind), D=D, H=H, Hprime=Hp, gamma=gamma, N=N, seed=seed, **out)
    print("%s_inference_big: N=%d, capped gamma max %d, Hprime max %d, %d datapoints grown" % (
        kind, N, out["capped_gamma"].max(), out["capped_Hprime"].max(), int((out["capped_Hprime"] > Hp).sum())))


def mca_inference_case():
    """CAModel.inference (camodels/__init__.py:256-375) of MCA_ET: compute_lpj = select_Hprimes + E_step
    (mca_et.py:88-179), top-K states, marginals, adaptive H'/gamma."""
    D, H, Hp, gamma, N = 25, 10, 4, 2, 60
    rng = np.random.RandomState(43)
    W_gt = 10 * generate_bars_dict(H)
    W = W_gt + 0.5 * np.abs(rng.normal(size=(D, H)))
    params = {"W": W, "pi": 0.25, "sigma": 1.5}
    s = rng.random_sample((N, H)) < 0.25
    y = np.zeros((N, D))
    for n in range(N):
        if s[n].any():
            y[n] = np.maximum(0.0, W_gt.T[s[n]].max(axis=0))
    y += rng.normal(scale=1.5, size=(N, D))
    model = MCA_ET(D, H, Hp, gamma)
    anneal = FixedAnneal(T=1.0)
    out = _inference_runs(model, anneal, params, y, _INFERENCE_RUNS)
    assert (model.Hprime, model.gamma) == (Hp, gamma)
    np.savez_compressed(os.path.join(HERE, "mca_inference.npz"), D=D, H=H, Hprime=Hp, gamma=gamma, y=y, W=W,
                        pi=params["pi"], sigma=params["sigma"], **out)
    print("mca_inference: adaptive gamma max %d, Hprime max %d" % (out["adaptive_gamma"].max(), out["adaptive_Hprime"].max()))


def mmca_inference_case():
    """CAModel.inference of MMCA_ET (signed max-magnitude causes, mmca_et.py:96-199)."""
    D, H, Hp, gamma, N = 20, 10, 4, 2, 60
    rng = np.random.RandomState(44)
    W_gt = rng.normal(size=(D, H)) * 3.0
    W = W_gt * (1.0 + 0.2 * rng.uniform(-1, 1, size=(D, H)))
    params = {"W": W, "pi": 0.2, "sigma": 1.2}
    model = MMCA_ET(D, H, Hp, gamma)
    s = rng.random_sample((N, H)) < 0.2
    y = model.generate_from_hidden({"W": W_gt, "pi": 0.2, "sigma": 0.0}, {"s": s})["y"] + rng.normal(scale=1.0, size=(N, D))
    anneal = FixedAnneal(T=1.0)
    out = _inference_runs(model, anneal, params, y, _INFERENCE_RUNS)
    assert (model.Hprime, model.gamma) == (Hp, gamma)
    np.savez_compressed(os.path.join(HERE, "mmca_inference.npz"), D=D, H=H, Hprime=Hp, gamma=gamma, y=y, W=W,
                        pi=params["pi"], sigma=params["sigma"], **out)
    print("mmca_inference: adaptive gamma max %d, Hprime max %d" % (out["adaptive_gamma"].max(), out["adaptive_Hprime"].max()))


def gsc_posterior_hprime_case():
    """GSC.compute_posterior_hprime (gsc_et.py:260-398) on one data cluster -- the un-normalised sums over the multi-cause
    states E_step accumulates per cluster (:550) -- for a scalar and a full noise model, with a psi_sq that has been
    through an M-step (non-symmetric), T != 1."""
    out = {}
    for tag, sigma_type, seed in (("scalar", "scalar", 71), ("full", "full", 72)):
        D, H, Hp, gamma, N, T = 20, 10, 4, 3, 60, 1.3
        rng = np.random.RandomState(seed)
        W_gt = rng.normal(size=(D, H))
        s = rng.random_sample((N, H)) <= 0.2
        y = np.where(s, 1.5 + rng.normal(size=(N, H)), 0.0) @ W_gt.T + rng.normal(size=(N, D))
        params = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": np.full(H, 0.2), "mu": 1.5 + 0.2 * rng.normal(size=H),
                  "psi_sq": np.diag(rng.uniform(0.6, 1.6, size=H)), "sigma_sq": 1.3}
        if sigma_type == "full":
            Qs = 0.15 * rng.normal(size=(D, D))
            params["sigma_sq"] = np.diag(rng.uniform(0.8, 1.8, size=D)) + Qs @ Qs.T
        model = GSC(D, H, Hp, gamma, sigma_type)
        anneal = FixedAnneal(T=T)
        d0 = model.select_Hprimes(params, {"y": y.copy()})
        params = model.M_step(anneal, params, model.E_step(anneal, params, d0), d0)      # -> non-symmetric psi_sq
        params = {k: np.array(params[k], copy=True) for k in ("W", "pi", "mu", "psi_sq", "sigma_sq")}
        data = model.select_Hprimes(params, {"y": y.copy()})
        key = max(data["data_clusters"], key=lambda k: data["data_clusters"][k]["data"].shape[0])
        cl = data["data_clusters"][key]
        if sigma_type == "full":
            sinv = np.linalg.inv(params["sigma_sq"])
            B = sinv
        else:
            sinv = 1. / params["sigma_sq"]
            B = sinv * np.eye(D)
        mp = dict(params, sigma_sq_inv=sinv, B=B)
        res = model.compute_posterior_hprime(anneal, mp, {"y": cl["data"].copy(), "candidates": cl["hprimes"].copy()})
        out.update({tag + "_" + k: np.asarray(v) for k, v in params.items()})
        out.update({tag + "_y": cl["data"], tag + "_cand": np.asarray(cl["hprimes"]).astype(np.int64)})
        out.update({tag + "_" + k: np.asarray(v, dtype=np.float64) for k, v in res.items()})
        print("gsc_posterior_hprime %s: cluster of %d rows, candidates %s" % (tag, cl["data"].shape[0], cl["hprimes"]))
    np.savez_compressed(os.path.join(HERE, "gsc_posterior_hprime.npz"), D=20, H=10, Hprime=4, gamma=3, T=1.3, **out)


def gsc_inference_case():
    """CAModel.inference of GSC through its own compute_lpj (gsc_et.py:811-944), plus component_scores
    (gsc_et.py:752-809) of the same parameters."""
    D, H, Hp, gamma, N = 20, 10, 4, 2, 60
    rng = np.random.RandomState(45)
    W_gt = rng.normal(size=(D, H))
    pi_gt = np.full(H, 0.2)
    mu_gt = np.ones(H) * 1.5
    s = rng.random_sample((N, H)) <= pi_gt
    z = np.where(s, mu_gt[None, :] + rng.normal(size=(N, H)), 0.0)
    y = z @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": np.clip(pi_gt * rng.uniform(0.7, 1.4, size=H), 0.02, 0.9),
              "mu": mu_gt + 0.2 * rng.normal(size=H), "psi_sq": np.diag(rng.uniform(0.6, 1.6, size=H)), "sigma_sq": 1.3}
    model = GSC(D, H, Hp, gamma, "scalar")
    anneal = FixedAnneal(T=1.0)
    out = _inference_runs(model, anneal, params, y, _INFERENCE_RUNS)
    assert (model.Hprime, model.gamma) == (Hp, gamma)
    scores = model.component_scores({k: np.array(v, copy=True) for k, v in params.items()}, {"y": y.copy()})
    np.savez_compressed(os.path.join(HERE, "gsc_inference.npz"), D=D, H=H, Hprime=Hp, gamma=gamma, y=y,
                        component_scores=scores, **params, **out)
    print("gsc_inference: adaptive gamma max %d, Hprime max %d" % (out["adaptive_gamma"].max(), out["adaptive_Hprime"].max()))


def mca_step_case(name, D, H, Hp, gamma, N, seed, T, Ncut, bars=False):
    """One check_params -> select_Hprimes -> E_step -> M_step of MCA_ET on seeded data."""
    rng = np.random.RandomState(seed)
    if bars:
        W_gt = 10 * generate_bars_dict(H)
        pi_gt, sigma_gt = 2. / H, 2.0
        W0 = W_gt + 0.5 * np.abs(rng.normal(size=(D, H)))
    else:
        W_gt = np.abs(rng.normal(size=(D, H))) * 2.0 + 0.1
        pi_gt, sigma_gt = min(0.45, 2.0 / H), 1.0
        W0 = W_gt * (1.0 + 0.2 * rng.uniform(-1, 1, size=(D, H)))
    model = MCA_ET(D, H, Hp, gamma)
    s = rng.random_sample((N, H)) < pi_gt
    y = np.zeros((N, D))
    for n in range(N):
        if s[n].any():
            y[n] = np.maximum(0.0, W_gt.T[s[n]].max(axis=0))
    y += rng.normal(scale=sigma_gt, size=(N, D))
    params = {"W": W0, "pi": pi_gt * 1.3, "sigma": sigma_gt * 1.2}
    inp = {k: np.array(v, copy=True) for k, v in params.items()}
    anneal = FixedAnneal(T=T, Ncut_factor=Ncut)
    Capture.rows.clear()
    params = model.check_params(params)
    data = model.select_Hprimes(params, {"y": y.copy()})
    ss = model.E_step(anneal, params, data)
    new = model.M_step(anneal, params, ss, data)
    assert np.isfinite(new["W"]).all() and np.isfinite(new["Q"]), name
    np.savez_compressed(os.path.join(HERE, "mca_step_%s.npz" % name), D=D, H=H, Hprime=Hp, gamma=gamma, T=T,
                        Ncut_factor=Ncut, y=y, W=inp["W"], pi=inp["pi"], sigma=inp["sigma"],
                        candidates=data["candidates"].astype(np.int64), logpj=ss["logpj"],
                        W_new=new["W"], pi_new=new["pi"], sigma_new=new["sigma"], Q=new["Q"],
                        N_use=Capture.rows["N_use"][0], state_matrix=model.state_matrix)
    print("mca_step_%s: N=%d K=%d Q=%.6f N_use=%d" % (name, N, ss["logpj"].shape[1], new["Q"], Capture.rows["N_use"][0]))


def mmca_step_case(name, D, H, Hp, gamma, N, seed, T, Ncut):
    """One check_params -> select_Hprimes -> E_step -> M_step of MMCA_ET (signed max-magnitude causes)."""
    rng = np.random.RandomState(seed)
    W_gt = rng.normal(size=(D, H)) * 3.0
    pi_gt, sigma_gt = min(0.45, 2.0 / H), 1.0
    W0 = W_gt * (1.0 + 0.2 * rng.uniform(-1, 1, size=(D, H)))
    W0[rng.random_sample((D, H)) < 0.03] = 1e-6        # some entries below tol: check_params must lift them
    model = MMCA_ET(D, H, Hp, gamma)
    s = rng.random_sample((N, H)) < pi_gt
    gen = model.generate_from_hidden({"W": W_gt, "pi": pi_gt, "sigma": 0.0}, {"s": s})
    y_clean = gen["y"]
    y = y_clean + rng.normal(scale=sigma_gt, size=(N, D))
    params = {"W": W0, "pi": pi_gt * 1.3, "sigma": sigma_gt * 1.2}
    inp = {k: np.array(v, copy=True) for k, v in params.items()}
    anneal = FixedAnneal(T=T, Ncut_factor=Ncut)
    Capture.rows.clear()
    params = model.check_params(params)
    data = model.select_Hprimes(params, {"y": y.copy()})
    ss = model.E_step(anneal, params, data)
    new = model.M_step(anneal, params, ss, data)
    assert np.isfinite(new["W"]).all() and np.isfinite(new["Q"]), name
    np.savez_compressed(os.path.join(HERE, "mmca_step_%s.npz" % name), D=D, H=H, Hprime=Hp, gamma=gamma, T=T,
                        Ncut_factor=Ncut, y=y, s=s, y_clean=y_clean, W_gt=W_gt, W=inp["W"], pi=inp["pi"],
                        sigma=inp["sigma"], candidates=data["candidates"].astype(np.int64), logpj=ss["logpj"],
                        W_new=new["W"], pi_new=new["pi"], sigma_new=new["sigma"], Q=new["Q"],
                        N_use=Capture.rows["N_use"][0], state_matrix=model.state_matrix)
    print("mmca_step_%s: N=%d K=%d Q=%.6f N_use=%d" % (name, N, ss["logpj"].shape[1], new["Q"], Capture.rows["N_use"][0]))


def dsc_step_case(name, D, H, Hp, gamma, N, seed, T, Ncut, anneal_prior, states, pi_gt):
    """One select_Hprimes -> E_step -> M_step of DSC_ET (K-ary latents) on seeded data."""
    import warnings
    warnings.simplefilter("ignore")
    rng = np.random.RandomState(seed)
    states = np.asarray(states, dtype=np.float64)
    pi_gt = np.asarray(pi_gt, dtype=np.float64)
    W_gt = rng.normal(size=(D, H)) * 2.0
    sigma_gt = 1.0
    s = rng.choice(states, size=(N, H), replace=True, p=pi_gt)
    y = s @ W_gt.T + rng.normal(scale=sigma_gt, size=(N, D))
    model = DSC_ET(D, H, Hp, gamma, states=states)
    pi0 = pi_gt * rng.uniform(0.8, 1.25, size=pi_gt.shape)
    params = {"W": W_gt + 0.3 * rng.normal(size=(D, H)), "pi": pi0 / pi0.sum(), "sigma": sigma_gt * 1.2}
    inp = {k: np.array(v, copy=True) for k, v in params.items()}
    anneal = FixedAnneal(T=T, Ncut_factor=Ncut, anneal_prior=anneal_prior)
    Capture.rows.clear()
    data = model.select_Hprimes(params, {"y": y.copy()})
    ss = model.E_step(anneal, params, data)
    new = model.M_step(anneal, params, ss, data)
    assert np.isfinite(new["W"]).all(), name
    np.savez_compressed(os.path.join(HERE, "dsc_step_%s.npz" % name), D=D, H=H, Hprime=Hp, gamma=gamma, T=T,
                        Ncut_factor=Ncut, anneal_prior=anneal_prior, states=states, y=y, W=inp["W"], pi=inp["pi"],
                        sigma=inp["sigma"], candidates=data["candidates"].astype(np.int64), logpj=ss["logpj"],
                        W_new=new["W"], pi_new=new["pi"], sigma_new=new["sigma"], Q=new["Q"],
                        L=Capture.rows["L"][0], N_use=Capture.rows["N_use"][0], prior_mass=Capture.rows["prior_mass"][0],
                        state_matrix=model.state_matrix, single_state_matrix=model.single_state_matrix,
                        state_abs=model.state_abs)
    print("dsc_step_%s: N=%d K=%d L=%.6f N_use=%d" % (name, N, ss["logpj"].shape[1], Capture.rows["L"][0],
                                                      Capture.rows["N_use"][0]))


def dsc_inference_case():
    """DSC_ET.inference (dsc_et.py:927-1059): top-K states, marginals, adaptive H'/gamma -- including what the
    adaptive rounds really compute (the regenerated ``state_abs`` is 1-D there, dsc_et.py:1048)."""
    import io, contextlib, warnings
    warnings.simplefilter("ignore")
    D, H, Hp, gamma, N = 20, 9, 4, 2, 50
    rng = np.random.RandomState(71)
    states = np.array([-1., 0., 1.])
    pi = np.array([0.12, 0.76, 0.12])
    W = rng.normal(size=(D, H)) * 3.0
    s = rng.choice(states, size=(N, H), p=pi)
    y = s @ W.T + rng.normal(size=(N, D))
    params = {"W": W + 0.2 * rng.normal(size=(D, H)), "pi": pi, "sigma": 1.1}
    anneal = FixedAnneal(T=1.0)
    out = {}
    for tag, kw in (("plain", dict(topK=5, adaptive=False)), ("adaptive", dict(topK=4, adaptive=True)),
                    ("capped", dict(topK=3, adaptive=True, Hprime_max=5, gamma_max=3, logprob=True))):
        model = DSC_ET(D, H, Hp, gamma, states=states)          # fresh: inference leaves state_abs 1-D behind
        with contextlib.redirect_stdout(io.StringIO()):
            res = model.inference(anneal, {k: np.array(v, copy=True) for k, v in params.items()}, {"y": y.copy()}, **kw)
        for k, v in res.items():
            out["%s_%s" % (tag, k)] = v
    np.savez_compressed(os.path.join(HERE, "dsc_inference.npz"), D=D, H=H, Hprime=Hp, gamma=gamma, y=y, states=states,
                        W=params["W"], pi=pi, sigma=params["sigma"], **out)
    print("dsc_inference: adaptive gamma max %d, Hprime max %d" % (out["adaptive_gamma"].max(), out["adaptive_Hprime"].max()))


def _make_tsc(D, H, Hp, gamma):
    """TSC_ET cannot be constructed upstream (``states`` is undefined in tsc_et.py:131): build the
    object without __init__ and set what __init__ would set, with ``self.states`` in place of ``states``."""
    from mpi4py import MPI
    m = object.__new__(_tsc.TSC_ET)
    _Model.__init__(m, MPI.COMM_WORLD)
    m.to_learn = ['W', 'pi', 'sigma']
    m.states = np.array([-1., 0., 1.])
    m.gamma, m.D, m.H, m.Hprime = gamma, D, H, Hp
    m.single_state_matrix, m.state_matrix, m.no_states, m.state_abs = _tsc.generate_state_matrix(Hp, gamma, H, m.states)
    tol = 1e-5
    m.noise_policy = {'W': (-np.inf, +np.inf, False), 'pi': (tol, 1. - tol, False), 'sigma': (0., +np.inf, False)}
    return m


def tsc_step_case(name, D, H, Hp, gamma, N, seed, T, Ncut, anneal_prior):
    """One select_Hprimes -> E_step -> M_step of TSC_ET (ternary sparse coding, scalar pi)."""
    import warnings
    warnings.simplefilter("ignore")
    rng = np.random.RandomState(seed)
    W_gt = rng.normal(size=(D, H)) * 2.5
    pi_gt, sigma_gt = min(0.4, 2.5 / H), 1.0
    s = rng.choice([-1., 0., 1.], size=(N, H), p=[pi_gt / 2, 1 - pi_gt, pi_gt / 2])
    y = s @ W_gt.T + rng.normal(scale=sigma_gt, size=(N, D))
    model = _make_tsc(D, H, Hp, gamma)
    params = {"W": W_gt + 0.3 * rng.normal(size=(D, H)), "pi": pi_gt * 1.2, "sigma": sigma_gt * 1.15}
    inp = {k: np.array(v, copy=True) for k, v in params.items()}
    anneal = FixedAnneal(T=T, Ncut_factor=Ncut, anneal_prior=anneal_prior)
    Capture.rows.clear()
    data = model.select_Hprimes(params, {"y": y.copy()})
    ss = model.E_step(anneal, params, data)
    new = model.M_step(anneal, params, ss, data)
    assert np.isfinite(new["W"]).all(), name
    cand = data["candidates"].astype(np.int64)
    dup = np.mean([len(set(r)) < len(r) for r in cand])
    np.savez_compressed(os.path.join(HERE, "tsc_step_%s.npz" % name), D=D, H=H, Hprime=Hp, gamma=gamma, T=T,
                        Ncut_factor=Ncut, anneal_prior=anneal_prior, y=y, W=inp["W"], pi=inp["pi"], sigma=inp["sigma"],
                        candidates=cand, logpj=ss["logpj"], W_new=new["W"], pi_new=new["pi"], sigma_new=new["sigma"],
                        Q=new["Q"], L=Capture.rows["L"][0], N_use=Capture.rows["N_use"][0],
                        state_matrix=model.state_matrix, single_state_matrix=model.single_state_matrix,
                        no_states=model.no_states, state_abs=model.state_abs)
    print("tsc_step_%s: N=%d states=%d L=%.6f N_use=%d, %.0f%% rows with a repeated candidate" % (
        name, N, ss["logpj"].shape[1], Capture.rows["L"][0], Capture.rows["N_use"][0], 100 * dup))


def tsc_inference_case():
    """TSC_ET.inference (tsc_et.py:546-680) on the constructor-bypassed object: top-K states, signed and
    absolute marginals, adaptive H'/gamma."""
    import io, contextlib, warnings
    warnings.simplefilter("ignore")
    D, H, Hp, gamma, N = 20, 9, 4, 2, 50
    rng = np.random.RandomState(91)
    pi = 0.25
    W = rng.normal(size=(D, H)) * 3.0
    s = rng.choice([-1., 0., 1.], size=(N, H), p=[pi / 2, 1 - pi, pi / 2])
    y = s @ W.T + rng.normal(size=(N, D))
    params = {"W": W + 0.2 * rng.normal(size=(D, H)), "pi": pi, "sigma": 1.1}
    anneal = FixedAnneal(T=1.0)
    out = {}
    for tag, kw in (("plain", dict(topK=5, adaptive=False)), ("adaptive", dict(topK=4, adaptive=True)),
                    ("capped", dict(topK=3, adaptive=True, Hprime_max=5, gamma_max=3, logprob=True))):
        model = _make_tsc(D, H, Hp, gamma)
        with contextlib.redirect_stdout(io.StringIO()):
            res = model.inference(anneal, {k: np.array(v, copy=True) for k, v in params.items()}, {"y": y.copy()}, **kw)
        for k, v in res.items():
            out["%s_%s" % (tag, k)] = v
    np.savez_compressed(os.path.join(HERE, "tsc_inference.npz"), D=D, H=H, Hprime=Hp, gamma=gamma, y=y,
                        W=params["W"], pi=pi, sigma=params["sigma"], **out)
    print("tsc_inference: adaptive gamma max %d, Hprime max %d" % (out["adaptive_gamma"].max(), out["adaptive_Hprime"].max()))


class FixedAnneal(dict):
    """One annealing position; unknown keys -> 0.0 like LinearAnnealing.__getitem__."""
    crit_params = []

    def __missing__(self, k):
        return 0.0


def bsc_step_case(name, D, H, Hp, gamma, N, seed, T, Ncut, anneal_prior, bars=False,
                  mu=False, to_learn=("W", "pi", "sigma"), sigma_gt=1.0, amp=1.0, big=False, pi_gt=None):
    """``big`` (config-2 dimensions): the inputs are rounded to float32-representable values and stored as float32
    (the reference runs on their exact float64 upcasts), and the all-reduced statistics Wq / Wp the reference hands
    to ``np.linalg.lstsq`` (bsc_et.py:373-380) are captured: with N < H datapoints Wq is rank-deficient and W_new
    is then only defined up to the SVD cutoff -- the statistics are what pins the path."""
    rng = np.random.RandomState(seed)
    if bars:
        W_gt = 10 * generate_bars_dict(H)
        pi_gt, sigma_gt = 2. / H, 2.0
    else:
        W_gt = amp * rng.normal(size=(D, H))
        pi_gt = min(0.45, 2.0 / H) if pi_gt is None else pi_gt
    model = BSC_ET(D, H, Hp, gamma, to_learn=list(to_learn))
    s = rng.random_sample((N, H)) < pi_gt
    y = s.astype(float) @ W_gt.T + rng.normal(scale=sigma_gt, size=(N, D))
    mu_vec = rng.normal(scale=0.3, size=D) if mu else None
    if mu:
        y = y + mu_vec
    params = {"W": W_gt + 0.3 * amp * rng.normal(size=(D, H)), "pi": pi_gt * 1.3, "sigma": sigma_gt * 1.2}
    if mu:
        params["mu"] = mu_vec + 0.05 * rng.normal(size=D)
    if big:
        y = y.astype(np.float32).astype(np.float64)
        params["W"] = params["W"].astype(np.float32).astype(np.float64)
    anneal = FixedAnneal(T=T, Ncut_factor=Ncut, anneal_prior=anneal_prior)
    inp = {k: np.array(v, copy=True) for k, v in params.items()}
    data = {"y": y.copy()}
    Capture.rows.clear()
    data = model.select_Hprimes(params, data)
    ss = model.E_step(anneal, params, data)
    seen, lstsq = {}, np.linalg.lstsq

    def spy(a, b, rcond=None):
        seen["Wq"], seen["Wp"], seen["rcond"] = np.array(a, copy=True), np.array(b, copy=True), rcond
        return lstsq(a, b, rcond=rcond)
    np.linalg.lstsq = spy
    try:
        new = model.M_step(anneal, params, ss, data)
    finally:
        np.linalg.lstsq = lstsq
    assert np.isfinite(new["W"]).all() and np.isfinite(Capture.rows["L"][0]), name
    out = dict(D=D, H=H, Hprime=Hp, gamma=gamma, T=T, Ncut_factor=Ncut, anneal_prior=bool(anneal_prior),
               to_learn=np.array(list(to_learn)), y=y, W=inp["W"], pi=inp["pi"], sigma=inp["sigma"],
               mu=inp.get("mu", np.zeros(D)), has_mu=bool(mu),
               candidates=data["candidates"].astype(np.int64), logpj=ss["logpj"],
               W_new=new["W"], pi_new=new["pi"], sigma_new=new["sigma"], mu_new=new["mu"],
               L=Capture.rows["L"][0], N=Capture.rows["N"][0], N_use=Capture.rows["N_use"][0],
               state_matrix=model.state_matrix, state_abs=model.state_abs)
    if big:
        sv = np.linalg.svd(seen["Wq"], compute_uv=False)
        out.update(y=y.astype(np.float32), W=inp["W"].astype(np.float32), Wq=seen["Wq"], Wp=seen["Wp"],
                   rcond=-1.0 if seen["rcond"] is not None else np.nan, Wq_rank_ratio=sv[-1] / sv[0])
        assert np.array_equal(out["y"].astype(np.float64), y) and np.array_equal(out["W"].astype(np.float64), inp["W"])
    np.savez_compressed(os.path.join(HERE, "bsc_step_%s.npz" % name), **out)
    print("bsc_step_%s: N=%d K=%d L=%.6f N_use=%d%s" % (name, N, ss["logpj"].shape[1], out["L"], out["N_use"],
                                                         " smin/smax(Wq)=%.2e" % out["Wq_rank_ratio"] if big else ""))


def bsc_trajectory():
    D, H, Hp, gamma, N, steps = 25, 10, 5, 3, 2000, 20
    np.random.seed(7)
    model = BSC_ET(D, H, Hp, gamma)
    params_gt = {"W": 10 * generate_bars_dict(H), "pi": 2. / H, "sigma": 1.0}
    data = model.generate_data(params_gt, N)
    np.random.seed(11)
    init = model.standard_init(data)
    anneal = LinearAnnealing(steps)
    anneal["T"] = [(0, 2.), (.7, 1.)]
    anneal["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    anneal["anneal_prior"] = False
    Capture.rows.clear()
    em = EM(model=model, anneal=anneal)
    em.data = {"y": data["y"].copy()}
    em.lparams = {k: np.array(v, copy=True) for k, v in init.items()}
    Ws, pis, sigmas = [], [], []
    while not anneal.finished:                      # EM.run body (em/__init__.py:163-178), recording each step
        new = model.step(anneal, em.lparams, em.data)
        anneal.next(model.gain(em.lparams, new))
        em.lparams = new
        Ws.append(new["W"].copy()); pis.append(new["pi"]); sigmas.append(new["sigma"])
    np.savez_compressed(os.path.join(HERE, "bsc_traj_c1.npz"),
                        D=D, H=H, Hprime=Hp, gamma=gamma, steps=steps, y=data["y"], s=data["s"],
                        W_gt=params_gt["W"], W0=init["W"], pi0=init["pi"], sigma0=init["sigma"],
                        W=np.stack(Ws), pi=np.array(pis), sigma=np.array(sigmas),
                        L=np.array(Capture.rows["L"]), N_use=np.array(Capture.rows["N_use"]))
    print("bsc_traj_c1: L[0]=%.6f L[-1]=%.6f pi=%.5f sigma=%.5f N_use[-1]=%d" % (
        Capture.rows["L"][0], Capture.rows["L"][-1], pis[-1], sigmas[-1], Capture.rows["N_use"][-1]))


def bsc_init():
    D, H, Hp, gamma, N = 25, 10, 5, 3, 64
    model = BSC_ET(D, H, Hp, gamma)
    params_gt = {"W": 10 * generate_bars_dict(H), "pi": 2. / H, "sigma": 1.0}
    np.random.seed(3)
    data = model.generate_data(params_gt, N)
    np.random.seed(5)
    init = model.standard_init(data)
    np.savez_compressed(os.path.join(HERE, "bsc_init_c1.npz"), D=D, H=H, N=N, seed_data=3, seed_init=5,
                        W_gt=params_gt["W"], pi_gt=params_gt["pi"], sigma_gt=params_gt["sigma"],
                        y=data["y"], s=data["s"], W0=init["W"], pi0=init["pi"], sigma0=init["sigma"])
    print("bsc_init_c1 ok")


def anneal_tracks():
    steps = 50
    a = LinearAnnealing(steps)
    a["T"] = [(0, 2.), (.7, 1.)]
    a["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    a["anneal_prior"] = False
    a["W_noise"] = [(0, 0.5), (-10, 0.0)]
    names = sorted(a.as_dict().keys())
    rows = []
    while not a.finished:
        d = a.as_dict()
        rows.append([float(d[k]) for k in names] + [float(a["not_a_param"])])
        a.next()
    np.savez_compressed(os.path.join(HERE, "anneal_tracks.npz"), steps=steps, names=np.array(names + ["not_a_param"]),
                        values=np.array(rows))
    print("anneal_tracks ok", names)


def schedule_trajectory(name, kind, D, H, Hp, gamma, N, seed, steps=50):
    """The reference's own annealing schedule (bars-learning.py:77-80) for ``steps`` EM steps at a BASELINE configuration's
    DIMENSIONS: per-step free energy, N_use and scalar / vector parameters, the matrices after two of the steps.  Inputs: schedule_inputs.py."""
    import time
    from schedule_inputs import schedule_inputs
    y, p0 = schedule_inputs(kind, D, H, N, seed)
    if kind == "gsc":
        model = GSC(D, H, Hp, gamma, sigma_sq_type="scalar")
    elif kind == "dsc":
        from schedule_inputs import DSC_STATES
        model = DSC_ET(D, H, Hp, gamma, states=DSC_STATES.copy())
    elif kind == "tsc":
        model = _make_tsc(D, H, Hp, gamma)
    else:
        model = {"bsc": BSC_ET, "mca": MCA_ET, "mmca": MMCA_ET}[kind](D, H, Hp, gamma)
    anneal = LinearAnnealing(steps)
    anneal["T"] = [(0, 2.), (.7, 1.)]
    anneal["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    anneal["anneal_prior"] = False
    Capture.rows.clear()
    data = {"y": y.copy()}
    lparams = {k: np.array(v, copy=True) for k, v in p0.items()}
    keep = (steps // 5, steps - 1)          # (W, and GSC's psi_sq, after these steps only: megabytes each)
    Ws, scal = [], {k: [] for k in p0 if k != "W"}
    t0 = time.time()
    it = 0
    while not anneal.finished:                      # EM.run body (em/__init__.py:163-178)
        new = model.step(anneal, lparams, data)
        anneal.next(model.gain(lparams, new))
        lparams = new
        if it in keep:
            Ws.append(np.array(new["W"], copy=True))
        for k in scal:
            if np.ndim(new[k]) < 2 or it in keep:
                scal[k].append(np.array(new[k], copy=True))
        it += 1
    out = {"kind": kind, "D": D, "H": H, "Hprime": Hp, "gamma": gamma, "N": N, "seed": seed, "steps": steps,
           "keep": np.array(keep), "W": np.stack(Ws), "L": np.array(Capture.rows.get("L", [])),
           "N_use": np.array(Capture.rows.get("N_use", [N] * steps))}
    for k, v in scal.items():
        out[k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, "schedule_%s.npz" % name), **out)
    print("schedule_%s: %d steps in %.0f s, L[0]=%.6f L[-1]=%.6f N_use[-1]=%d" % (
        name, steps, time.time() - t0, out["L"][0] if len(out["L"]) else np.nan, out["L"][-1] if len(out["L"]) else np.nan,
        out["N_use"][-1]))


def noise_trajectory(kind, D, H, Hp, gamma, N, seed, steps=12):
    """EM.run with parameter noise and partial data (noisify_params em/__init__.py:63-107, select_partial_data
    camodels/__init__.py:124-152): both draw from NumPy's global stream, seeded here -- a drop-in that consumes the stream in the
    same order reproduces the trajectory."""
    from schedule_inputs import schedule_inputs
    y, p0 = schedule_inputs(kind, D, H, N, seed)
    if kind == "gsc":
        model = GSC(D, H, Hp, gamma, sigma_sq_type="scalar")
    elif kind == "dsc":
        from schedule_inputs import DSC_STATES
        model = DSC_ET(D, H, Hp, gamma, states=DSC_STATES.copy())
    elif kind == "tsc":
        model = _make_tsc(D, H, Hp, gamma)
    else:
        model = {"bsc": BSC_ET, "mca": MCA_ET, "mmca": MMCA_ET}[kind](D, H, Hp, gamma)
    anneal = LinearAnnealing(steps)
    anneal["T"] = [(0, 1.6), (.7, 1.)]
    anneal["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    anneal["anneal_prior"] = False
    anneal["partial"] = [(0, .6), (.5, 1.)]
    anneal["W_noise"] = [(0, .08), (.6, 0.)]
    anneal["pi_noise"] = [(0, .002), (.6, 0.)]
    anneal["sigma_noise" if kind != "gsc" else "sigma_sq_noise"] = [(0, .03), (.6, 0.)]
    Capture.rows.clear()
    np.random.seed(1000 + seed)
    data = {"y": y.copy()}
    lparams = {k: np.array(v, copy=True) for k, v in p0.items()}
    hist = {k: [] for k in p0}
    while not anneal.finished:
        new = model.step(anneal, lparams, data)
        anneal.next(model.gain(lparams, new))
        lparams = new
        for k in hist:
            hist[k].append(np.array(new[k], copy=True))
    out = {"kind": kind, "D": D, "H": H, "Hprime": Hp, "gamma": gamma, "N": N, "seed": seed, "steps": steps,
           "N_use": np.array(Capture.rows.get("N_use", [])), "L": np.array(Capture.rows.get("L", []))}
    for k, v in hist.items():
        out[k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, "noise_traj_%s.npz" % kind), **out)
    print("noise_traj_%s: %d steps, N_use %s" % (kind, steps, out["N_use"][:6]))


def standard_init_cases():
    """``standard_init`` of every model (camodels/__init__.py:196-235; gsc_et.py:59-114, dsc_et.py's own) from a seeded NumPy
    stream, inputs from schedule_inputs.py: a drop-in that draws in the same order returns the same parameters."""
    from schedule_inputs import schedule_inputs, DSC_STATES
    out = {}
    for tag, kind, mk in (("mca", "mca", lambda: MCA_ET(40, 16, 5, 3)), ("mmca", "mmca", lambda: MMCA_ET(40, 16, 5, 3)),
                          ("dsc", "dsc", lambda: DSC_ET(40, 16, 5, 3, states=DSC_STATES.copy())),
                          ("tsc", "tsc", lambda: _make_tsc(40, 16, 5, 3)),
                          ("gsc_scalar", "gsc", lambda: GSC(40, 16, 5, 3, sigma_sq_type="scalar")),
                          ("gsc_diagonal", "gsc", lambda: GSC(40, 16, 5, 3, sigma_sq_type="diagonal")),
                          ("gsc_full", "gsc", lambda: GSC(40, 16, 5, 3, sigma_sq_type="full"))):
        y, _ = schedule_inputs(kind, 40, 16, 300, 500)
        np.random.seed(77)
        init = mk().standard_init({"y": y.copy()})
        for k, v in init.items():
            out["%s_%s" % (tag, k)] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "standard_init_all.npz"), **out)
    print("standard_init_all:", sorted(out))


def generate_data_cases():
    """``generate_data`` of every model from a seeded NumPy stream (the reference draws latents and noise per datapoint)."""
    from schedule_inputs import schedule_inputs, DSC_STATES
    out = {}
    for tag, kind, mk in (("bsc", "bsc", lambda: BSC_ET(24, 10, 4, 3)), ("mca", "mca", lambda: MCA_ET(24, 10, 4, 3)),
                          ("mmca", "mmca", lambda: MMCA_ET(24, 10, 4, 3)),
                          ("dsc", "dsc", lambda: DSC_ET(24, 10, 4, 3, states=DSC_STATES.copy())),
                          ("gsc", "gsc", lambda: GSC(24, 10, 4, 3, sigma_sq_type="scalar"))):
        _, p0 = schedule_inputs(kind, 24, 10, 8, 600)
        np.random.seed(91)
        data = mk().generate_data({k: np.array(v, copy=True) for k, v in p0.items()}, 40)
        for k, v in data.items():
            out["%s_%s" % (tag, k)] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "generate_data_all.npz"), **out)
    print("generate_data_all:", sorted(out))


def main(only=None, cases=None):
    """``only``: regenerate just the fixtures whose maker's name starts with this prefix (e.g. ``mmca``);
    ``cases``: of those, just the named step cases (e.g. ``c2_plain,c2_cut``)."""
    want = lambda fn: only is None or fn.__name__.startswith(only)
    g = globals()
    for _n in ("bsc_step_case", "gsc_step_case", "mca_step_case", "mmca_step_case", "dsc_step_case", "dsc_inference_case", "tsc_step_case", "tsc_inference_case", "bsc_inference_case",
               "mca_inference_case", "mmca_inference_case", "gsc_inference_case", "gsc_posterior_hprime_case", "bsc_trajectory",
               "bsc_init", "anneal_tracks", "schedule_trajectory", "inference_big_case", "noise_trajectory", "standard_init_cases", "generate_data_cases"):
        if not want(g[_n]):
            g[_n] = (lambda *a, **k: None)
    if cases:
        for _n in ("bsc_step_case", "gsc_step_case", "mca_step_case", "mmca_step_case", "dsc_step_case", "tsc_step_case",
                   "schedule_trajectory", "noise_trajectory"):
            g[_n] = (lambda fn: (lambda name, *a, **k: fn(name, *a, **k) if name in cases else None))(g[_n])
    # BASELINE config-1 dims (D=25 H=10 H'=5 gamma=3)
    bsc_step_case("c1_plain", 25, 10, 5, 3, 400, seed=1, T=1.0, Ncut=0.0, anneal_prior=False, bars=True)
    bsc_step_case("c1_anneal_cut", 25, 10, 5, 3, 333, seed=2, T=1.7, Ncut=0.6, anneal_prior=True, bars=True)
    bsc_step_case("c1_fullcut", 25, 10, 5, 3, 257, seed=3, T=1.25, Ncut=1.0, anneal_prior=False, bars=True)
    # other shapes: H not a multiple of 64, H' = gamma (2^H' - H' - 1 multi states), learned mu
    bsc_step_case("h32", 64, 32, 6, 3, 200, seed=4, T=1.0, Ncut=0.0, anneal_prior=False)
    bsc_step_case("h100_cut", 48, 100, 7, 4, 150, seed=5, T=1.4, Ncut=0.8, anneal_prior=False)
    bsc_step_case("gamma_eq_hp", 30, 12, 4, 4, 120, seed=6, T=1.0, Ncut=0.0, anneal_prior=True)
    bsc_step_case("mu", 40, 16, 5, 3, 180, seed=8, T=1.1, Ncut=0.5, anneal_prior=False, mu=True,
                  to_learn=("W", "pi", "sigma", "mu"))
    bsc_step_case("h256", 96, 256, 8, 4, 96, seed=9, T=1.0, Ncut=0.0, anneal_prior=False, amp=0.5)
    # BASELINE config-2 dims (D=1024 H=256 H'=8 gamma=4, K=411); sigma_gt = 2 keeps the reference's
    # un-stabilised exp(logpj) sums above the underflow threshold for every datapoint of the sample
    bsc_step_case("c2_plain", 1024, 256, 8, 4, 128, seed=12, T=1.0, Ncut=0.0, anneal_prior=False, sigma_gt=2.0, big=True)
    bsc_step_case("c2_cut", 1024, 256, 8, 4, 128, seed=13, T=1.2, Ncut=0.7, anneal_prior=False, sigma_gt=2.0, big=True)
    # ... and with enough datapoints for a full-rank Wq (N = 512 > H and a seed for which every latent is active in at least one datapoint): W_new
    # itself is pinned at config-2 dimensions
    bsc_step_case("c2_fullrank", 1024, 256, 8, 4, 512, seed=131, T=1.0, Ncut=0.0, anneal_prior=False, sigma_gt=2.0, big=True)
    gsc_step_case("small", 16, 8, 4, 3, 200, seed=31, T=1.0)
    gsc_step_case("small_T", 16, 8, 4, 3, 151, seed=32, T=1.5, full_psi=True)
    gsc_step_case("h24", 40, 24, 5, 3, 120, seed=33, T=1.0, full_psi=True)
    gsc_step_case("g4", 30, 12, 5, 4, 100, seed=34, T=1.2)
    gsc_step_case("h128", 64, 128, 6, 3, 64, seed=35, T=1.0)
    gsc_step_case("c4", 256, 128, 6, 3, 96, seed=40, T=1.0)           # BASELINE config-4 dims
    gsc_step_case("diag", 16, 8, 4, 3, 200, seed=36, T=1.0, sigma_type="diagonal")
    gsc_step_case("diag_T", 40, 24, 5, 3, 120, seed=37, T=1.4, full_psi=True, sigma_type="diagonal")
    gsc_step_case("full", 16, 8, 4, 3, 200, seed=38, T=1.0, sigma_type="full")
    gsc_step_case("full_T", 40, 24, 5, 3, 120, seed=39, T=1.3, full_psi=True, sigma_type="full")
    # second / third EM step: non-symmetric psi_sq in, non-symmetric sum xpt_szsz inverted (round 4)
    gsc_step_case("step2", 40, 24, 5, 3, 150, seed=41, T=1.0, presteps=1)
    gsc_step_case("step3_T", 30, 12, 5, 4, 120, seed=42, T=1.2, full_psi=True, presteps=2)
    gsc_step_case("step2_c4", 256, 128, 6, 3, 160, seed=43, T=1.0, presteps=1)           # BASELINE config-4 dims
    gsc_step_case("step2_diag", 40, 24, 5, 3, 150, seed=44, T=1.0, sigma_type="diagonal", presteps=1)
    gsc_step_case("step2_full", 40, 24, 5, 3, 150, seed=45, T=1.1, sigma_type="full", presteps=1)
    mca_step_case("small", 16, 8, 4, 3, 300, seed=21, T=1.0, Ncut=0.0)
    mca_step_case("small_cut", 16, 8, 4, 3, 257, seed=22, T=1.4, Ncut=0.5)
    mca_step_case("bars", 25, 10, 5, 3, 300, seed=23, T=1.0, Ncut=1.0, bars=True)
    mca_step_case("h40", 48, 40, 6, 3, 150, seed=24, T=2.0, Ncut=0.7)
    mca_step_case("h128", 64, 128, 8, 3, 96, seed=25, T=1.0, Ncut=0.0)
    mca_step_case("c5", 256, 128, 8, 3, 96, seed=26, T=1.0, Ncut=0.0)    # BASELINE config-5 dims
    mca_step_case("c5_cut", 256, 128, 8, 3, 80, seed=27, T=1.3, Ncut=0.6)
    tsc_step_case("small", 16, 8, 4, 3, 300, seed=81, T=1.0, Ncut=0.0, anneal_prior=False)
    tsc_step_case("cut", 24, 12, 5, 3, 257, seed=82, T=1.4, Ncut=0.6, anneal_prior=True)
    tsc_step_case("g2", 30, 20, 6, 2, 200, seed=83, T=1.0, Ncut=1.0, anneal_prior=False)
    tsc_step_case("h64", 48, 64, 6, 3, 120, seed=84, T=1.2, Ncut=0.0, anneal_prior=False)
    tsc_inference_case()
    bsc_inference_case()
    dsc_inference_case()
    mca_inference_case()
    mmca_inference_case()
    gsc_inference_case()
    gsc_posterior_hprime_case()
    inference_big_case("bsc", 64, 40, 6, 3, 3000, seed=301)
    inference_big_case("mca", 64, 40, 6, 3, 2000, seed=302)
    inference_big_case("gsc", 48, 24, 5, 3, 1500, seed=303)
    bsc_trajectory()
    # the reference's 50-step schedule at the dimensions of BASELINE configs 2, 4 and 5 (round 6)
    schedule_trajectory("bsc_c2", "bsc", 1024, 256, 8, 4, 2500, seed=201)
    schedule_trajectory("gsc_c4", "gsc", 256, 128, 6, 3, 200, seed=202)
    schedule_trajectory("mca_c5", "mca", 256, 128, 8, 3, 200, seed=203)
    schedule_trajectory("mmca", "mmca", 256, 128, 8, 3, 200, seed=204)
    schedule_trajectory("dsc", "dsc", 128, 64, 6, 3, 800, seed=205)
    schedule_trajectory("tsc", "tsc", 128, 64, 6, 3, 800, seed=206)
    noise_trajectory("bsc", 40, 16, 5, 3, 600, seed=401)
    noise_trajectory("mca", 40, 16, 5, 3, 500, seed=402)
    noise_trajectory("gsc", 30, 12, 4, 3, 400, seed=403)
    noise_trajectory("mmca", 40, 16, 5, 3, 500, seed=404)
    noise_trajectory("dsc", 40, 16, 5, 3, 600, seed=405)
    noise_trajectory("tsc", 40, 16, 5, 3, 600, seed=406)
    standard_init_cases()
    generate_data_cases()
    bsc_init()
    anneal_tracks()
    mmca_step_case("small", 16, 8, 4, 3, 300, seed=51, T=1.0, Ncut=0.0)
    mmca_step_case("small_cut", 16, 8, 4, 3, 257, seed=52, T=1.5, Ncut=0.5)
    mmca_step_case("h40", 48, 40, 6, 3, 150, seed=53, T=2.5, Ncut=0.7)
    mmca_step_case("h128", 64, 128, 8, 3, 96, seed=54, T=1.0, Ncut=0.0)
    dsc_step_case("ternary", 16, 8, 4, 3, 300, seed=61, T=1.0, Ncut=0.0, anneal_prior=False,
                  states=[-1., 0., 1.], pi_gt=[0.1, 0.8, 0.1])
    dsc_step_case("ternary_cut", 20, 10, 5, 3, 257, seed=62, T=1.5, Ncut=0.6, anneal_prior=True,
                  states=[-1., 0., 1.], pi_gt=[0.08, 0.8, 0.12])
    dsc_step_case("k4", 24, 12, 4, 2, 200, seed=63, T=1.2, Ncut=0.0, anneal_prior=False,
                  states=[0., 1., 2., 3.], pi_gt=[0.82, 0.1, 0.05, 0.03])
    dsc_step_case("binary", 25, 10, 5, 3, 200, seed=64, T=1.0, Ncut=1.0, anneal_prior=False,
                  states=[0., 1.], pi_gt=[0.8, 0.2])
    dsc_step_case("h64", 48, 64, 6, 3, 120, seed=65, T=1.0, Ncut=0.0, anneal_prior=False,
                  states=[-2., -1., 0., 1., 2.], pi_gt=[0.02, 0.03, 0.9, 0.03, 0.02])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None, sys.argv[2].split(",") if len(sys.argv) > 2 else None)
