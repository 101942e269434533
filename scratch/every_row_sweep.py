"""Exploratory: every row of a large shard against the vectorised oracles at annealing points OTHER than T = 1
(what tests/test_*_gpu.py::*full_shard* do at T = 1).  Prints the worst log-joint / moment deviation in units of the
tolerance (rtol 1e-10 / 1e-9, atol 1e-9 / 1e-12)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["gsc", "mca", "bsc", "dsc", "tsc", "mmca"]

def report(name, worst, extra=""):
    print("%-28s worst = %.3g x tolerance %s" % (name, worst, extra), flush=True)

if "gsc" in which:
    from oracle import gsc_oracle as G
    from prosper_amd.em.camodels.gsc_et import GSC
    D, H, Hp, gamma, N = 256, 128, 6, 3, 200_000
    for T in (2.0, 1.3):
        gen = torch.Generator(device=dev).manual_seed(44)
        W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
        Y = torch.empty(N, D, dtype=torch.float64, device=dev)
        for lo in range(0, N, 50_000):
            S = (torch.rand(50_000, H, generator=gen, device=dev) < 2.0 / H).to(torch.float64)
            Z = S * (1.5 + torch.randn(50_000, H, generator=gen, device=dev, dtype=torch.float64))
            Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, D, generator=gen, device=dev, dtype=torch.float64)
        rng = np.random.RandomState(44)
        p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4), "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
        m = GSC(D, H, Hp, gamma, 'scalar')
        cp = lambda q: {k: np.array(v, copy=True) for k, v in q.items()}
        data = m.select_Hprimes(cp(p), {"y": Y})
        ss = m.E_step(An(T=T), cp(p), data)
        model = G.make_model(D, H, Hp, gamma)
        worst = 0.0
        for lo in range(0, N, 2000):
            y_m = Y[lo:lo + 2000].cpu().numpy()
            c_m = data["candidates"].tensor[lo:lo + 2000].cpu().numpy().astype(np.int64)
            suff = G.e_step(G.Anneal(T=T), model, p, y_m, c_m)
            for k in ("xpt_s", "xpt_sz"):
                got = ss[k].tensor[lo:lo + 2000].cpu().numpy()
                worst = max(worst, float(np.max(np.abs(got - suff[k]) / (1e-12 + 1e-9 * np.abs(suff[k])))))
        report("GSC c4 T=%.1f" % T, worst)
        del Y, m, data, ss

if "mca" in which or "mmca" in which:
    for kind in [k for k in ("mca", "mmca") if k in which]:
        if kind == "mca":
            from oracle import mca_oracle as M
            from prosper_amd.em.camodels.mca_et import MCA_ET as cls
        else:
            from oracle import mmca_oracle as M
            from prosper_amd.em.camodels.mmca_et import MMCA_ET as cls
        D, H, Hp, gamma, N = 256, 128, 8, 3, 50_000
        rng = np.random.RandomState(9)
        W_gt = (np.abs(rng.normal(size=(D, H))) * 2 + 0.1) if kind == "mca" else rng.normal(size=(D, H)) * 3.0
        y = np.empty((N, D))
        for lo in range(0, N, 10_000):
            s = rng.random_sample((10_000, H)) < 2.0 / H
            if kind == "mca":
                y[lo:lo + 10_000] = np.where(s[:, None, :], W_gt[None, :, :], 0).max(axis=2)
            else:
                y[lo:lo + 10_000] = M.generate_from_hidden(W_gt, s)
        y += rng.normal(size=(N, D))
        for T in (2.0, 1.4, 1.1):
            m = cls(D, H, Hp, gamma)
            p = m.check_params({"W": W_gt * (1 + 0.03 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.2 / H, "sigma": 1.1})
            model = M.make_model(D, H, Hp, gamma)
            data = m.select_Hprimes(p, {"y": y})
            ss = m.E_step(An(T=T), p, data)
            cand, lp = np.asarray(data["candidates"]), ss["logpj"]
            worst = 0.0
            for lo in range(0, N, 2048):
                ref = M.e_step_vec(M.Anneal(T=T), p["W"], p["pi"], p["sigma"], y[lo:lo + 2048], cand[lo:lo + 2048], model["SM"], model["state_abs"])
                got = np.asarray(lp[lo:lo + 2048])
                worst = max(worst, float(np.max(np.abs(got - ref) / (1e-9 + 1e-10 * np.abs(ref)))))
            report("%s c5 dims T=%.1f" % (kind.upper(), T), worst)
            del m, data, ss

if "bsc" in which:
    from oracle import bsc_oracle as O
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 1024, 256, 8, 4, 100_000
    rs0 = np.random.RandomState(0)
    W_gt_h = rs0.randn(D, H)
    W0 = np.ascontiguousarray((W_gt_h + 0.1 * rs0.randn(D, H)).T).T
    rs = np.random.RandomState(0)
    y = np.empty((N, D))
    for lo in range(0, N, 25_000):
        S = (rs.random_sample((25_000, H)) < 4.0 / H).astype(np.float64)
        y[lo:lo + 25_000] = rs.normal(size=(25_000, D)) + S @ W_gt_h.T
    Y = torch.from_numpy(y).to(dev)
    om = O.make_model(D, H, Hp, gamma)
    for T, ap in ((1.5, True), (2.0, False)):
        params = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
        m = BSC_ET(D, H, Hp, gamma)
        d = m.select_Hprimes(dict(params), {"y": Y})
        ss = m.E_step(An(T=T, anneal_prior=ap), dict(params), d)
        cand, lp = d["candidates"].tensor, ss["logpj"].tensor
        worst, bad = 0.0, 0
        for lo in range(0, N, 8192):
            y_c = y[lo:lo + 8192]
            c_ref = O.select_hprimes_vec(W0, y_c, Hp)
            bad += int((cand[lo:lo + 8192].cpu().numpy() != c_ref).any(axis=1).sum())
            ref = O.e_step_vec(O.Anneal(T=T, anneal_prior=ap), W0, params["pi"], params["sigma"], params["mu"], y_c, c_ref, om["SM"], om["state_abs"])
            got = lp[lo:lo + 8192].cpu().numpy()
            worst = max(worst, float(np.max(np.abs(got - ref) / (1e-9 + 1e-10 * np.abs(ref)))))
        report("BSC c2 T=%.1f prior=%s" % (T, ap), worst, "(%d rows with other candidates)" % bad)
        del m, d, ss

for kind in [k for k in ("dsc", "tsc") if k in which]:
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    D, H, Hp, gamma, N = 256, 128, 6, 3, 70000
    rng = np.random.RandomState(5)
    W_gt = 2.0 * rng.normal(size=(D, H))
    u = rng.random_sample((N, H))
    y = ((u < 1.0 / H).astype(float) - (u > 1 - 1.0 / H)) @ W_gt.T + rng.normal(size=(N, D))
    W0 = W_gt + 0.1 * rng.normal(size=(D, H))
    states = np.array([-1., 0., 1.])
    for T, ap in ((1.7, True), (1.2, False)):
        if kind == "dsc":
            from oracle import dsc_oracle as M
            m = DSC_ET(D, H, Hp, gamma, states=states)
            pi = np.array([1.0 / H, 1 - 2.0 / H, 1.0 / H])
            om = M.make_model(D, H, Hp, gamma, states)
        else:
            from oracle import tsc_oracle as M
            m = TSC_ET(D, H, Hp, gamma)
            pi = 2.0 / H
            om = M.make_model(D, H, Hp, gamma)
        params = {"W": W0, "pi": pi, "sigma": 1.0}
        data = m.select_Hprimes(params, {"y": y})
        ss = m.E_step(An(T=T, anneal_prior=ap), params, data)
        cand_all, lp_all = np.asarray(data["candidates"]), ss["logpj"]
        worst = 0.0
        oan = M.Anneal(T=T, Ncut_factor=0.0, anneal_prior=ap)
        for lo in range(0, N, 4096):
            ref = M.e_step_vec(oan, om, W0, pi, 1.0, y[lo:lo + 4096], cand_all[lo:lo + 4096])
            got = np.asarray(lp_all[lo:lo + 4096])
            worst = max(worst, float(np.max(np.abs(got - ref) / (1e-9 + 1e-10 * np.abs(ref)))))
        report("%s T=%.1f prior=%s" % (kind.upper(), T, ap), worst)
        del m, data, ss
