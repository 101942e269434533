import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import golden
from prosper_amd.em.camodels.bsc_et import BSC_ET
class An(dict):
    def __missing__(self, k): return 0.0
for case in sys.argv[1:]:
    g = golden(case)
    for path in ("fused", "rows16"):
        m = BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
        m.use_fused = path == "fused"
        an = An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
        params = {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}
        if bool(g["has_mu"]): params["mu"] = g["mu"].copy()
        d = m.select_Hprimes(params, {"y": g["y"]})
        ss = m.E_step(an, params, d)
        cand = np.asarray(d["candidates"]); lp = np.asarray(ss["logpj"])
        H = int(g["H"])
        bad = np.abs(lp - g["logpj"]) > 1e-9 + 1e-10 * np.abs(g["logpj"])
        print(case, path, "cand equal:", np.array_equal(cand, g["candidates"]), "bad logpj:", bad.sum(), "of", bad.size,
              "| null col bad", bad[:, 0].sum(), "singles bad", bad[:, 1:H + 1].sum(), "multi bad", bad[:, H + 1:].sum())
        if bad.any():
            r, c = np.argwhere(bad)[0]
            print("  first bad row", r, "col", c, "got", lp[r, c], "want", g["logpj"][r, c], "cols bad in that row:", np.where(bad[r])[0][:20])
            print("  rows bad:", np.unique(np.argwhere(bad)[:, 0])[:20], "lse finite:", np.isfinite(ss["logpj"].lse.cpu().numpy()).all())
