import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden
from prosper_amd.em.camodels.bsc_et import BSC_ET
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
g = golden("bsc_step_c2_cut.npz")
res = {}
for tile in ("4", "8"):
    m = BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    m.fused_tile = tile
    an = An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
    params = {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}
    d = m.select_Hprimes(params, {"y": g["y"]})
    ss = m.E_step(an, params, d)
    res[tile] = (np.asarray(ss["logpj"]), ss["logpj"].lse.cpu().numpy())
l4, l8 = res["4"][1], res["8"][1]
ref = np.log(np.exp(res["4"][0] - res["4"][0].max(1, keepdims=True)).sum(1)) + res["4"][0].max(1)
bad = np.where(~np.isclose(l4, l8, rtol=1e-12, atol=0))[0]
print("T", float(g["T"]), "N", l4.shape, "bad rows", bad)
for b in bad[:16]:
    print(b, l4[b], l8[b], ref[b])
print("max rel diff vs ref tile4 %.3e tile8 %.3e" % (np.abs(l4 - ref).max(), np.abs(l8 - ref).max()))
