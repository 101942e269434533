"""Which rows of the config-4 shard differ from the oracle's posterior moments, and what their largest log-joint is."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import gsc_oracle as G
from prosper_amd.em.camodels.gsc_et import GSC
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
D, H, Hp, gamma, N = 256, 128, 6, 3, 200_000
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
ss = m.E_step(An(T=1.0), cp(p), data)
model = G.make_model(D, H, Hp, gamma)
rows = []
for lo in range(0, N, 2000):
    y_m = Y[lo:lo + 2000].cpu().numpy()
    c_m = data["candidates"].tensor[lo:lo + 2000].cpu().numpy().astype(np.int64)
    suff = G.e_step(G.Anneal(T=1.0), model, p, y_m, c_m)
    for k in ("xpt_s", "xpt_sz"):
        got = ss[k].tensor[lo:lo + 2000].cpu().numpy()
        r = np.max(np.abs(got - suff[k]) / (1e-12 + 1e-9 * np.abs(suff[k])), axis=1)
        for i in np.nonzero(r > 1.0)[0]:
            rows.append((lo + int(i), k, float(r[i])))
print(len(rows), "row/key pairs above tolerance; first:", rows[:6])
bad = sorted(set(r[0] for r in rows))[:8]
if bad:
    yb = Y[torch.tensor(bad, device=dev)].cpu().numpy()
    cb = data["candidates"].tensor[torch.tensor(bad, device=dev)].cpu().numpy().astype(np.int64)
    lpj = G.compute_lpj(model, p, yb, cb)
    lpj = lpj[0] if isinstance(lpj, tuple) else lpj
    print("max log-joint of those rows:", np.max(lpj, axis=1))
