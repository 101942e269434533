import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
from prosper_amd.em.camodels.bsc_et import BSC_ET
class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)
D, H, Hp, gamma, N = [int(x) for x in sys.argv[1:6]]
rng = np.random.RandomState(N + H)
W_gt = rng.normal(size=(D, H))
y = (rng.random_sample((N, H)) < 3.0 / H) @ W_gt.T + rng.normal(size=(N, D))
params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.05}
res = {}
for fuse in (True, False):
    m = BSC_ET(D, H, Hp, gamma); m.fuse_mstats = fuse
    new = m.step(An(T=1.1), dict(params), {"y": y})
    res[fuse] = (m._ws["stats"].cpu().numpy().copy(), m._ws["expect"].cpu().numpy().copy())
lib = _lib.load()
o = [0, lib.pm_bsc_stats_offset_wq(H, D), lib.pm_bsc_stats_offset_qdiag(H, D), lib.pm_bsc_stats_offset_mus(H, D), lib.pm_bsc_stats_offset_scalars(H, D), len(res[True][0])]
names = ["Wp", "Wq", "qdiag", "mus", "scalars"]
for k in range(5):
    a, b = res[True][0][o[k]:o[k+1]], res[False][0][o[k]:o[k+1]]
    print(names[k], "nan:", np.isnan(a).sum(), "of", a.size, "maxdiff", np.nanmax(np.abs(a - b)) if a.size else 0)
    if names[k] in ("scalars",): print("  fused", a[:12], "\n  ref  ", b[:12])
ea, eb = res[True][1], res[False][1]
print("expect nan", np.isnan(ea).sum(), "rows with nan", np.unique(np.argwhere(np.isnan(ea))[:, 0])[:20], "maxdiff", np.nanmax(np.abs(ea - eb)))
