"""BSC at config-2 dimensions through a hot start (T = 30 -> 1): the first steps overflow the non-zero lists (dense
product behind the device-side gate), the later ones take the sparse product; with and without the pipeline features the
trajectories must agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
from prosper_amd.em.camodels.bsc_et import BSC_ET
dev = torch.device("cuda", 0)
class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)
D, H, Hp, g, N = 1024, 256, 8, 4, 40000
gen = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
S = (torch.rand(N, H, generator=gen, device=dev) < 4.0 / H).to(torch.float64)
Y = S @ W_gt.t() + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
W0 = (W_gt + 0.3 * torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)).cpu().numpy()
res = {}
for flags in (True, False):
    m = BSC_ET(D, H, Hp, g)
    m.speculate_estep = m.fuse_mstats = m.sparse_wp = flags
    p = {"W": W0.copy(), "pi": 2.0 / H, "sigma": 2.0}
    over = []
    for it in range(40):
        T = 30.0 if it < 6 else max(1.0, 30.0 - 29.0 * (it - 6) / 14.0)
        p = m.step(An(T=T), p, {"y": Y})
        if flags:
            o_sc = _lib.load().pm_bsc_stats_offset_scalars(H, D)
            over.append(int(m._ws["stats"][o_sc + 3].item()) if "stats" in m._ws else -1)
    res[flags] = p
    if flags:
        print("rows with overflowing lists per step (as left in the statistics buffer):", over)
for k in ("W", "pi", "sigma"):
    a, b = np.asarray(res[True][k]), np.asarray(res[False][k])
    print(k, "max rel diff %.3g" % (np.abs(a - b).max() / np.abs(b).max()), "finite", bool(np.isfinite(a).all()))
