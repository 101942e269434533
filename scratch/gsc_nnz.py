"""How many entries of a GSC xpt_s row exceed a threshold (config 4, bench data, after k EM steps)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.gsc_et import GSC
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
Dm, Hm, N = 256, 128, 200_000
g = torch.Generator(device=dev).manual_seed(3)
rng = np.random.RandomState(3)
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, Hm, generator=g, device=dev) < 2.0 / Hm).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, Hm, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, Dm, generator=g, device=dev, dtype=torch.float64)
p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(Dm, Hm)), "pi": np.full(Hm, 2.0 / Hm),
     "mu": np.full(Hm, 1.4), "psi_sq": np.eye(Hm) * 1.1, "sigma_sq": 1.2}
m = GSC(Dm, Hm, 6, 3, 'scalar')
for T in (1.0, 2.0, 4.0):
    q = {k: np.array(v, copy=True) for k, v in p.items()}
    for step in range(0, 31):
        if step in (0, 1, 5, 30):
            d = m.select_Hprimes(q, {"y": Y})
            ss = m.E_step(An(T=T), q, d)
            xs = ss["xpt_s"].tensor
            for thr in (1e-30, 1e-20, 1e-12):
                cnt = (xs > thr).sum(1)
                print("T %.0f step %2d thr %.0e: mean %.2f max %d  frac>16 %.5f frac>12 %.5f  min colsum %.3e" % (
                    T, step, thr, cnt.double().mean().item(), cnt.max().item(), (cnt > 16).double().mean().item(),
                    (cnt > 12).double().mean().item(), xs.sum(0).min().item()), flush=True)
        q = m.step(An(T=T), q, {"y": Y})
