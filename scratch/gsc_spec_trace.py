"""GSC config 4: per-step wall-clock (each step ends in a blocking download) and speculation hits over a long loop --
are the slow steps of bench.py's 20-step window misses of the speculative E-step, or something else?"""
import os, sys, time, gc
import numpy as np, torch
from prosper_amd.em.camodels.gsc_et import GSC
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
Dm, Hm, N = 256, 128, 200_000
g = torch.Generator(device=dev).manual_seed(3); rng = np.random.RandomState(3)
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, Hm, generator=g, device=dev) < 2.0 / Hm).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, Hm, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, Dm, generator=g, device=dev, dtype=torch.float64)
p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(Dm, Hm)), "pi": np.full(Hm, 2.0 / Hm), "mu": np.full(Hm, 1.4),
     "psi_sq": np.eye(Hm) * 1.1, "sigma_sq": 1.2}
m = GSC(Dm, Hm, 6, 3, 'scalar')
gc.collect(); gc.disable()
ts, hits = [], []
for it in range(700):
    h0 = m.spec_hits
    t = time.perf_counter()
    p = m.step(An(T=1.0), p, {"y": Y})
    ts.append((time.perf_counter() - t) * 1e3); hits.append(m.spec_hits - h0)
ts, hits = np.array(ts), np.array(hits)
print("first 12 steps ms:", np.round(ts[:12], 2), "hits", hits[:12])
for lo in range(0, 700, 100):
    w = ts[lo:lo + 100]
    print("steps %3d-%3d  mean %.3f  median %.3f  p90 %.3f  max %.3f  hits %d  fallbacks %d" % (lo, lo + 99, w.mean(), np.median(w), np.percentile(w, 90), w.max(), hits[lo:lo + 100].sum(), m.inverse_fallbacks))
miss = np.nonzero(hits[50:] == 0)[0] + 50
print("misses after step 50:", miss[:40], "their ms:", np.round(ts[miss[:40]], 2))
slow = np.nonzero(ts[50:] > 1.4)[0] + 50
print("steps > 1.4 ms after step 50:", slow[:40], np.round(ts[slow[:40]], 2))
