"""MMCA at config-5 dimensions (signed data): EM iteration, flat and on a truncation step, per-kernel times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.mmca_et import MMCA_ET
from prosper_amd.em.camodels._device import KernelTimer
dev = torch.device("cuda", 0)
Dm, Hm, N = 256, 128, 100000
g = torch.Generator(device=dev).manual_seed(3)
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64) * 2
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    u = torch.rand(25000, Hm, generator=g, device=dev)
    S = (u < 1.0 / Hm).to(torch.float64) - (u > 1 - 1.0 / Hm).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(25000, Dm, generator=g, device=dev, dtype=torch.float64)
W0 = (W_gt + 0.1 * torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)).cpu().numpy()
class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)
for T, ncut in ((1.0, 0.0), (1.5, 0.0), (1.0, 1.0)):
    m = MMCA_ET(Dm, Hm, 8, 3)
    p = {"W": W0, "pi": 2.0 / Hm, "sigma": 1.0}
    an = An(T=T, Ncut_factor=ncut)
    for _ in range(10):
        p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
    m.timer = kt = KernelTimer()
    for _ in range(2):
        p = m.step(an, p, {"y": Y})
    m.timer = None
    ks = kt.summary()
    print("T %.2f rho %.3f Ncut %.1f: %.3f ms/step  %s" % (T, m._rho(T), ncut, ms, {k: (v[0] // 2, round(v[1], 3)) for k, v in sorted(ks.items())}))
