"""MCA config 5 (D=256 H=128 H'=8 gamma=3, N=100k): EM iteration and kernel times against the temperature (rho = 1 / (1 - 1/T):
21 at T <= 1.05 -- the log/exp-free power --, any real value on an annealing ramp -- the table power) and Ncut_factor."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.mca_et import MCA_ET
from prosper_amd.em.camodels._device import KernelTimer

dev = torch.device("cuda", 0)
Dm, Hm, N = 256, 128, 100000
g = torch.Generator(device=dev).manual_seed(3)
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64).abs() * 2 + 0.1
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = torch.rand(25000, Hm, generator=g, device=dev) < 2.0 / Hm
    Wm = torch.where(S[:, None, :], W_gt[None, :, :].expand(25000, Dm, Hm), torch.zeros((), dtype=torch.float64, device=dev)).max(dim=2).values
    Y[lo:lo + 25000] = Wm + torch.randn(25000, Dm, generator=g, device=dev, dtype=torch.float64)
p0 = {"W": (W_gt * (1 + 0.1 * (2 * torch.rand(Dm, Hm, generator=g, device=dev, dtype=torch.float64) - 1))).cpu().numpy(), "pi": 2.0 / Hm, "sigma": 1.0}


class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)


for T, ncut in ((1.0, 0.0), (1.5, 0.0), (1.3, 0.0), (2.0, 0.0), (1.0, 1.0), (1.3, 0.5)):
    m = MCA_ET(Dm, Hm, 8, 3)
    p = dict(p0)
    an = An(T=T, Ncut_factor=ncut)
    for _ in range(15):
        p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(15):
        p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 15 * 1e3
    m.timer = kt = KernelTimer()
    for _ in range(2):
        p = m.step(an, p, {"y": Y})
    m.timer = None
    ks = kt.summary()
    print("T %.2f rho %.3f Ncut %.1f: %.3f ms/step  %s" % (T, m._rho(T), ncut, ms, {k: (v[0] // 2, round(v[1], 3)) for k, v in sorted(ks.items())}))
