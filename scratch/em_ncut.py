"""BSC EM iteration at config 2 with and without data truncation (the reference's schedules end with Ncut_factor = 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.bsc_et import BSC_ET
from prosper_amd.em.camodels._device import KernelTimer
D, H, Hp, g, N = 1024, 256, 8, 4, 200000
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(25000, H, generator=gen, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(25000, D, generator=gen, device=dev, dtype=torch.float64)
W0 = (W_gt + 0.1 * torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)).cpu().numpy()
class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)
for ncut in (0.0, 1.0):
    m = BSC_ET(D, H, Hp, g)
    p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0}
    an = An(T=1.0, Ncut_factor=ncut)
    for _ in range(15): p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 30 * 1e3
    m.timer = KernelTimer()
    for _ in range(5): p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize()
    print("Ncut_factor %.1f: EM iteration %.3f ms, spec hits %d" % (ncut, ms, m.spec_hits), {k: round(v[1], 3) for k, v in m.timer.summary().items()})
