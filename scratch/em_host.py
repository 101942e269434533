"""Host-side profile of steady EM steps (BSC config 2): where the time between the M-step's download and the next
E-step launch goes."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.bsc_et import BSC_ET
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
an = An(T=1.0)
m = BSC_ET(D, H, Hp, g)
p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
data = {"y": Y}
for _ in range(5): p = m.step(an, p, data)
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
t = time.perf_counter()
for _ in range(20): p = m.step(an, p, data)
torch.cuda.synchronize()
print("ms/iter %.3f" % ((time.perf_counter() - t) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): p = m.step(an, p, data)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
