import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from prosper_amd.em.camodels.gsc_et import GSC
from prosper_amd.em.camodels._device import KernelTimer
dev = torch.device('cuda', 0)
Dm, Hm = 256, 128
g = torch.Generator(device=dev).manual_seed(3)
rng = np.random.RandomState(3)
N = 200_000
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, Hm, generator=g, device=dev) < 2.0 / Hm).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, Hm, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, Dm, generator=g, device=dev, dtype=torch.float64)
p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(Dm, Hm)), "pi": np.full(Hm, 2.0 / Hm),
     "mu": np.full(Hm, 1.4), "psi_sq": np.eye(Hm) * 1.1, "sigma_sq": 1.2}
class An(dict):
    crit_params=[]
    def __missing__(s,k): return 0.0
    def as_dict(s): return dict(s)
m = GSC(Dm, Hm, 6, 3, 'scalar')
import gc
gc.collect(); gc.disable()
for rep in range(3):
    for _ in range(10): p = m.step(An(T=1.0), p, {"y": Y})
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): p = m.step(An(T=1.0), p, {"y": Y})
    torch.cuda.synchronize()
    print("GSC c4 EM iter ms", round((time.perf_counter() - t) / 20 * 1e3, 3))
