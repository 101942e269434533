"""MMCA EM iteration at config-5 dimensions (D=256 H=128 H'=8 gamma=3, N=100k), per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
if os.environ.get('PM_LIB_PATH'):
    _lib.LIB_PATH = os.path.abspath(os.environ['PM_LIB_PATH'])
from prosper_amd.em.camodels.mmca_et import MMCA_ET
from prosper_amd.em.camodels._device import KernelTimer
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
Dm, Hm, N = 256, 128, 100_000
g = torch.Generator(device=dev).manual_seed(3)
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64) * 2
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 25_000):
    u = torch.rand(25_000, Hm, generator=g, device=dev)
    S = (u < 1.0 / Hm).to(torch.float64) - (u > 1 - 1.0 / Hm).to(torch.float64)
    Y[lo:lo + 25_000] = S @ W_gt.t() + torch.randn(25_000, Dm, generator=g, device=dev, dtype=torch.float64)
p = {"W": (W_gt + 0.1 * torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)).cpu().numpy(), "pi": 2.0 / Hm, "sigma": 1.0}
m = MMCA_ET(Dm, Hm, 8, 3)
t = time.perf_counter()
while time.perf_counter() - t < 0.5:
    p = m.step(An(T=1.0), p, {"y": Y})
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10):
    p = m.step(An(T=1.0), p, {"y": Y})
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / 10 * 1e3
m.timer = kt = KernelTimer()
for _ in range(2):
    p = m.step(An(T=1.0), p, {"y": Y})
print("mmca em_iter %.3f ms" % ms, {k: round(v[1], 3) for k, v in sorted(kt.summary().items())})
