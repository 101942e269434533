"""BSC config 2 EM iteration in its steady state (as bench.py's em_iter_steady_ms: 170 steps in, 4 x 25 timed) + per-kernel
times.  PM_LIB_PATH=<library> selects a variant build (same-box A/B).  PYTHONPATH=. python scratch/bsc_em_time.py"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
if os.environ.get('PM_LIB_PATH'):
    _lib.LIB_PATH = os.path.abspath(os.environ['PM_LIB_PATH'])
from prosper_amd.em.camodels.bsc_et import BSC_ET, KernelTimer
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
D, H, HP, GAMMA, N = 1024, 256, 8, 4, 200000
dev = torch.device("cuda", 0)
g0 = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)
W0 = (W_gt + 0.1 * torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)).cpu().numpy()
gr = torch.Generator(device=dev).manual_seed(100)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(25000, H, generator=gr, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(25000, D, generator=gr, device=dev, dtype=torch.float64)
m = BSC_ET(D, H, HP, GAMMA)
p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0}
an = An(T=1.0)
for _ in range(170):
    p = m.step(an, p, {"y": Y})
torch.cuda.synchronize()
gc.collect(); gc.disable()
best = 1e9
for _ in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(25):
        p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / 25 * 1e3)
m.timer = kt = KernelTimer()
for _ in range(3):
    p = m.step(an, p, {"y": Y})
m.timer = None
print("bsc em_iter %.4f ms" % best, {k: round(v[1], 4) for k, v in sorted(kt.summary().items())})
