"""Steady-state ms per BSC EM iteration at config 2 (device-generated data): 0.6 s of warm-up, then 6 blocks of 20 steps;
prints min / median block.  Run from the repo root or from a copy of the tree (scratch/ab_base): imports the prosper_amd
next to the directory given as argv[1] (default: this file's parent's parent)."""
import os, sys, time, gc
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
from prosper_amd.em.camodels.bsc_et import BSC_ET
D, H, HP, GAMMA, N = 1024, 256, 8, 4, 200000
dev = torch.device('cuda', 0)
g0 = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)
W0 = (W_gt + 0.1 * torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)).cpu().numpy()
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(25000, H, generator=g0, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(25000, D, generator=g0, device=dev, dtype=torch.float64)
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
an = An(T=1.0)
m = BSC_ET(D, H, HP, GAMMA)
q = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
data = {"y": Y}
t = time.perf_counter()
while time.perf_counter() - t < 0.6:
    q = m.step(an, q, data)
torch.cuda.synchronize()
gc.collect(); gc.disable()
blocks = []
for b in range(6):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        q = m.step(an, q, data)
    torch.cuda.synchronize()
    blocks.append((time.perf_counter() - t) / 20 * 1e3)
print("ms/iter min %.4f median %.4f  spec_hits %s" % (min(blocks), sorted(blocks)[3], getattr(m, "spec_hits", None)))
