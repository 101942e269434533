"""The first EM steps of the bench's BSC loop (config 2, from the perturbed ground truth): per-step wall-clock, the count of
overflowed non-zero lists (scalars[3]: non-zero = the dense product ran for the whole shard) and the mean list length."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
from prosper_amd.em.camodels.bsc_et import BSC_ET
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
o = _lib.load().pm_bsc_stats_offset_scalars(H, D) + 3
gc.disable()
for it in range(40):
    torch.cuda.synchronize(); t = time.perf_counter()
    p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    over = float(m._ws["stats"][o])
    idx = m._ws["nz_idx"].view(torch.uint8).view(N, 16, 2)
    nn = float(((m._ws["nz_idx"] != -1).sum(dim=1).double()).mean())
    print("step %2d  %.3f ms  overflowed lists %d  mean list %.2f  sigma %.4f" % (it, dt, over, nn, p["sigma"]))
# ---- per-kernel times of the first steps of a FRESH model (same data, resident)
from prosper_amd.em.camodels.bsc_et import KernelTimer
m2 = BSC_ET(D, H, HP, GAMMA)
p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0}
for it in range(14):
    m2.timer = kt = KernelTimer()
    torch.cuda.synchronize(); t = time.perf_counter()
    p = m2.step(an, p, {"y": Y})
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    m2.timer = None
    print("fresh step %2d %.3f ms" % (it, dt), {k: round(v[1], 3) for k, v in sorted(kt.summary().items())})
