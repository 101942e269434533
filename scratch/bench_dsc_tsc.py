"""EM-iteration time + kernel times of DSC / TSC at the bench's shapes (D=256 H=128 H'=6 gamma=3, N=100k)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
if os.environ.get('PM_LIB_PATH'):
    _lib.LIB_PATH = os.path.abspath(os.environ['PM_LIB_PATH'])
from prosper_amd.em.camodels.dsc_et import DSC_ET
from prosper_amd.em.camodels.tsc_et import TSC_ET
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
W0 = (W_gt + 0.1 * torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)).cpu().numpy()
for fuse in (True, False):
    for name, m, p in (("dsc", DSC_ET(Dm, Hm, 6, 3, states=np.array([-1., 0., 1.])),
                        {"W": W0, "pi": np.array([1.0 / Hm, 1 - 2.0 / Hm, 1.0 / Hm]), "sigma": 1.0}),
                       ("tsc", TSC_ET(Dm, Hm, 6, 3), {"W": W0, "pi": 2.0 / Hm, "sigma": 1.0})):
        m.fuse_mstats = fuse
        t = time.perf_counter()
        while time.perf_counter() - t < 0.4:
            p = m.step(An(T=1.0), p, {"y": Y})
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(25):
                p = m.step(An(T=1.0), p, {"y": Y})
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / 25 * 1e3)
        gc.enable()
        m.timer = kt = KernelTimer()
        for _ in range(3):
            p = m.step(An(T=1.0), p, {"y": Y})
        m.timer = None
        ks = {k: round(v[1], 4) for k, v in sorted(kt.summary().items())}
        print("%s fuse=%s: %.4f ms/iter  %s" % (name, fuse, best, ks), flush=True)
