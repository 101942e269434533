"""BSC config 2: per-kernel times of data-truncation steps (Ncut_factor = 1, T = 1; the plateau of the reference's schedule)
beside the flat-schedule step -- where the deferred-statistics path spends its time."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.bsc_et import BSC_ET, KernelTimer

D, H, HP, GAMMA, N = 1024, 256, 8, 4, int(os.environ.get("N", 200000))
dev = torch.device("cuda", 0)
g0 = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)
W0 = np.ascontiguousarray((W_gt + 0.1 * torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)).cpu().numpy().T).T
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    hi = min(N, lo + 25000)
    S = (torch.rand(hi - lo, H, generator=g0, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:hi] = S @ W_gt.t() + torch.randn(hi - lo, D, generator=g0, device=dev, dtype=torch.float64)


class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)


for ncut in (0.0, 1.0, 0.5):
    m = BSC_ET(D, H, HP, GAMMA)
    p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0}
    an = An(T=1.0, Ncut_factor=ncut)
    for _ in range(60):
        p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(40):
        p = m.step(an, p, {"y": Y})
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 40 * 1e3
    m.timer = kt = KernelTimer()
    for _ in range(3):
        p = m.step(an, p, {"y": Y})
    m.timer = None
    ks = kt.summary()
    print("Ncut %.1f: %.3f ms/step  hits %d  kernels(sum %.3f): %s" % (
        ncut, ms, m.spec_hits, sum(v[1] * v[0] / 3 for v in ks.values()),
        {k: (v[0] // 3, round(v[1], 4)) for k, v in sorted(ks.items())}))
