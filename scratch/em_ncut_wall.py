"""BSC config 2, Ncut_factor = 1 / 0.5: wall-clock per step with the apply kernel beside the sparse product (second stream) and behind it."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.bsc_et import BSC_ET
D, H, HP, GAMMA, N = 1024, 256, 8, 4, 200000
dev = torch.device("cuda", 0)
g0 = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)
W0 = np.ascontiguousarray((W_gt + 0.1 * torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)).cpu().numpy().T).T
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(25000, H, generator=g0, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(25000, D, generator=g0, device=dev, dtype=torch.float64)
class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)
for rep in range(2):
    for ncut in (1.0, 0.5):
        for ov in (True, False):
            m = BSC_ET(D, H, HP, GAMMA)
            m.overlap_apply = ov
            p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0}
            an = An(T=1.0, Ncut_factor=ncut)
            for _ in range(60):
                p = m.step(an, p, {"y": Y})
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(60):
                p = m.step(an, p, {"y": Y})
            torch.cuda.synchronize()
            print("Ncut %.1f overlap %s: %.3f ms/step" % (ncut, ov, (time.perf_counter() - t) / 60 * 1e3))
