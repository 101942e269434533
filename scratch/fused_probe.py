"""Occupancy and launch time of the fused E-step kernel at config 2 (PM_FUSED_STAGES=3|4)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
from prosper_amd.em.camodels.bsc_et import BSC_ET
D, H, Hp, g, N = 1024, 256, 8, 4, int(os.environ.get("N", 200000))
lib = _lib.load()
print("stages", os.environ.get("PM_FUSED_STAGES", "4"), "occupancy", lib.pm_bsc_fused_occupancy(H, D, Hp, 154))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(min(25000, N - lo), H, generator=gen, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(S.shape[0], D, generator=gen, device=dev, dtype=torch.float64)
W0 = (W_gt + 0.1 * torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)).cpu().numpy()
params = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
class An(dict):
    def __missing__(self, k): return 0.0
an = An(T=1.0)
m = BSC_ET(D, H, Hp, g)
data = {"y": Y}
def one():
    d = m.select_Hprimes(params, data); return m.E_step(an, params, d)
IT = int(os.environ.get('ITERS', 30))
for _ in range(IT): one()
torch.cuda.synchronize()
for rep in range(3 if IT >= 30 else 1):
    t = time.perf_counter()
    for _ in range(IT): one()
    torch.cuda.synchronize()
    print("pass ms %.4f" % ((time.perf_counter() - t) / IT * 1e3))
