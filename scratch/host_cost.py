"""Host (enqueue) time vs device time of one bench-style E-step pass."""
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
params = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
class An(dict):
    def __missing__(self, k): return 0.0
an = An(T=1.0)
m = BSC_ET(D, H, Hp, g)
data = {"y": Y}
Wt_host = np.ascontiguousarray(W0.T); Wt_dev = torch.from_numpy(Wt_host).to(dev)
def one():
    m.install_parameters(data, Wt_dev, Wt_host)
    d = m.select_Hprimes(params, data); return m.E_step(an, params, d)
for _ in range(30): one()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50): one()
t_enq = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print("enqueue ms/pass %.3f   total ms/pass %.3f" % (t_enq / 50 * 1e3, t_all / 50 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): one()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
