"""BSC config-2 E-step passes alone with per-kernel timers (usable with timing-only kernel variants)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosper_amd.em.camodels.bsc_et import BSC_ET
from prosper_amd.em.camodels._device import KernelTimer
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
params = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
data = {"y": Y}
Wt_host = np.ascontiguousarray(W0.T)
Wt_dev = torch.from_numpy(Wt_host).to(dev)
def estep_pass():
    m.install_parameters(data, Wt_dev, Wt_host)
    d = m.select_Hprimes(params, data)
    return m.E_step(an, params, d)
for _ in range(3): estep_pass()
m.timer = KernelTimer()
for _ in range(8): estep_pass()
torch.cuda.synchronize()
print({k: round(v[1], 3) for k, v in m.timer.summary().items()})
