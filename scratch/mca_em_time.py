"""The MCA block of bench.py's other_models alone (config 5, one GPU's share): EM iterations from the perturbed
ground truth for 0.5 s, then the timed iterations and the per-kernel times -- the state the bench line reports (the
posteriors of converged parameters skip fewer terms than those of the starting point, which scratch/mca_kernel_time.py
times).  PYTHONPATH=. python scratch/mca_em_time.py"""
import os
import time
import torch
from prosper_amd import _lib
if os.environ.get('PM_LIB_PATH'):
    _lib.LIB_PATH = os.path.abspath(os.environ['PM_LIB_PATH'])
from prosper_amd.em.camodels.bsc_et import KernelTimer
from prosper_amd.em.camodels.mca_et import MCA_ET

Dm, Hm, N = 256, 128, 100_000
dev = torch.device("cuda", 0)


class Anneal(dict):
    crit_params = []

    def __missing__(self, k):
        return 0.0

    def as_dict(self):
        return dict(self)


g = torch.Generator(device=dev).manual_seed(1)
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64).abs() * 2 + 0.1
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 25_000):
    S = torch.rand(25_000, Hm, generator=g, device=dev) < 2.0 / Hm
    Wm = torch.where(S[:, None, :], W_gt[None, :, :].expand(25_000, Dm, Hm),
                     torch.zeros((), dtype=torch.float64, device=dev)).max(dim=2).values
    Y[lo:lo + 25_000] = Wm + torch.randn(25_000, Dm, generator=g, device=dev, dtype=torch.float64)
p = {"W": (W_gt * (1 + 0.1 * (2 * torch.rand(Dm, Hm, generator=g, device=dev, dtype=torch.float64) - 1))).cpu().numpy(),
     "pi": 2.0 / Hm, "sigma": 1.0}
m = MCA_ET(Dm, Hm, 8, 3)
t_warm = time.perf_counter()
while time.perf_counter() - t_warm < 0.5:
    p = m.step(Anneal(T=1.0), p, {"y": Y})
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    p = m.step(Anneal(T=1.0), p, {"y": Y})
torch.cuda.synchronize()
em = (time.perf_counter() - t) / 20 * 1e3
m.timer = kt = KernelTimer()
for _ in range(3):
    p = m.step(Anneal(T=1.0), p, {"y": Y})
m.timer = None
print("em_iter %.3f ms" % em, {k: round(v[1], 3) for k, v in sorted(kt.summary().items())})
