"""MCA config 5 EM loop (flat schedule; argv[1] = cut for data-truncation steps): for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
from prosper_amd.em.camodels.mca_et import MCA_ET
D, H, N = 256, 128, 100_000
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64).abs() * 2 + 0.1
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25_000):
    S = torch.rand(25_000, H, generator=g, device=dev) < 2.0 / H
    Wm = torch.where(S[:, None, :], W_gt[None, :, :].expand(25_000, D, H), torch.zeros((), dtype=torch.float64, device=dev)).max(dim=2).values
    Y[lo:lo + 25_000] = Wm + torch.randn(25_000, D, generator=g, device=dev, dtype=torch.float64)
p = {"W": (W_gt * 1.05).cpu().numpy(), "pi": 2.0 / H, "sigma": 1.0}
m = MCA_ET(D, H, 8, 3)
cut = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
an = lambda: An(T=1.0, Ncut_factor=cut)
for _ in range(6):
    p = m.step(an(), p, {"y": Y})
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(12):
    p = m.step(an(), p, {"y": Y})
torch.cuda.synchronize()
print("cut %.2f: %.3f ms per EM iteration" % (cut, (time.perf_counter() - t) / 12 * 1e3))
