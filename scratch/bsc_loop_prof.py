"""BSC config 2 EM loop (argv[1] = Ncut_factor, argv[2] = det|default): for rocprofv3 --kernel-trace + scratch/timeline.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
from prosper_amd.em.camodels.bsc_et import BSC_ET
D, H, N = 1024, 256, 200_000
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25_000):
    S = (torch.rand(25_000, H, generator=g, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25_000] = S @ W_gt.t() + torch.randn(25_000, D, generator=g, device=dev, dtype=torch.float64)
p = {"W": (W_gt + 0.1 * torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)).cpu().numpy(), "pi": 4.0 / H, "sigma": 1.0}
cut = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
m = BSC_ET(D, H, 8, 4); m.deterministic = len(sys.argv) > 2 and sys.argv[2] == "det"
an = lambda: An(T=1.0, Ncut_factor=cut)
for _ in range(15):
    p = m.step(an(), p, {"y": Y})
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(30):
    p = m.step(an(), p, {"y": Y})
torch.cuda.synchronize()
print("cut %.2f: %.3f ms per EM iteration, %d adopted" % (cut, (time.perf_counter() - t) / 30 * 1e3, m.spec_hits))
