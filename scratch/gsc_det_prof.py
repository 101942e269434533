"""GSC config 4 EM loop, default or deterministic build (argv[1] = det|default): for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
rng = np.random.RandomState(3)
from prosper_amd.em.camodels.gsc_et import GSC
D, H, N = 256, 128, 200_000
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, H, generator=g, device=dev) < 2.0 / H).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, H, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, D, generator=g, device=dev, dtype=torch.float64)
p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4), "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
m = GSC(D, H, 6, 3, 'scalar'); m.deterministic = sys.argv[1] == "det"
for _ in range(20):
    p = m.step(An(T=1.0), p, {"y": Y})
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(40):
    p = m.step(An(T=1.0), p, {"y": Y})
torch.cuda.synchronize()
print("%s: %.3f ms per EM iteration, %d adopted" % (sys.argv[1], (time.perf_counter() - t) / 40 * 1e3, m.spec_hits))
