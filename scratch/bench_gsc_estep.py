"""GSC E-step kernel alone at config 4 (no M-step: usable with timing-only kernel variants)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosper_amd.em.camodels.gsc_et import GSC
from prosper_amd.em.camodels._device import KernelTimer
D, H, HP, GAMMA, N = 256, 128, 6, 3, 200000
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 50000):
    S = (torch.rand(50000, H, generator=g, device=dev) < 2.0 / H).to(torch.float64)
    Z = S * (1.5 + torch.randn(50000, H, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50000] = Z @ W_gt.t() + torch.randn(50000, D, generator=g, device=dev, dtype=torch.float64)
rng = np.random.RandomState(0)
p = {"W": (W_gt.cpu().numpy() + 0.1 * rng.normal(size=(D, H))), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
     "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
m = GSC(D, H, HP, GAMMA, 'scalar')
res = m._resident(Y)
for _ in range(3):
    m._run(1.0, p, res, None)
m.timer = KernelTimer()
for _ in range(10):
    m._run(1.0, p, res, None)
torch.cuda.synchronize()
print({k: round(v[1], 3) for k, v in m.timer.summary().items()})
cand = m._run(1.0, p, res, None)[0]
m.timer = KernelTimer()
for _ in range(10):
    m._run(1.0, p, res, cand)
torch.cuda.synchronize()
print("given candidates:", {k: round(v[1], 3) for k, v in m.timer.summary().items()})
