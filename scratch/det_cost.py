"""EM-iteration cost of `model.deterministic = True` (libprosper_hip_det.so) against the default build, BASELINE configs 2, 4, 5 (one GPU's share)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
def loop(m, p, Y, secs=0.4, steps=40):
    t = time.perf_counter()
    while time.perf_counter() - t < secs:
        p = m.step(An(T=1.0), p, {"y": Y})
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        p = m.step(An(T=1.0), p, {"y": Y})
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
g = torch.Generator(device=dev).manual_seed(3)
rng = np.random.RandomState(3)
out = {}
# BSC config 2
from prosper_amd.em.camodels.bsc_et import BSC_ET
D, H, N = 1024, 256, 200_000
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25_000):
    S = (torch.rand(25_000, H, generator=g, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25_000] = S @ W_gt.t() + torch.randn(25_000, D, generator=g, device=dev, dtype=torch.float64)
p0 = {"W": (W_gt + 0.1 * torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)).cpu().numpy(), "pi": 4.0 / H, "sigma": 1.0}
for det in (False, True):
    m = BSC_ET(D, H, 8, 4); m.deterministic = det
    out["bsc_c2_%s" % ("det" if det else "default")] = round(loop(m, dict(p0), Y), 3)
    del m
del Y; torch.cuda.empty_cache()
# GSC config 4
from prosper_amd.em.camodels.gsc_et import GSC
D, H, N = 256, 128, 200_000
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, H, generator=g, device=dev) < 2.0 / H).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, H, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, D, generator=g, device=dev, dtype=torch.float64)
p0 = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4), "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
for det in (False, True):
    m = GSC(D, H, 6, 3, 'scalar'); m.deterministic = det
    out["gsc_c4_%s" % ("det" if det else "default")] = round(loop(m, {k: np.array(v, copy=True) for k, v in p0.items()}, Y), 3)
    del m
del Y; torch.cuda.empty_cache()
# MCA config 5
from prosper_amd.em.camodels.mca_et import MCA_ET
N = 100_000
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64).abs() * 2 + 0.1
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25_000):
    S = torch.rand(25_000, H, generator=g, device=dev) < 2.0 / H
    Wm = torch.where(S[:, None, :], W_gt[None, :, :].expand(25_000, D, H), torch.zeros((), dtype=torch.float64, device=dev)).max(dim=2).values
    Y[lo:lo + 25_000] = Wm + torch.randn(25_000, D, generator=g, device=dev, dtype=torch.float64)
p0 = {"W": (W_gt * 1.05).cpu().numpy(), "pi": 2.0 / H, "sigma": 1.0}
for det in (False, True):
    m = MCA_ET(D, H, 8, 3); m.deterministic = det
    out["mca_c5_%s" % ("det" if det else "default")] = round(loop(m, dict(p0), Y, secs=0.6, steps=16), 3)
    del m
print(out)
