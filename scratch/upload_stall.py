"""Which host->device copy of a GSC EM loop stalls for ~80 ms, and what is special about it."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels import _device
from prosper_amd.em.camodels.gsc_et import GSC
log = []
orig = _device.DeviceCAModel._upload
def upload(self, name, host, keep=False):
    slot = self._pin.get(name)
    fresh = slot is None or slot["bufs"][slot["i"] ^ 1] is None
    t = time.perf_counter()
    out = orig(self, name, host, keep)
    log.append((time.perf_counter() - t, name, fresh, host.size, time.perf_counter() - T0))
    return out
_device.DeviceCAModel._upload = upload
dev = torch.device('cuda', 0)
D, H, N = 256, 128, 200000
g = torch.Generator(device=dev).manual_seed(3)
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.randn(N, D, generator=g, device=dev, dtype=torch.float64)
rng = np.random.RandomState(3)
p = {"W": W_gt.cpu().numpy(), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4), "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
m = GSC(D, H, 6, 3, 'scalar')
gc.collect(); gc.disable()
T0 = time.perf_counter()
for it in range(60):
    p = m.step(An(T=1.0), p, {"y": Y})
torch.cuda.synchronize()
for e in sorted(log, reverse=True)[:6]:
    print("%.1f ms  %-10s fresh=%s size=%d  at t=%.0f ms" % (e[0] * 1e3, e[1], e[2], e[3], e[4] * 1e3))
print("uploads", len(log), "total s", time.perf_counter() - T0)
ts = []
for it in range(60):
    torch.cuda.synchronize(); t = time.perf_counter()
    p = m.step(An(T=1.0), p, {"y": Y})
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("per-step ms: max %.1f at %d, median %.2f" % (max(ts), ts.index(max(ts)), sorted(ts)[30]))
m2 = GSC(D, H, 6, 3, 'scalar')
p2 = {"W": W_gt.cpu().numpy(), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4), "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
ts = []
T1 = time.perf_counter()
for it in range(120):
    torch.cuda.synchronize(); t = time.perf_counter()
    p2 = m2.step(An(T=1.0), p2, {"y": Y})
    torch.cuda.synchronize(); ts.append(((time.perf_counter() - t) * 1e3, (t - T1) * 1e3))
big = [(round(a, 1), it, round(b)) for it, (a, b) in enumerate(ts) if a > 10]
print("second model: steps > 10 ms (ms, index, start time ms):", big, "median %.2f" % sorted(a for a, _ in ts)[60])
