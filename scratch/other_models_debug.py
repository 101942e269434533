import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from prosper_amd.em.camodels.gsc_et import GSC
from prosper_amd.em.camodels._device import KernelTimer
dev = torch.device('cuda', 0)
Dm, Hm = 256, 128
g = torch.Generator(device=dev).manual_seed(3)
rng = np.random.RandomState(3)
N = 200_000
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, Hm, generator=g, device=dev) < 2.0 / Hm).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, Hm, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, Dm, generator=g, device=dev, dtype=torch.float64)
p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(Dm, Hm)), "pi": np.full(Hm, 2.0 / Hm),
     "mu": np.full(Hm, 1.4), "psi_sq": np.eye(Hm) * 1.1, "sigma_sq": 1.2}
class An(dict):
    crit_params=[]
    def __missing__(s,k): return 0.0
    def as_dict(s): return dict(s)
m = GSC(Dm, Hm, 6, 3, 'scalar')
m.timer = KernelTimer()
import gc, threading
gc.collect(); gc.disable()
ev = []
def prof(frame, event, arg):
    if event in ("call", "return", "c_call", "c_return"):
        name = arg.__name__ if event.startswith("c_") else frame.f_code.co_name
        ev.append((time.perf_counter(), event, name, frame.f_lineno))
for it in range(12):
    torch.cuda.synchronize(); t = time.perf_counter()
    ev.clear()
    sys.setprofile(prof)
    p = m.step(An(T=1.0), p, {"y": Y})
    sys.setprofile(None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(it, round(dt * 1e3, 2), "threads", threading.active_count())
    if dt > 0.02 and it > 0:
        gaps = sorted(((ev[k + 1][0] - ev[k][0], k) for k in range(len(ev) - 1)), reverse=True)[:3]
        for g, k in gaps:
            print("GAP %.1f ms between" % (g * 1e3), ev[k][1:], "and", ev[k + 1][1:])
print({k: (v[0], round(v[1], 3)) for k, v in m.timer.summary().items()})
