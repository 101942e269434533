"""GSC config 4: how long the host WAITS per EM step (torch.cuda.Event.synchronize / Stream.synchronize inside step()) -- the
slack of the host path: if it is near zero the loop is host-bound on this box."""
import os, sys, time, gc
import numpy as np, torch
from prosper_amd.em.camodels.gsc_et import GSC
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
Dm, Hm, N = 256, 128, 200_000
g = torch.Generator(device=dev).manual_seed(3); rng = np.random.RandomState(3)
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, Hm, generator=g, device=dev) < 2.0 / Hm).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, Hm, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, Dm, generator=g, device=dev, dtype=torch.float64)
p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(Dm, Hm)), "pi": np.full(Hm, 2.0 / Hm), "mu": np.full(Hm, 1.4),
     "psi_sq": np.eye(Hm) * 1.1, "sigma_sq": 1.2}
m = GSC(Dm, Hm, 6, 3, 'scalar')
waits = [0.0]
for cls, name in ((torch.cuda.Event, "synchronize"), (torch.cuda.Stream, "synchronize")):
    orig = getattr(cls, name)
    def wrapped(self, _o=orig):
        t = time.perf_counter(); r = _o(self); waits[0] += time.perf_counter() - t; return r
    setattr(cls, name, wrapped)
for _ in range(300):
    p = m.step(An(T=1.0), p, {"y": Y})
gc.collect(); gc.disable()
waits[0] = 0.0
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200):
    p = m.step(An(T=1.0), p, {"y": Y})
el = time.perf_counter() - t
print("em_iter %.4f ms   host waiting %.4f ms per step   host busy %.4f ms per step" % (el / 200 * 1e3, waits[0] / 200 * 1e3, (el - waits[0]) / 200 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    p = m.step(An(T=1.0), p, {"y": Y})
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
