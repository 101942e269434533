"""GSC config 4 EM iteration (steady state) + per-kernel times, as bench.py's other_models block measures it."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
if os.environ.get('PM_LIB_PATH'):
    _lib.LIB_PATH = os.path.abspath(os.environ['PM_LIB_PATH'])
from prosper_amd.em.camodels.gsc_et import GSC
from prosper_amd.em.camodels._device import KernelTimer
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
Dm, Hm, N = 256, 128, 200_000
g = torch.Generator(device=dev).manual_seed(3)
rng = np.random.RandomState(3)
W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, Hm, generator=g, device=dev) < 2.0 / Hm).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, Hm, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, Dm, generator=g, device=dev, dtype=torch.float64)
p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(Dm, Hm)), "pi": np.full(Hm, 2.0 / Hm), "mu": np.full(Hm, 1.4),
     "psi_sq": np.eye(Hm) * 1.1, "sigma_sq": 1.2}
m = GSC(Dm, Hm, 6, int(os.environ.get('GAMMA', '3')), 'scalar')
m.sparse_moments = os.environ.get('SPM', '1') == '1'
m.overlap_moments = os.environ.get('OVL', '1') == '1'
m.fuse_scores = os.environ.get('FUSE', '0') == '1'
m.early_inverse = os.environ.get('EARLY', '1') == '1'
m.list_pairs = os.environ.get('PAIRS', '1') == '1'
m.overlap_scores = os.environ.get('OVS', '1') == '1'
t = time.perf_counter()
while time.perf_counter() - t < 0.5:
    p = m.step(An(T=1.0), p, {"y": Y})
torch.cuda.synchronize()
gc.collect(); gc.disable()
best = 1e9
for _ in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(25):
        p = m.step(An(T=1.0), p, {"y": Y})
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / 25 * 1e3)
m.timer = kt = KernelTimer()
for _ in range(3):
    p = m.step(An(T=1.0), p, {"y": Y})
m.timer = None
print("gsc em_iter %.4f ms  spec_hits %d" % (best, m.spec_hits), {k: round(v[1], 4) for k, v in sorted(kt.summary().items())})
