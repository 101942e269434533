"""Where GSC's deterministic mode loses its time: the default build with the features the mode switches off, one at a time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
def loop(m, p, Y, secs=0.4, steps=30):
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
from prosper_amd.em.camodels.gsc_et import GSC
D, H, N = 256, 128, 200_000
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 50_000):
    S = (torch.rand(50_000, H, generator=g, device=dev) < 2.0 / H).to(torch.float64)
    Z = S * (1.5 + torch.randn(50_000, H, generator=g, device=dev, dtype=torch.float64))
    Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, D, generator=g, device=dev, dtype=torch.float64)
p0 = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4), "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
def run(label, det=False, **kw):
    m = GSC(D, H, 6, 3, 'scalar'); m.deterministic = det
    for k, v in kw.items():
        assert hasattr(m, k), k
        setattr(m, k, v)
    t = loop(m, {k: np.array(v, copy=True) for k, v in p0.items()}, Y)
    print("%-50s %.3f ms" % (label, t), flush=True)
run("default")
run("default, no lists (sparse_moments=False)", sparse_moments=False)
run("default, no lists, no early inverse", sparse_moments=False, early_inverse=False)
run("default, no speculative E-step", speculate_estep=False)
run("default, no spec E-step, no lists, no early inv", speculate_estep=False, sparse_moments=False, early_inverse=False)
run("deterministic", det=True)
for a in sys.argv[1:]:
    run("deterministic, " + a, det=True, **{a: True})
