"""Host time between the enqueue of the M-step's download and the launch of the next E-step (BSC config 2, flat loop)."""
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
from prosper_amd.em.camodels import _device
D, H, N = 1024, 256, 200_000
W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25_000):
    S = (torch.rand(25_000, H, generator=g, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25_000] = S @ W_gt.t() + torch.randn(25_000, D, generator=g, device=dev, dtype=torch.float64)
p = {"W": (W_gt + 0.1 * torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)).cpu().numpy(), "pi": 4.0 / H, "sigma": 1.0}
m = BSC_ET(D, H, 8, 4)
marks = {}
dl, sp, le, rs = BSC_ET._download, BSC_ET._speculate_estep, BSC_ET._launch_estep, BSC_ET._run_select_estep
def _download(self, flat, slot="default", then=None):
    if then is not None:
        marks["t0"] = time.perf_counter()
    return dl(self, flat, slot=slot, then=then)
def _speculate_estep(self, *a, **k):
    marks["t1"] = time.perf_counter()
    r = sp(self, *a, **k)
    marks["t3"] = time.perf_counter()
    acc.append(((marks["t1"] - marks["t0"]) * 1e6, (marks["t2"] - marks["t1"]) * 1e6, (marks["t3"] - marks["t2"]) * 1e6))
    return r
def _run_select_estep(self, *a, **k):
    marks["t2"] = time.perf_counter()
    return rs(self, *a, **k)
BSC_ET._download, BSC_ET._speculate_estep, BSC_ET._run_select_estep = _download, _speculate_estep, _run_select_estep
acc = []
for _ in range(40):
    p = m.step(An(T=1.0), p, {"y": Y})
torch.cuda.synchronize()
a = np.array(acc[10:])
print("us (median): download-enqueue -> _speculate_estep %.1f, -> kernel launch call %.1f, launch call itself %.1f" % tuple(np.median(a, axis=0)))
