"""The statistics-carrying E-step pass (select + E-step + M-step rows in one launch) on the bench workload with the
parameters of a running EM loop -- posteriors after a few iterations are broader than at the perturbed ground truth the
C++ harness (f8_bench) uses, and the statistics' atomics scale with the states that carry weight.
  python scratch/em_estep_time.py save            -> EM steps, parameters to scratch/em_params.npz
  PM_LIB_PATH=scratch/libs/libpm_x.so python scratch/em_estep_time.py   -> times the pass with that build"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosper_amd import _lib
if os.environ.get("PM_LIB_PATH"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PM_LIB_PATH"])
from prosper_amd.em.camodels.bsc_et import BSC_ET, KernelTimer

D, H, HP, GAMMA, N = 1024, 256, 8, 4, 200000
dev = torch.device("cuda", 0)
g0 = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)
W0 = np.ascontiguousarray((W_gt + 0.1 * torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)).cpu().numpy().T).T
gr = torch.Generator(device=dev).manual_seed(100)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(25000, H, generator=gr, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(25000, D, generator=gr, device=dev, dtype=torch.float64)


class Anneal(dict):
    crit_params = []

    def __missing__(self, k):
        return 0.0

    def as_dict(self):
        return dict(self)


anneal = Anneal(T=1.0, Ncut_factor=0.0, anneal_prior=False)
model = BSC_ET(D, H, HP, GAMMA)
data = {"y": Y}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "em_params.npz")
if len(sys.argv) > 1 and sys.argv[1] == "save":
    p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
    for _ in range(8):
        p = model.step(anneal, p, data)
    np.savez(path, W=np.asarray(p["W"]), pi=p["pi"], sigma=p["sigma"])
    print("saved", path, "sigma", p["sigma"], "pi*H", p["pi"] * H)
    sys.exit(0)
z = np.load(path)
p = {"W": z["W"], "pi": float(z["pi"]), "sigma": float(z["sigma"]), "mu": np.zeros(D)}
model._in_step = True
for phase in range(2):
    model.timer = KernelTimer() if phase else None
    for _ in range(12):
        d = model.select_Hprimes(p, dict(data))
        model.E_step(anneal, p, d)
    torch.cuda.synchronize()
print(os.environ.get("PM_LIB_PATH", "library"), {k: round(v[1], 4) for k, v in sorted(model.timer.summary().items())})
