"""Non-zeros per row of E[s] (the E-step pass's M-step rows) on the bench workload: how sparse is the left operand of
the statistics GEMM?  python scratch/nnz_hist.py [N]"""
import sys
import numpy as np
import torch
from prosper_amd.em.camodels.bsc_et import BSC_ET

D, H, HP, GAMMA = 1024, 256, 8, 4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device("cuda", 0)
g0 = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)
W0 = np.ascontiguousarray((W_gt + 0.1 * torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)).cpu().numpy().T).T
gr = torch.Generator(device=dev).manual_seed(100)
S = (torch.rand(N, H, generator=gr, device=dev) < 4.0 / H).to(torch.float64)
Y = S @ W_gt.t() + torch.randn(N, D, generator=gr, device=dev, dtype=torch.float64)


class Anneal(dict):
    crit_params = []

    def __missing__(self, k):
        return 0.0

    def as_dict(self):
        return dict(self)


for T in (1.0, 2.0, 8.0):
    anneal = Anneal(T=T, Ncut_factor=0.0, anneal_prior=False)
    model = BSC_ET(D, H, HP, GAMMA)
    data = {"y": Y}
    p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
    for it in range(4):
        p = model.step(anneal, p, data)
        torch.cuda.synchronize()
        e = model._buf("expect", (N, H))
        nnz = (e != 0).sum(1).cpu().numpy()
        big = (e > 1e-12).sum(1).cpu().numpy()
        print("T=%g it=%d nnz mean %.2f median %d p99 %d max %d  frac>31: %.5f  (>1e-12: mean %.2f)  sigma %.4f" % (
            T, it, nnz.mean(), np.median(nnz), np.percentile(nnz, 99), nnz.max(), (nnz > 31).mean(), big.mean(),
            p["sigma"]))
