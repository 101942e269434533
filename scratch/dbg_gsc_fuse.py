import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
if os.environ.get('PM_LIB_PATH'):
    _lib.LIB_PATH = os.path.abspath(os.environ['PM_LIB_PATH'])
from prosper_amd.em.camodels.gsc_et import GSC
D, H, Hp, gamma = 256, 128, 6, 3
for N in (16, 1003):
    rng = np.random.RandomState(N)
    W_gt = rng.normal(size=(D, H))
    S = rng.random_sample((N, H)) < 2.0 / H
    y = (S * (1.5 + rng.normal(size=(N, H)))) @ W_gt.T + rng.normal(size=(N, D))
    p = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
         "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
    m = GSC(D, H, Hp, gamma, 'scalar')
    res = m._resident(y)
    par = m._tables_for(p, res)
    A = m._gemm_nt(res["Y"], par["Wst"], torch.empty((N, H), dtype=torch.float64, device=m.device), "scores_gemm")
    m._buf("scores", (N, H)).fill_(-7.0)
    out = m._launch_estep(res, None, par["G"], par["psi_d"], par["yn"], par["tables"], par["s2"], 1.0, None, Wst=par["Wst"])
    torch.cuda.synchronize()
    sc = m._buf("scores", (N, H))
    d = (sc - A).abs()
    bad = (d > 1e-9 * A.abs().max()).nonzero()
    print("N", N, "max err", float(d.max()), "bad entries", bad.shape[0], "untouched", int((sc == -7.0).sum()))
    if bad.shape[0]:
        b = bad.cpu().numpy()
        print(" rows", np.unique(b[:, 0])[:40], " cols", np.unique(b[:, 1])[:40])
        print(sc[b[0, 0], :8].cpu().numpy(), A[b[0, 0], :8].cpu().numpy())
    ref = m._launch_estep(res, A, par["G"], par["psi_d"], par["yn"], par["tables"], par["s2"], 1.0, None)
    print(out[0][:3].cpu().numpy(), ref[0][:3].cpu().numpy(), out[1][0, :6].cpu().numpy(), ref[1][0, :6].cpu().numpy())
    print("  cand equal", bool(torch.equal(out[0], ref[0])), "xs err", float((out[1] - ref[1]).abs().max()))
