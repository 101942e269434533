"""Why the warm start of the Wq inverse is rejected on data-truncation steps: ||I - Wq_t X_(t-1)||_F per EM step at a
constant Ncut_factor, with and without rescaling X by N_use(t-1) / N_use(t)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
from prosper_amd.em.camodels.bsc_et import BSC_ET
from prosper_amd.utils.datalog import dlog, StoreInMemory

D, H, HP, GAMMA, N = 1024, 256, 8, 4, 100000
dev = torch.device("cuda", 0)
g0 = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)
W0 = np.ascontiguousarray((W_gt + 0.1 * torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)).cpu().numpy().T).T
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(25000, H, generator=g0, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(25000, D, generator=g0, device=dev, dtype=torch.float64)


class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)


lib = _lib.load()
o_wq, o_qd, o_mus = lib.pm_bsc_stats_offset_wq(H, D), lib.pm_bsc_stats_offset_qdiag(H, D), lib.pm_bsc_stats_offset_mus(H, D)
for ncut in (0.5, 1.0):
    m = BSC_ET(D, H, HP, GAMMA)
    m.speculate_estep = False
    p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0}
    h = dlog.set_handler(("N_use",), StoreInMemory)
    Xp, nprev = None, None
    for t in range(12):
        p = m.step(An(T=1.0, Ncut_factor=ncut), p, {"y": Y})
        st = m._ws["stats"].cpu().numpy()
        wq = st[o_wq:o_qd].reshape(H, H)
        A = np.triu(wq, 1)
        A = A + A.T + np.diag(np.diag(wq) + st[o_qd:o_mus])
        nu = int(h.tables["N_use"][-1])
        if Xp is not None:
            r = np.linalg.norm(np.eye(H) - A @ Xp)
            r2 = np.linalg.norm(np.eye(H) - A @ (Xp * nprev / nu))
            print("ncut %.1f step %2d N_use %d pi %.6f sigma %.5f  resid %.4f  rescaled %.4f  refine_next %s warm %s" % (
                ncut, t, nu, p["pi"], p["sigma"], r, r2, m._refine_next, m._winv_was_warm))
        Xp, nprev = np.linalg.inv(A), nu
    dlog.remove_handler(h)
