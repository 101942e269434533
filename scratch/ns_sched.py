"""Spectrum of Wq_t X_(t-1) along the reference's annealing schedule (BSC config 2 dims): what a scaled Newton-Schulz start
from the previous inverse needs (steps until ||R|| < 1e-8 with alpha = 1 / ||A X0||_inf)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
from prosper_amd.em import EM
from prosper_amd.em.annealing import LinearAnnealing
from prosper_amd.em.camodels.bsc_et import BSC_ET

D, H, HP, GAMMA, N = 1024, 256, 8, 4, 100000
dev = torch.device("cuda", 0)
g0 = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)
W0 = np.ascontiguousarray((W_gt + 0.1 * torch.randn(D, H, generator=g0, device=dev, dtype=torch.float64)).cpu().numpy().T).T
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(25000, H, generator=g0, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(25000, D, generator=g0, device=dev, dtype=torch.float64)
lib = _lib.load()
o_wq, o_qd, o_mus = lib.pm_bsc_stats_offset_wq(H, D), lib.pm_bsc_stats_offset_qdiag(H, D), lib.pm_bsc_stats_offset_mus(H, D)
m = BSC_ET(D, H, HP, GAMMA)
m.speculate_estep = False
an = LinearAnnealing(50)
an['T'] = [(0, 2.), (.7, 1.)]
an['Ncut_factor'] = [(0, 0.), (2. / 3, 1.)]
p = {"W": W0, "pi": 4.0 / H, "sigma": 1.0}
Xp = None
t = 0
while not an.finished:
    p = m.step(an, p, {"y": Y})
    st = m._ws["stats"].cpu().numpy()
    wq = st[o_wq:o_qd].reshape(H, H)
    A = np.triu(wq, 1)
    A = A + A.T + np.diag(np.diag(wq) + st[o_qd:o_mus])
    if Xp is not None:
        T = A @ Xp
        lam = np.sort(np.linalg.eigvals(T).real)
        ninf = np.abs(T).sum(axis=1).max()
        R = np.eye(H) - T / ninf
        k = 0
        while np.linalg.norm(R) >= 1e-8 and k < 30:
            R = R @ R
            k += 1
        print("step %2d T %.3f Ncut %.3f sigma %.4f resid_F %.3f lam [%.3f, %.3f] |T|inf %.2f  steps(before LAST) %d cond(A) %.1f" % (
            t, an['T'], an['Ncut_factor'], p["sigma"], np.linalg.norm(np.eye(H) - T), lam[0], lam[-1], ninf, k, np.linalg.cond(A)))
    Xp = np.linalg.inv(A)
    an.next(0.)
    t += 1
