"""Debug aid: the list / dense split of GSC's moment contraction step by step (threshold, dense rows, column sums)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.gsc_et import GSC
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
D, H, Hp, gamma, N = 256, 128, 6, 3, 6000
rng = np.random.RandomState(41)
W_gt = rng.normal(size=(D, H))
S = rng.random_sample((N, H)) < 2.0 / H
y = (S * (1.5 + rng.normal(size=(N, H)))) @ W_gt.T + rng.normal(size=(N, D))
p0 = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
      "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
p0["pi"][7] = 1e-300
m = GSC(D, H, Hp, gamma, 'scalar', to_learn=['W', 'mu', 'psi_sq', 'sigma_sq'])
p = {k: np.array(v, copy=True) for k, v in p0.items()}
orig = m._launch_estep
def spy(res, A, G, psi_d, yn, tables, s2, T, cand_in, logpj=None, lists=False):
    out = orig(res, A, G, psi_d, yn, tables, s2, T, cand_in, logpj, lists)
    torch.cuda.synchronize()
    L = getattr(out[3], "_pm_lists", None)
    st = out[3]
    csz = st[2 * H * H + H:2 * H * H + 2 * H].cpu().numpy()
    print("  launch lists=%s s2=%g thr slot %.3e  dense %s  min|csz| %.3e (h=%d) xs[:,7] max %.3e" % (
        lists, s2, float(tables.reshape(-1)[8 * H + 1]) if tables.numel() > 8 * H + 1 else -1,
        int(L[3].item()) if L else None, np.abs(csz).min(), int(np.abs(csz).argmin()), float(out[1][:, 7].max())), flush=True)
    return out
m._launch_estep = spy
for it in range(6):
    p = m.step(An(T=1.0), p, {"y": y})
    print("step", it, "spec_hits", m.spec_hits, "pi7", p["pi"][7], "W7 norm", np.linalg.norm(p["W"][:, 7]), flush=True)
