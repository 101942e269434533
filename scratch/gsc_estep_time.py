"""The GSC E-step kernel alone (list form, as the EM loop's steady state launches it) on the parameters of a RUNNING EM loop at
config 4:  python scratch/gsc_estep_time.py save  -> scratch/gsc_params.npz;  PM_LIB_PATH=... python scratch/gsc_estep_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
if os.environ.get('PM_LIB_PATH'):
    _lib.LIB_PATH = os.path.abspath(os.environ['PM_LIB_PATH'])
from prosper_amd.em.camodels.gsc_et import GSC
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
m = GSC(Dm, Hm, 6, 3, 'scalar')
f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gsc_params.npz")
if sys.argv[1:] == ["save"]:
    p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(Dm, Hm)), "pi": np.full(Hm, 2.0 / Hm), "mu": np.full(Hm, 1.4),
         "psi_sq": np.eye(Hm) * 1.1, "sigma_sq": 1.2}
    t = time.perf_counter()
    while time.perf_counter() - t < 0.5:
        p = m.step(An(T=1.0), p, {"y": Y})
    res = m._resident(Y)
    csz = None
    out = m.E_step(An(T=1.0), p, m.select_Hprimes(p, {"y": Y}))
    dmin = min(float(out['_sums'][0].abs().min()), float(out['xpt_szsz'].sum(axis=0).diagonal().abs().min()))
    np.savez(f, csz_min=float(out['_sums'][1].abs().min()), diag_min=dmin, **p)
    sys.exit(0)
z = np.load(f)
p = {k: z[k] for k in ("W", "pi", "mu", "psi_sq")}
p["sigma_sq"] = float(z["sigma_sq"])
res = m._resident(Y)
par = m._tables_for(p, res)
A = m._gemm_nt(res["Y"], par["Wst"], m._buf("scores", (N, Hm)), "scores_gemm")
tdev = torch.zeros(9 * Hm, dtype=torch.float64, device=dev)
tdev[:8 * Hm] = par["tables"].reshape(-1)[:8 * Hm]
tdev[8 * Hm] = 1.0 / par["s2"]
tdev[8 * Hm + 1] = float(z["csz_min"]) * 2.0 ** -75
if os.environ.get("PAIR_THR", "1") == "1":
    tdev[8 * Hm + 2] = float(z["diag_min"]) * 2.0 ** -75
lists = os.environ.get("LISTS", "1") == "1"
fuse = os.environ.get("FUSE", "0") == "1"
if fuse:
    _A, A = A, None
kw = {"Wst": par["Wst"]} if fuse else {}
for _ in range(30):
    out = m._launch_estep(res, A, par["G"], par["psi_d"], par["yn"], tdev, 0.0, 1.0, None, lists=lists, **kw)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = m._launch_estep(res, A, par["G"], par["psi_d"], par["yn"], tdev, 0.0, 1.0, None, lists=lists, **kw)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
L = getattr(out[3], "_pm_lists", None)
print("fuse %d " % fuse, end="")
print("estep call %.4f ms (incl. stats zero fill + fold launch)  dense rows %s" % (best, int(L[3].item()) if L else None))
