"""Long EM loops with and without the pipeline features (speculative E-step / seeded selection, warm-started inverse,
M-statistics in the E-step pass, list forms of the contractions): the trajectories must agree.  BSC at config-2 dimensions,
GSC at config-4, DSC / TSC / MCA at the bench's dimensions (N = 20-40k).  STEPS=600 MODELS=BSC,GSC,DSC,TSC,MCA python scratch/soak.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
STEPS = int(os.environ.get("STEPS", 200))
dev = torch.device("cuda", 0)
class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)

def bsc(flags):
    os.environ["PM_WARM_INVERSE"] = "1" if flags else "0"
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, g, N = 1024, 256, 8, 4, 40000
    gen = torch.Generator(device=dev).manual_seed(0)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
    S = (torch.rand(N, H, generator=gen, device=dev) < 4.0 / H).to(torch.float64)
    Y = S @ W_gt.t() + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    p = {"W": (W_gt + 0.3 * torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)).cpu().numpy(), "pi": 2.0 / H, "sigma": 2.0}
    m = BSC_ET(D, H, Hp, g)
    m.speculate_estep = m.fuse_mstats = flags
    t = time.perf_counter()
    for it in range(STEPS):
        T = 1.5 if it < 20 else (1.5 - 0.5 * (it - 20) / 30 if it < 50 else 1.0)
        p = m.step(An(T=T, Ncut_factor=0.5 if 60 <= it < 70 else 0.0), p, {"y": Y})
    torch.cuda.synchronize()
    return p, getattr(m, "spec_hits", 0), (time.perf_counter() - t) / STEPS * 1e3

def gsc(flags):
    os.environ["PM_WARM_INVERSE"] = "1" if flags else "0"
    from prosper_amd.em.camodels.gsc_et import GSC
    D, H, Hp, g, N = 256, 128, 6, 3, 40000
    gen = torch.Generator(device=dev).manual_seed(1)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
    S = (torch.rand(N, H, generator=gen, device=dev) < 2.0 / H).to(torch.float64)
    Y = (S * (1.5 + torch.randn(N, H, generator=gen, device=dev, dtype=torch.float64))) @ W_gt.t() + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    rng = np.random.RandomState(0)
    p = {"W": W_gt.cpu().numpy() + 0.2 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.0), "psi_sq": np.eye(H), "sigma_sq": 2.0}
    m = GSC(D, H, Hp, g, 'scalar')
    m.speculate_estep = flags
    m.fuse_moment_gemm = flags
    t = time.perf_counter()
    for it in range(STEPS):
        p = m.step(An(T=1.3 if it < 30 else 1.0), p, {"y": Y})
    torch.cuda.synchronize()
    return p, m.spec_hits, (time.perf_counter() - t) / STEPS * 1e3

def dsc(flags, cls="dsc"):
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    D, H, Hp, g, N = 256, 128, 6, 3, 40000
    gen = torch.Generator(device=dev).manual_seed(2)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64) * 2
    u = torch.rand(N, H, generator=gen, device=dev)
    S = (u < 1.0 / H).to(torch.float64) - (u > 1 - 1.0 / H).to(torch.float64)
    Y = S @ W_gt.t() + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    W0 = (W_gt + 0.1 * torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)).cpu().numpy()
    if cls == "dsc":
        m = DSC_ET(D, H, Hp, g, states=np.array([-1., 0., 1.]))
        p = {"W": W0, "pi": np.array([1.0 / H, 1 - 2.0 / H, 1.0 / H]), "sigma": 1.0}
    else:
        m = TSC_ET(D, H, Hp, g)
        p = {"W": W0, "pi": 1.0 / H, "sigma": 1.0}
    m.speculate = flags
    m.fuse_mstats = flags
    t = time.perf_counter()
    for it in range(STEPS):
        p = m.step(An(T=1.2 if it < 30 else 1.0), p, {"y": Y})
        p = {k: p[k] for k in ("W", "pi", "sigma")}
    torch.cuda.synchronize()
    return p, 0, (time.perf_counter() - t) / STEPS * 1e3

def tsc(flags):
    return dsc(flags, "tsc")

def mca(flags):
    from prosper_amd.em.camodels.mca_et import MCA_ET
    D, H, Hp, g, N = 256, 128, 8, 3, 20000
    gen = torch.Generator(device=dev).manual_seed(3)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64).abs() * 2 + 0.1
    S = torch.rand(N, H, generator=gen, device=dev) < 2.0 / H
    Y = torch.where(S[:, None, :], W_gt[None, :, :].expand(1, D, H), torch.zeros((), dtype=torch.float64, device=dev)).amax(dim=2) \
        if False else torch.stack([torch.where(S[lo:lo + 2000, None, :], W_gt[None], torch.zeros((), dtype=torch.float64, device=dev)).amax(dim=2)
                                   for lo in range(0, N, 2000)]).reshape(N, D)
    Y = Y + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    p = {"W": (W_gt * (1 + 0.1 * (2 * torch.rand(D, H, generator=gen, device=dev, dtype=torch.float64) - 1))).cpu().numpy(),
         "pi": 2.0 / H, "sigma": 1.0}
    m = MCA_ET(D, H, Hp, g)
    # MCA_SOAK=same: both runs identical (what the f64 atomics' order alone does to a long trajectory); =spec: only the seeding differs
    mode = os.environ.get('MCA_SOAK', 'all')
    m.speculate = True if mode == 'same' else flags
    m.fuse_em = True if mode in ('same', 'spec') else flags
    t = time.perf_counter()
    for it in range(STEPS):
        p = m.step(An(T=1.2 if it < 30 else 1.0), p, {"y": Y})
        p = {k: p[k] for k in ("W", "pi", "sigma")}
    torch.cuda.synchronize()
    return p, 0, (time.perf_counter() - t) / STEPS * 1e3

MODELS = {"BSC": (bsc, ("W", "pi", "sigma")), "GSC": (gsc, ("W", "pi", "mu", "psi_sq", "sigma_sq")),
          "DSC": (dsc, ("W", "pi", "sigma")), "TSC": (tsc, ("W", "pi", "sigma")), "MCA": (mca, ("W", "pi", "sigma"))}
which = os.environ.get("MODELS", "BSC,GSC").split(",")
for name, fn, keys in [(n,) + MODELS[n] for n in which]:
    steps, STEPS = STEPS, 5
    fn(True); fn(False)              # one-time costs (code objects, allocator pools) out of the clocks
    STEPS = steps
    a, hits, ms_a = fn(True)
    b, _, ms_b = fn(False)
    print(name, "steps", STEPS, "speculative hits", hits, "ms/step %.2f (features on) %.2f (off)" % (ms_a, ms_b))
    for k in keys:
        x, y = np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64)
        print("   %-9s max rel diff %.2e   finite %s" % (k, np.abs(x - y).max() / max(np.abs(y).max(), 1e-300), np.isfinite(x).all()))
