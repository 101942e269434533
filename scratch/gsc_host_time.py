"""Host share of a GSC EM iteration: the bench's GSC loop (config 4) at N = 200k and at N = 4096 (device work ~0: what is
left is the host path + launch/sync latencies), and a cProfile of the host side.  PYTHONPATH=. python scratch/gsc_host_time.py"""
import os, sys, time, gc, cProfile, pstats
import numpy as np, torch
from prosper_amd.em.camodels.gsc_et import GSC
dev = torch.device("cuda", 0)
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
Dm, Hm = 256, 128
for N in (200_000, 4096):
    g = torch.Generator(device=dev).manual_seed(3); rng = np.random.RandomState(3)
    W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64)
    S = (torch.rand(N, Hm, generator=g, device=dev) < 2.0 / Hm).to(torch.float64)
    Z = S * (1.5 + torch.randn(N, Hm, generator=g, device=dev, dtype=torch.float64))
    Y = Z @ W_gt.t() + torch.randn(N, Dm, generator=g, device=dev, dtype=torch.float64)
    p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(Dm, Hm)), "pi": np.full(Hm, 2.0 / Hm), "mu": np.full(Hm, 1.4),
         "psi_sq": np.eye(Hm) * 1.1, "sigma_sq": 1.2}
    m = GSC(Dm, Hm, 6, 3, 'scalar')
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        p = m.step(An(T=1.0), p, {"y": Y})
    gc.collect(); gc.disable()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200):
        p = m.step(An(T=1.0), p, {"y": Y})
    torch.cuda.synchronize()
    print("N=%d em_iter %.4f ms" % (N, (time.perf_counter() - t) / 200 * 1e3))
    if N == 4096:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(200):
            p = m.step(An(T=1.0), p, {"y": Y})
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    gc.enable()
