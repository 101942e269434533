"""Host share of an MCA EM iteration (config-5 dimensions): the loop at N = 100k and at N = 2048 (device work ~0: what is left
is the host path + launch / sync latencies), and a cProfile of the host side.  PYTHONPATH=. python scratch/mca_host_time.py"""
import os, sys, time, gc, cProfile, pstats
import numpy as np, torch
from prosper_amd.em.camodels.mca_et import MCA_ET
dev = torch.device("cuda", 0)
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
Dm, Hm = 256, 128
for N in (100_000, 2048):
    g = torch.Generator(device=dev).manual_seed(1)
    W_gt = torch.randn(Dm, Hm, generator=g, device=dev, dtype=torch.float64).abs() * 2 + 0.1
    S = torch.rand(N, Hm, generator=g, device=dev) < 2.0 / Hm
    Y = torch.empty(N, Dm, dtype=torch.float64, device=dev)
    for lo in range(0, N, 25_000):
        s = S[lo:lo + 25_000]
        Y[lo:lo + 25_000] = torch.where(s[:, None, :], W_gt[None], torch.zeros((), dtype=torch.float64, device=dev)).max(dim=2).values
    Y += torch.randn(N, Dm, generator=g, device=dev, dtype=torch.float64)
    p = {"W": (W_gt * (1 + 0.1 * (torch.rand(Dm, Hm, generator=g, device=dev, dtype=torch.float64) - 0.5))).cpu().numpy(),
         "pi": 2.0 / Hm, "sigma": 1.0}
    m = MCA_ET(Dm, Hm, 8, 3)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        p = m.step(An(T=1.0), p, {"y": Y})
    gc.collect(); gc.disable()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100):
        p = m.step(An(T=1.0), p, {"y": Y})
    torch.cuda.synchronize()
    print("N=%d em_iter %.4f ms" % (N, (time.perf_counter() - t) / 100 * 1e3))
    if N == 2048:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(200):
            p = m.step(An(T=1.0), p, {"y": Y})
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
    gc.enable()
