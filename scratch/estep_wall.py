"""Wall-clock of BSC config-2 E-step passes (no per-kernel events)."""
import os, sys, time, numpy as np, torch, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_bsc_estep.py")).read().split("for _ in range(3): estep_pass()")[0]
exec(src)
for _ in range(5): estep_pass()
gc.collect(); gc.disable()
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): estep_pass()
    torch.cuda.synchronize()
    print("ms/pass", round((time.perf_counter() - t) / 20 * 1e3, 4))
