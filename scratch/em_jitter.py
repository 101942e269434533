"""Per-step wall time of the BSC EM loop at config 2 (looking for allocator / host stalls)."""
import sys, time, numpy as np, torch
sys.path.insert(0,'.')
exec(open('scratch/em_loop.py').read().split("for _ in range(3): m.step")[0])
import os
q=dict(p)
ts=[]
for i in range(60):
    t=time.perf_counter(); q=m.step(an,q,data); ts.append((time.perf_counter()-t)*1e3)
torch.cuda.synchronize()
print("per-step host-return times (ms):", " ".join("%.1f"%t for t in ts))
print("mean of last 40: %.2f  max %.1f"%(np.mean(ts[20:]), max(ts[20:])))
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
