"""Soak: 400 EM steps of BSC config 2 through EM.run semantics (step loop), L must not decrease at T=1."""
import sys, time, numpy as np, torch
sys.path.insert(0,'.')
exec(open('scratch/em_loop.py').read().split("for _ in range(3): m.step")[0])
from prosper_amd.utils.datalog import dlog, StoreInMemory
h=dlog.set_handler(("L",), StoreInMemory)
q=dict(p); t=time.perf_counter()
for i in range(400): q=m.step(an,q,data)
torch.cuda.synchronize(); dt=time.perf_counter()-t
L=np.array(h.tables["L"],dtype=float)
print("400 steps %.2f s (%.2f ms/step); L first %.4f last %.4f; min diff %.3e; finite %s; sigma %.4f pi %.5f"%(dt,dt/400*1e3,L[0],L[-1],np.diff(L).min(),np.isfinite(q["W"]).all(),q["sigma"],q["pi"]))
print("mem", torch.cuda.max_memory_allocated()/1e9, "GB")
