import sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0,'.')
src=open('scratch/bench_mca.py').read().split("for _ in range(2): q=m.step")[0]
exec(src)
for _ in range(3): q=m.step(an,dict(p),data)
torch.cuda.synchronize()
pr=cProfile.Profile()
q=dict(p)
t=time.perf_counter(); pr.enable()
for _ in range(10): q=m.step(an,q,data)
torch.cuda.synchronize(); pr.disable()
print("ms/iter",(time.perf_counter()-t)*100)
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
