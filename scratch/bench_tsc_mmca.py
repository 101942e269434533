"""EM-iteration timing of TSC_ET and MMCA_ET at D=256, H=128, H'=6, gamma=3, N=100k."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosper_amd.em.camodels.tsc_et import TSC_ET
from prosper_amd.em.camodels.mmca_et import MMCA_ET
from prosper_amd.em.camodels._device import KernelTimer
D,H,HP,GAMMA,N=256,128,6,3,100000
dev=torch.device('cuda',0)
g=torch.Generator(device=dev).manual_seed(0)
W_gt=torch.randn(D,H,generator=g,device=dev,dtype=torch.float64)*2
Y=torch.empty(N,D,dtype=torch.float64,device=dev)
for lo in range(0,N,25000):
    u=torch.rand(25000,H,generator=g,device=dev)
    S=(u<1.0/H).to(torch.float64)-(u>1-1.0/H).to(torch.float64)
    Y[lo:lo+25000]=S@W_gt.t()+torch.randn(25000,D,generator=g,device=dev,dtype=torch.float64)
W0=(W_gt+0.1*torch.randn(D,H,generator=g,device=dev,dtype=torch.float64)).cpu().numpy()
class An(dict):
    crit_params=[]
    def __missing__(s,k): return 0.0
    def as_dict(s): return dict(s)
an=An(T=1.0)
import gc
def run(name, m, p):
    for _ in range(6): q=m.step(an,dict(p),data)
    m.timer=KernelTimer()
    gc.collect(); gc.disable()
    torch.cuda.synchronize(); t=time.perf_counter()
    q=dict(p)
    for _ in range(10): q=m.step(an,q,data)
    torch.cuda.synchronize(); print(name, "EM iter ms", (time.perf_counter()-t)/10*1e3)
    gc.enable()
    print({k:round(v[1],3) for k,v in m.timer.summary().items()})
    import cProfile, pstats
    m.timer=None
    pr=cProfile.Profile(); pr.enable()
    for _ in range(5): q=m.step(an,q,data)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(8)
data={"y":Y}
run("TSC", TSC_ET(D,H,HP,GAMMA), {"W":W0,"pi":2.0/H,"sigma":1.0})
mm=MMCA_ET(D,H,HP,GAMMA)
run("MMCA", mm, mm.check_params({"W":W0.copy(),"pi":2.0/H,"sigma":1.0}))
