import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosper_amd.em.camodels.gsc_et import GSC
from prosper_amd.em.camodels._device import KernelTimer
D,H,HP,GAMMA,N=256,128,6,3,200000
dev=torch.device('cuda',0)
g=torch.Generator(device=dev).manual_seed(0)
W_gt=torch.randn(D,H,generator=g,device=dev,dtype=torch.float64)
Y=torch.empty(N,D,dtype=torch.float64,device=dev)
for lo in range(0,N,50000):
    S=(torch.rand(50000,H,generator=g,device=dev)<2.0/H).to(torch.float64)
    Z=S*(1.5+torch.randn(50000,H,generator=g,device=dev,dtype=torch.float64))
    Y[lo:lo+50000]=Z@W_gt.t()+torch.randn(50000,D,generator=g,device=dev,dtype=torch.float64)
rng=np.random.RandomState(0)
p={"W":(W_gt.cpu().numpy()+0.1*rng.normal(size=(D,H))),"pi":np.full(H,2.0/H),"mu":np.full(H,1.4),"psi_sq":np.eye(H)*1.1,"sigma_sq":1.2}
class An(dict):
    crit_params=[]
    def __missing__(s,k): return 0.0
    def as_dict(s): return dict(s)
an=An(T=1.0)
m=GSC(D,H,HP,GAMMA,'scalar')
data={"y":Y}
cp=lambda q:{k:np.array(v,copy=True) for k,v in q.items()}
for _ in range(2): q=m.step(an,cp(p),data)
m.timer=KernelTimer()
torch.cuda.synchronize(); t=time.perf_counter()
q=cp(p)
for _ in range(3): q=m.step(an,q,data)
torch.cuda.synchronize(); print("GSC EM iter ms", (time.perf_counter()-t)/3*1e3, "sigma_sq", q["sigma_sq"])
print({k:round(v[1],3) for k,v in m.timer.summary().items()})
import cProfile, pstats
m.timer=None
pr=cProfile.Profile(); pr.enable()
q=cp(p)
for _ in range(3): q=m.step(an,q,data)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
