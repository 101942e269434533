"""PCIe-inclusive rates of the BSC E-step at config 2: host y in (first call), host logpj out."""
import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from prosper_amd.em.camodels.bsc_et import BSC_ET
D,H,HP,GAMMA,N=1024,256,8,4,200000
rng=np.random.RandomState(0)
W=rng.normal(size=(D,H)); Y=rng.normal(size=(N,D))
class An(dict):
    crit_params=[]
    def __missing__(s,k): return 0.0
m=BSC_ET(D,H,HP,GAMMA)
p={"W":W,"pi":4.0/H,"sigma":1.0}
torch.cuda.synchronize()
t=time.perf_counter(); d=m.select_Hprimes(p,{"y":Y}); ss=m.E_step(An(T=1.0),p,d); torch.cuda.synchronize(); t1=time.perf_counter()-t
print("first pass incl. upload of y (1.64 GB pageable): %.1f ms -> %.2f M datapoints/s"%(t1*1e3, N/t1/1e6))
for _ in range(3):
    m._par={}; d=m.select_Hprimes(p,{"y":Y}); ss=m.E_step(An(T=1.0),p,d)
torch.cuda.synchronize()
t=time.perf_counter(); m._par={}; d=m.select_Hprimes(p,{"y":Y}); ss=m.E_step(An(T=1.0),p,d); torch.cuda.synchronize(); t2=time.perf_counter()-t
print("resident pass: %.2f ms"%(t2*1e3))
t=time.perf_counter(); m._par={}; d=m.select_Hprimes(p,{"y":Y}); ss=m.E_step(An(T=1.0),p,d); F=np.asarray(ss["logpj"]); c=np.asarray(d["candidates"]); t3=time.perf_counter()-t
print("pass + logpj/candidates copied to host NumPy (%.0f MB): %.1f ms -> %.2f M datapoints/s"%(F.nbytes/1e6, t3*1e3, N/t3/1e6))
