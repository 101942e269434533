"""Host profile (cProfile) of the DSC EM loop at the bench's dimensions: where the Python between two launches goes."""
import sys, time, cProfile, pstats, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosper_amd.em.camodels.dsc_et import DSC_ET
from prosper_amd.em.camodels.tsc_et import TSC_ET
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
which = sys.argv[1] if len(sys.argv) > 1 else "dsc"
if which == "dsc":
    m=DSC_ET(D,H,HP,GAMMA,states=np.array([-1.,0.,1.]))
    p={"W":W0,"pi":np.array([1.0/H,1-2.0/H,1.0/H]),"sigma":1.0}
else:
    m=TSC_ET(D,H,HP,GAMMA)
    p={"W":W0,"pi":1.0/H,"sigma":1.0}
data={"y":Y}
q=dict(p)
for _ in range(50): q=m.step(an,q,data)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
t=time.perf_counter()
for _ in range(300): q=m.step(an,q,data)
torch.cuda.synchronize()
el=(time.perf_counter()-t)/300*1e3
pr.disable()
print("EM iter ms (under cProfile)", el)
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
