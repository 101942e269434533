import numpy as np, sys, torch
sys.path.insert(0,'.')
from oracle import mca_oracle as M
from prosper_amd.em.camodels.mca_et import MCA_ET
g=dict(np.load('tests/golden/mca_step_h128.npz'))
class An(dict):
    crit_params=[]
    def __missing__(s,k): return 0.0
an=An(T=float(g["T"]),Ncut_factor=float(g["Ncut_factor"]))
m=MCA_ET(int(g["D"]),int(g["H"]),int(g["Hprime"]),int(g["gamma"]))
params=m.check_params({"W":g["W"].copy(),"pi":float(g["pi"]),"sigma":float(g["sigma"])})
data=m.select_Hprimes(params,{"y":g["y"]}); ss=m.E_step(an,params,data); new=m.M_step(an,params,ss,data)
err=np.abs(new["W"]-g["W_new"])
bad=np.where(err.max(axis=0)>1e-8)[0]
print("bad latents",len(bad), bad[:20])
model=M.make_model(int(g["D"]),int(g["H"]),int(g["Hprime"]),int(g["gamma"]))
ref,log=M.m_step(M.Anneal(T=float(g["T"])),model,params["W"],params["pi"],params["sigma"],g["y"],g["candidates"],g["logpj"],vec=True)
st=m._ws["mca_stats"].cpu().numpy(); H,D=128,64
G1=st[:H*D].reshape(H,D); Wpm=st[H*D:2*H*D].reshape(H,D); Wqm=st[2*H*D:3*H*D].reshape(H,D); q1s=st[3*H*D:3*H*D+H]
Wt=params["W"].T
Wq=q1s[:,None]*Wt*Wt+Wqm; Wp=G1*Wt*Wt+Wpm
rWq=log["stats"]["Wq"]; rWp=log["stats"]["Wp"]
for h in bad[:6]:
    print(h, "hmax", m._ws["mca_hmax"][h].item(), "q1sum", q1s[h], "Wq mine/ref", Wq[h,:3], rWq[h,:3], "Wqm", Wqm[h,:3], "ratio", (Wq[h]/rWq[h])[:3])
