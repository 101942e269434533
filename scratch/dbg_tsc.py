import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import golden
from prosper_amd.em.camodels.tsc_et import TSC_ET
class An(dict):
    crit_params=[]
    def __missing__(s,k): return 0.0
g=golden("tsc_inference.npz")
D,H,Hp,gamma=int(g["D"]),int(g["H"]),int(g["Hprime"]),int(g["gamma"])
m=TSC_ET(D,H,Hp,gamma)
p={"W":g["W"].copy(),"pi":float(g["pi"]),"sigma":float(g["sigma"])}
res=m.inference(An(T=1.0),p,{"y":g["y"]},topK=5,adaptive=False)
bad=np.where((res["s"]!=g["plain_s"]).any(axis=(1,2)))[0]
print("bad rows", bad)
lp,cand=m.compute_lpj(An(T=1.0),p,{"y":g["y"]})
lp=np.asarray(lp); cand=np.asarray(cand)
for n in bad[:3]:
    print("n",n,"cand",cand[n])
    print(" mine s", res["s"][n].tolist()); print(" ref  s", g["plain_s"][n].tolist())
    print(" mine p", res["p"][n], "\n ref  p", g["plain_p"][n])
    o=np.argsort(lp[n])[::-1][:6]; print(" top idx", o, lp[n][o], m.state_matrix[o].tolist())
