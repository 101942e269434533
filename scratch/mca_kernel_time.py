"""Time of the fused MCA E-step + M-statistics kernel alone at config 5 (select_Hprimes + E_step repeated with the same
parameters: valid for the -DPM_MCA_ABL timing builds, whose results are wrong by design)."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_mca.py")).read().split("for _ in range(2)")[0])
q = m.check_params(dict(p))
for it in range(3):
    d = m.select_Hprimes(q, dict(data)); e = m.E_step(an, q, d)
m.timer = KernelTimer()
for it in range(6):
    d = m.select_Hprimes(q, dict(data)); e = m.E_step(an, q, d)
torch.cuda.synchronize()
print({k: round(v[1], 3) for k, v in m.timer.summary().items()})
