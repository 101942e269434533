"""FETCH_SIZE calibration for the 8-byte-per-lane row loads: select-only pass over a known score matrix
(N x 256 f64 = 2048 B per datapoint), run under rocprofv3 --pmc FETCH_SIZE."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from prosper_amd.em.camodels.bsc_et import BSC_ET
D, H, HP, GAMMA, N = 64, 256, 8, 4, 196608
dev = torch.device('cuda', 0)
m = BSC_ET(D, H, HP, GAMMA)
Y = torch.randn(N, D, dtype=torch.float64, device=dev)
W = np.random.RandomState(0).normal(size=(D, H))
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
data = {"y": Y}
p = {"W": W, "pi": 0.1, "sigma": 1.0}
for _ in range(3):
    m._par = {}
    d = m.select_Hprimes(p, data)
    c = d['candidates'].tensor          # materialise: select-only pass (mode 1)
torch.cuda.synchronize()
print("done", tuple(c.shape))
