"""bench.py's other_shapes on their own (E-step pass at off-config shapes)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
for s in bench.other_shapes(torch.device("cuda", 0), An, budget_s=60.0):
    if s.get("D"):
        print((s["D"], s["H"], s["Hprime"], s["gamma"]), s.get("ms_per_pass"), s.get("estep_mfma_frac"), s.get("kernels_ms"))
