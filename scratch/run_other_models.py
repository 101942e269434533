import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
class Anneal(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)
for _ in range(2):
    print(bench.other_models(torch.device('cuda', 0), Anneal))
