"""pm_gsc_det_quanta_f64 on an idle device: time per call (the deterministic library)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prosper_amd import _lib
H = 128
dev = torch.device("cuda", 0)
G = torch.rand(H, H, dtype=torch.float64, device=dev) + torch.eye(H, dtype=torch.float64, device=dev) * 300
psi = torch.eye(H, dtype=torch.float64, device=dev)
tab = torch.rand(9 * H, dtype=torch.float64, device=dev)
q = torch.empty(16, dtype=torch.float64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def call():
    _lib.call("pm_gsc_det_quanta_f64", p(G), H, p(psi), p(tab), H, 3, ctypes.c_double(5.0), ctypes.c_double(60.0), ctypes.c_double(2e5), p(q), None, det=True)
for _ in range(20): call()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(200): call()
ev1.record(); torch.cuda.synchronize()
print("%.1f us per call (200 back-to-back launches)" % (ev0.elapsed_time(ev1) / 200 * 1e3))
