"""Statistics GEMM (Wp = E[s]^T . Y, config 2) alone."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosper_amd import _lib
N, H, D = 200000, 256, 1024
dev = torch.device('cuda', 0)
E = torch.rand(N, H, dtype=torch.float64, device=dev)
Y = torch.randn(N, D, dtype=torch.float64, device=dev)
C = torch.zeros(H, D, dtype=torch.float64, device=dev)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(3):
    _lib.call("pm_gemm_tn_acc_f64", p(E), H, p(Y), D, p(C), D, H, D, N, s)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); _lib.call("pm_gemm_tn_acc_f64", p(E), H, p(Y), D, p(C), D, H, D, N, s); b.record()
torch.cuda.synchronize()
print("tn gemm ms", sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2])
