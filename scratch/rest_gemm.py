import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prosper_amd import _lib
dev = torch.device("cuda", 0)
M, N, K = 3392, 256, 1024
a = torch.randn(M, K, dtype=torch.float64, device=dev); b = torch.randn(N, K, dtype=torch.float64, device=dev)
c = torch.empty(M, N, dtype=torch.float64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(20): _lib.call("pm_gemm_nt_f64", p(a), K, p(b), K, p(c), N, M, N, K, st)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): _lib.call("pm_gemm_nt_f64", p(a), K, p(b), K, p(c), N, M, N, K, st)
torch.cuda.synchronize()
print("nsplit", os.environ.get("PM_GEMM_NSPLIT", "auto"), "us per call %.1f" % ((time.perf_counter() - t) / 200 * 1e6), "err", float((c - a @ b.t()).abs().max()))
