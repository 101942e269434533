"""Would GSC's two M-step contractions (Y^T xsz: 256 x 128, [xs|xsz]^T xsz: 256 x 128, K = 200k) run faster as ONE
pm_gemm_tn_acc_f64 over a [Y | xs | xsz] buffer (512 x 128)?"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prosper_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
N, D, H = 200000, 256, 128
big = torch.randn(N, D + 2 * H, device=dev, dtype=torch.float64)
Y = big[:, :D].contiguous()
both = big[:, D:].contiguous()
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.zeros((D + 2 * H) * H, device=dev, dtype=torch.float64)
def two():
    _lib.call("pm_gemm_tn_acc_f64", p(Y), D, ctypes.c_void_p(both.data_ptr() + 8 * H), 2 * H, p(out), H, D, H, N, st)
    _lib.call("pm_gemm_tn_acc_f64", p(both), 2 * H, ctypes.c_void_p(both.data_ptr() + 8 * H), 2 * H, ctypes.c_void_p(out.data_ptr() + 8 * D * H), H, 2 * H, H, N, st)
def one():
    _lib.call("pm_gemm_tn_acc_f64", p(big), D + 2 * H, ctypes.c_void_p(big.data_ptr() + 8 * (D + H)), D + 2 * H, p(out), H, D + 2 * H, H, N, st)
for name, f in (("two launches", two), ("one launch ", one)):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): f()
    torch.cuda.synchronize(); print(name, "%.3f ms" % ((time.perf_counter() - t) / 30 * 1e3))
