"""Probe (no product code): could the 196608 x 1024 x 256 f64 scores product of BSC config 2 run on the narrow
matrix cores, error-free (Ozaki scheme I)?

  Y (N x K) and W^T (K x H) are split into s slices of 7-bit signed integers each (row- resp. column-wise power-of-two scaling):
  a product of two slices summed over K = 1024 needs 14 + 10 = 24 bits -- EXACT in the i32 accumulator of the int8 MFMA.  The
  score is sum_{i + j < s} 2^-(7 (i + j + 2)) (Y_i . W_j) scaled back: s (s + 1) / 2 integer GEMMs, recombined in f64.

Measured here: (1) the accuracy of the emulation against the f64 product for s = 5, 6, 7 (on a row sample; torch int64 matmul on the
host is the exact integer reference), (2) the time of ONE int8 slice-pair GEMM of the full shape through the library (torch._int_mm ->
hipBLASLt), (3) projected totals: pairs x that time (a fused kernel would recombine in registers: no extra traffic), against the
present 1.43 ms K-loop.  Slicing Y is once per shard (the data does not change between EM steps); W^T per step is 0.26 M elements."""
import json, os, sys, time
import numpy as np, torch
dev = torch.device("cuda", 0)
N, K, H = 196608, 1024, 256
g = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(K, H, generator=g, device=dev, dtype=torch.float64)
Y = torch.empty(N, K, dtype=torch.float64, device=dev)
for lo in range(0, N, 32768):
    S = (torch.rand(32768, H, generator=g, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 32768] = S @ W.t() + torch.randn(32768, K, generator=g, device=dev, dtype=torch.float64)

def slices(X, s, dim):
    """X = 2^e * sum_i 2^(-7 (i + 1)) X_i with 7-bit signed integer slices X_i (|X_i| <= 64): exact for the bits it keeps."""
    e = torch.ceil(torch.log2(X.abs().amax(dim=dim, keepdim=True).clamp_min(1e-300))) + 1      # |X| / 2^e < 1/2
    R = X / torch.exp2(e)
    out = []
    for i in range(s):
        R = R * 128.0
        Xi = torch.round(R)                                   # in [-64, 64]
        R = R - Xi
        out.append(Xi.to(torch.int8))
    return e, out

res = {"shape": [N, K, H]}
# (1) accuracy on a row sample (exact integer products on the host side of the check: int64 matmul)
rows = torch.arange(0, N, N // 512, device=dev)[:512]
Ys = Y[rows]
ref = (Ys @ W)
for s in (5, 6, 7):
    ey, Yi = slices(Ys, s, 1)
    ew, Wi = slices(W, s, 0)
    acc = torch.zeros_like(ref)
    for i in range(s):
        for j in range(s - i):
            P = (Yi[i].to(torch.float64) @ Wi[j].to(torch.float64))          # exact: |sum| < 2^24
            acc += P * 2.0 ** (-7 * (i + j + 2))
    emu = acc * torch.exp2(ey) * torch.exp2(ew)
    err = (emu - ref).abs().max().item()
    scale = ref.abs().max().item()
    res["s%d" % s] = {"pairs": s * (s + 1) // 2, "max_abs_err_scores": err, "rel_to_max_score": err / scale}
# (2) one int8 slice-pair GEMM of the full shape through the library
ey, Yi = slices(Y[:4096], 1, 1)
A8 = torch.randint(-64, 65, (N, K), device=dev, dtype=torch.int8)
B8 = torch.randint(-64, 65, (K, H), device=dev, dtype=torch.int8)
for _ in range(5):
    C = torch._int_mm(A8, B8)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    C = torch._int_mm(A8, B8)
e1.record(); torch.cuda.synchronize()
t1 = e0.elapsed_time(e1) / 20
res["int8_gemm_ms"] = t1
res["int8_gemm_Tops"] = 2.0 * N * K * H / (t1 * 1e-3) / 1e12
# the f64 product through the library for scale (the product path's own K-loop: 1.43 ms)
for _ in range(3):
    Cf = Y @ W
torch.cuda.synchronize(); e0.record()
for _ in range(5):
    Cf = Y @ W
e1.record(); torch.cuda.synchronize()
res["f64_library_gemm_ms"] = e0.elapsed_time(e1) / 5
for s in (5, 6, 7):
    res["s%d" % s]["projected_ms_fused"] = res["s%d" % s]["pairs"] * t1
    res["s%d" % s]["vs_1.43ms_kloop"] = res["s%d" % s]["pairs"] * t1 / 1.43
print(json.dumps(res, indent=1))
