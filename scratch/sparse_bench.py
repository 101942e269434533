"""Times pm_bsc_wp_sparse_f64 alone on synthetic lists of config-2 shape.  PYTHONPATH=. python scratch/sparse_bench.py [nnz]"""
import ctypes
import sys
import numpy as np
import torch
from prosper_amd import _lib

import os
N, H, D = int(os.environ.get("SP_N", 200000)), 256, 1024
nnz = float(sys.argv[1]) if len(sys.argv) > 1 else 3.7
variants = sys.argv[2:]        # standalone builds of bsc_wp_sparse.hip (scratch/sparse_variants.sh); default: the library
dev = torch.device("cuda", 0)
lib = _lib.load()
rng = np.random.RandomState(0)
cnt = np.clip(rng.poisson(nnz, size=N), 0, 16)
idx = np.full((N, 16), 0xFFFF, dtype=np.uint16)
for n in range(N):
    idx[n, :cnt[n]] = rng.choice(H, size=cnt[n], replace=False)
val = rng.random_sample((N, 16))
Y = torch.randn(N, D, dtype=torch.float64, device=dev)
idx_d = torch.from_numpy(idx.view(np.int16)).to(dev)
val_d = torch.from_numpy(val).to(dev)
stats = torch.zeros(lib.pm_bsc_stats_len(H, D), dtype=torch.float64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


fn = lib.pm_bsc_wp_sparse_f64


def run():
    rc = fn(p(idx_d), p(val_d), p(Y), D, p(stats), N, H, D, st)
    assert rc == 0, rc


for path in variants or [None]:
  if path:
    fn = ctypes.CDLL(path).pm_bsc_wp_sparse_f64
    fn.argtypes = lib.pm_bsc_wp_sparse_f64.argtypes
    print(path, end=": ")
  stats.zero_()
  run()
  torch.cuda.synchronize()
  E = np.zeros((N, H))
  m = idx != 0xFFFF
  E[np.repeat(np.arange(N), 16).reshape(N, 16)[m], idx[m].astype(np.int64)] = val[m]
  ref = torch.from_numpy(E).to(dev).t() @ Y
  got = stats[:H * D].view(H, D)
  print("max rel err", float((got - ref).abs().max() / ref.abs().max()))
  for _ in range(20):
      run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50):
      run()
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 50
  print("nnz/row %.2f: %.4f ms per launch, Y stream %.2f TB/s" % (cnt.mean(), ms, N * D * 8 / ms * 1e-9))
