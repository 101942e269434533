"""Phase timeline of gsc_estep_kernel (a -DPM_GSC_STAMPS build, PM_LIB_PATH): per datapoint of wavefront 0 of 32 workgroups,
microseconds between the stamps: 0 start | 1 selection done | 2 candidate gathers landed (vmcnt 0) | 3 pair atomics sent |
4 multi-cause loop done | 5 singletons done | 6 expectations done | 7 outputs stored (issued) | 8 lists written."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ['PM_LIB_PATH'])
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gsc_estep_time.py")).read())

lib = _lib.load()

# the stamps live in a __device__ array of the library: fetch through hipMemcpyFromSymbol's underlying API


get = lib.pm_gsc_stamps_get
get.argtypes = [ctypes.c_void_p]; get.restype = ctypes.c_int
buf = np.zeros((32, 12, 10), dtype=np.uint64)
assert get(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.float64) / 100.0          # microseconds
d = np.diff(t[:, :, :9], axis=2)            # (wg, dp, 8 phases)
ok = (buf[:, :, 8] > 0) & (buf[:, :, 0] > 0)
names = ["select", "gather wait", "pair atomics", "multi-cause", "singletons", "expectations", "stores", "lists"]
print("datapoints sampled", int(ok.sum()))
for k, nm in enumerate(names):
    v = d[:, :, k][ok]
    print("%-14s mean %7.2f us  median %7.2f  p90 %7.2f" % (nm, v.mean(), np.median(v), np.percentile(v, 90)))
tot = (t[:, :, 8] - t[:, :, 0])[ok]
print("datapoint total mean %.2f us median %.2f" % (tot.mean(), np.median(tot)))
gap = (t[:, 1:, 0] - t[:, :-1, 8])[ok[:, 1:] & ok[:, :-1]]
print("between datapoints mean %.2f us" % gap.mean())
