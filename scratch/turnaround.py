"""Host turnaround between the M-step download and the next scores-GEMM launch (BSC config 2)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
exec(open('scratch/em_loop.py').read().split("for _ in range(3)")[0])
from prosper_amd.em.camodels import bsc_et, _device
stamps = []
def wrap(cls, name, tag):
    f = getattr(cls, name)
    def g(self, *a, **k):
        stamps.append((tag + ":in", time.perf_counter()))
        r = f(self, *a, **k)
        stamps.append((tag + ":out", time.perf_counter()))
        return r
    setattr(cls, name, g)
wrap(_device.DeviceCAModel, "_download", "download")
wrap(bsc_et.BSC_ET, "_scores_chunk", "gemm_launch")
wrap(bsc_et.BSC_ET, "_finalize", "finalize")
wrap(bsc_et.BSC_ET, "M_step", "M_step")
wrap(bsc_et.BSC_ET, "select_Hprimes", "select")
wrap(bsc_et.BSC_ET, "_same_W", "sameW")
wrap(bsc_et.BSC_ET, "E_step", "E_step")
q = dict(p)
for _ in range(6):
    q = m.step(an, q, data)
torch.cuda.synchronize()
# print the last iteration boundary
idx = [i for i, s in enumerate(stamps) if s[0] == "download:out"]
i0 = idx[-2]
t0 = stamps[i0][1]
for tag, t in stamps[i0 - 1:idx[-1] + 1]:
    print("%9.1f us  %s" % ((t - t0) * 1e6, tag))
