"""Per-workgroup timeline of the fused E-step kernel (diagnostic build with -DPM_FUSED_STAMPS, scratch/fused_stamps.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd import _lib
from prosper_amd.em.camodels.bsc_et import BSC_ET
D, H, Hp, g, N = 1024, 256, 8, 4, int(os.environ.get("N", 200000))
lib = _lib.load()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
Y = torch.empty(N, D, dtype=torch.float64, device=dev)
for lo in range(0, N, 25000):
    S = (torch.rand(min(25000, N - lo), H, generator=gen, device=dev) < 4.0 / H).to(torch.float64)
    Y[lo:lo + 25000] = S @ W_gt.t() + torch.randn(S.shape[0], D, generator=gen, device=dev, dtype=torch.float64)
W0 = (W_gt + 0.1 * torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)).cpu().numpy()
params = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
class An(dict):
    def __missing__(self, k): return 0.0
m = BSC_ET(D, H, Hp, g)
for _ in range(40):
    d = m.select_Hprimes(params, {"y": Y}); m.E_step(An(T=1.0), params, d)
torch.cuda.synchronize()
nb = min(8192, (N + 63) // 64)
buf = (ctypes.c_ulonglong * (8 * nb))()
fn = lib.pm_fused_read_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, nb) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
t0 = st[:, 0].min()
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
nb = st.shape[0]
us5 = (st[:, :5] - t0) / 100.0
us6 = (st[:, 5] - t0) / 100.0
print('select x4 %.1f | fetch issue %.1f | compute x4 %.1f' % ((us5[:,3]-us5[:,2]).mean(), (us6-us5[:,3]).mean(), (us5[:,4]-us6).mean()))
us = us5[:, [0, 1, 4]]
print('phases us (mean): K-loop %.1f | tables+barrier %.1f | select x4 %.1f | estep x4 %.1f' % tuple((us5[:, i + 1] - us5[:, i]).mean() for i in range(4)))          # s_memrealtime ticks at 100 MHz
hw = st[:, 7] & 0xFFFFFFFF
xcc = (st[:, 7] >> 32) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 20 + cu
print("blocks", nb, "CUs seen", len(np.unique(cuid)))
k = us[:, 1] - us[:, 0]; e = us[:, 2] - us[:, 1]
print("K-loop us: mean %.1f min %.1f max %.1f   epilogue us: mean %.1f min %.1f max %.1f" % (k.mean(), k.min(), k.max(), e.mean(), e.min(), e.max()))
print("kernel span us %.1f" % us[:, 2].max())
# timeline of the first CU
c0 = cuid[0]
sel = np.where(cuid == c0)[0]
order = sel[np.argsort(us[sel, 0])]
for b in order[:16]:
    print("cu %d block %5d start %8.1f kend %8.1f end %8.1f" % (c0, b, us[b, 0], us[b, 1], us[b, 2]))
# how many blocks run concurrently per CU at a sample time
for t in (50.0, 400.0, 1000.0):
    live = [(np.sum((us[cuid == c, 0] <= t) & (us[cuid == c, 2] > t))) for c in np.unique(cuid)]
    print("t=%6.0f us: live blocks per CU min %d max %d mean %.2f" % (t, min(live), max(live), np.mean(live)))
# overlap of epilogues on a CU: fraction of epilogue time during which the partner is in its epilogue too
tot = ov = 0.0
for c in np.unique(cuid)[:64]:
    idx = np.where(cuid == c)[0]
    for i in idx:
        for j in idx:
            if i != j:
                lo = max(us[i, 1], us[j, 1]); hi = min(us[i, 2], us[j, 2])
                if hi > lo: ov += hi - lo
        tot += us[i, 2] - us[i, 1]
print("epilogue time overlapped by a partner's epilogue: %.1f %%" % (100 * ov / tot))
