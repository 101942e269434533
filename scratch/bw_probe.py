"""Read bandwidth of a 1.64 GB buffer (torch.sum) fresh after allocation and after a kernel wrote it, and of the kept-rows pattern
of mca_defer_scatter_kernel expressed as an index_select (63 % of 16 KB rows)."""
import torch, time
dev = torch.device("cuda", 0)
N, R = 100000, 2048
x = torch.empty(N, R, dtype=torch.float64, device=dev)
x.normal_()
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
dt = t(lambda: x.sum())
print("sum over 1.64 GB: %.3f ms = %.2f TB/s" % (dt * 1e3, x.numel() * 8 / dt / 1e12))
keep = (torch.rand(N, device=dev) < 0.63).nonzero().flatten()
dt = t(lambda: x.index_select(0, keep).sum())
print("gather 63%% of the rows + sum: %.3f ms (%.2f GB read + written + read)" % (dt * 1e3, keep.numel() * R * 8 / 1e9))
dt = t(lambda: x[keep[:1000]].sum(), 5)
y = x.view(N * 8, 256)
dt = t(lambda: y.sum(dim=0))
print("column sums of the (800k, 256) view: %.3f ms = %.2f TB/s" % (dt * 1e3, x.numel() * 8 / dt / 1e12))
