"""E-step passes (select_Hprimes + E_step, parameters resident, new W^T every pass -- bench.py's estep_pass) over shapes the
reference accepts but BASELINE does not name: datapoints/s, fraction of the f64 MFMA roof of the scores GEMM, code path."""
import os, sys, time, gc, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosper_amd.em.camodels.bsc_et import BSC_ET
from prosper_amd.em.camodels._device import KernelTimer
class An(dict):
    crit_params = []
    def __missing__(s, k): return 0.0
    def as_dict(s): return dict(s)
dev = torch.device("cuda", 0)
shapes = [(1024, 256, 8, 4), (256, 128, 6, 3), (1024, 256, 6, 3), (1024, 256, 10, 4), (784, 400, 8, 3), (1024, 512, 8, 4), (4096, 1024, 10, 3)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (D, H, Hp, gamma) in shapes:
    N = int(min(200_000, (1.2e9 / (8 * D)) // 1024 * 1024))
    g = torch.Generator(device=dev).manual_seed(0)
    W_gt = torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)
    Y = torch.empty(N, D, dtype=torch.float64, device=dev)
    step = 25_000
    for lo in range(0, N, step):
        n = min(step, N - lo)
        S = (torch.rand(n, H, generator=g, device=dev) < 4.0 / H).to(torch.float64)
        Y[lo:lo + n] = S @ W_gt.t() + torch.randn(n, D, generator=g, device=dev, dtype=torch.float64)
    Wt_dev = (W_gt + 0.1 * torch.randn(D, H, generator=g, device=dev, dtype=torch.float64)).t().contiguous()
    Wt_host = Wt_dev.cpu().numpy()
    params = {"W": Wt_host.T, "pi": 4.0 / H, "sigma": 1.0}
    try:
        m = BSC_ET(D, H, Hp, gamma)
        data = {"y": Y}
        an = An(T=1.0)
        def estep_pass():
            m.install_parameters(data, Wt_dev, Wt_host)
            return m.E_step(an, params, m.select_Hprimes(params, data))
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.3:
            estep_pass(); torch.cuda.synchronize()
        gc.disable()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps = 10
        for _ in range(steps): estep_pass()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
        gc.enable()
        m.timer = kt = KernelTimer()
        for _ in range(2): estep_pass()
        m.timer = None
        ks = {k: round(v[1], 4) for k, v in kt.summary().items()}
        path = "fused8 (16-wavefront tile)" if m._tile8_whole_shard() and m._fused() else "fused (4-wavefront tile)" if m._fused() else \
            ("two kernels: GEMM + rows16" if "select_estep" in ks else "two kernels: GEMM + select + estep")
        dps = N / (ms * 1e-3)
        print(json.dumps({"shape": "D=%d H=%d H'=%d gamma=%d N=%d" % (D, H, Hp, gamma, N), "K": 1 + H + m.no_states, "ms": round(ms, 4),
                          "dp_per_s": round(dps), "estep_mfma_frac": round(dps * 2 * D * H / 78.6e12, 4),
                          "estep_hbm_frac": round(dps * 8 * (D + 1 + H + m.no_states) / 8e12, 4), "path": path, "kernels_ms": ks}))
    except Exception as e:
        print(json.dumps({"shape": "D=%d H=%d H'=%d gamma=%d" % (D, H, Hp, gamma), "error": repr(e)[:300]}))
    del Y
    torch.cuda.empty_cache()
