"""Wall time of the MCA EM step's phases (synchronised after each) at config 5."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_mca.py")).read().split("for _ in range(2)")[0])
acc = {}
def wrap(name):
    f = getattr(m, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(m, name, g)
for n in ("noisify_params", "check_params", "select_partial_data", "select_Hprimes", "E_step", "M_step"):
    wrap(n)
q = dict(p)
for it in range(8):
    if it == 3:
        acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q = m.step(an, q, data)
    torch.cuda.synchronize(); print(it, "step %.2f ms" % ((time.perf_counter() - t0) * 1e3), "pi %.5f sigma %.4f" % (q["pi"], q["sigma"]))
print({k: round(v / 5 * 1e3, 3) for k, v in acc.items()})
