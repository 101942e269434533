"""GSC: steps 0..2 teacher-forced against the oracle, with the pipeline features toggled (single process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import gsc_oracle as GO
from prosper_amd.em.camodels.gsc_et import GSC


class An(dict):
    crit_params = []
    def __missing__(self, k): return 0.0
    def as_dict(self): return dict(self)

D, H, Hp, gamma, N = 256, 128, 6, 3, 1001
rng = np.random.RandomState(29)
gt = {"W": rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.5), "psi_sq": np.eye(H), "sigma_sq": 1.0}
yg, _, _ = GO.generate_gsc_data(gt, N, rng)
Q = 0.05 * rng.normal(size=(H, H))
p0 = {"W": gt["W"] + 0.1 * rng.normal(size=(D, H)),
      "pi": np.clip(gt["pi"] * rng.uniform(0.8, 1.3, size=H), 0.01, 0.9), "mu": gt["mu"] + 0.1 * rng.normal(size=H),
      "psi_sq": np.diag(rng.uniform(0.7, 1.4, size=H)) + Q @ Q.T, "sigma_sq": 1.2}
gkeys = ("W", "pi", "mu", "psi_sq", "sigma_sq")
gmodel = GO.make_model(D, H, Hp, gamma)
for label, env in (("default", {}), ("nospec", {"spec": False}), ("nowarm", {"PM_WARM_INVERSE": "0"}),
                   ("nospec_estep", {"spec_estep": False})):
    for k, v in env.items():
        if k.startswith("PM_"): os.environ[k] = v
    mg = GSC(D, H, Hp, gamma, "scalar")
    if env.get("spec") is False: mg.speculate = False
    if env.get("spec_estep") is False: mg.speculate_estep = False
    p = {k: np.array(v, copy=True) for k, v in p0.items()}
    for step, T in enumerate([1.1, 1.0, 1.0]):
        start = {k: np.array(v, copy=True) for k, v in p.items()}
        p = mg.step(An(T=T), p, {"y": yg})
        ref, log = GO.em_step(GO.Anneal(T=T), gmodel, start, yg)
        print(label, "step", step, " ".join("%s %.2e" % (k, np.abs(np.asarray(p[k]) - ref[k]).max()) for k in gkeys), flush=True)
        p = {k: np.array(p[k], copy=True) for k in gkeys}
    os.environ.pop("PM_WARM_INVERSE", None)
