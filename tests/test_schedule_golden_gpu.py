"""The reference's OWN annealing schedule (examples/barstests/bars-learning.py:77-80: T = [(0, 2.), (.7, 1.)], Ncut_factor =
[(0, 0.), (2/3, 1.)], 50 steps) at the dimensions of BASELINE configs 2, 4 and 5 (and MMCA at config 5's, DSC / TSC at D = 128, H = 64), against trajectories the reference itself
produced here (tests/golden/make_golden.py::schedule_trajectory; inputs re-created from seeds, tests/golden/schedule_inputs.py).
The drop-in loop ``EM(model, anneal).run()`` runs on the fast path -- next E-step launched by the M-step across the ramp,
deferred statistics on the truncation steps -- and has to follow the reference step by step: the same N_use at every step,
free energies, scalar / vector parameters at every step and the matrices after steps 10 and 49 within the tolerances below (50 steps of rounding-level differences through a
discontinuous map -- candidate selection, the cut; measured: 4e-15 / 7e-13 / 3e-11 of W's largest entry for BSC / GSC / MCA)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _run(kind, z, fast):
    from schedule_inputs import schedule_inputs
    from prosper_amd.em import EM
    from prosper_amd.em.annealing import LinearAnnealing
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    D, H, Hp, gamma, N, steps = (int(z[k]) for k in ("D", "H", "Hprime", "gamma", "N", "steps"))
    y, p0 = schedule_inputs(kind, D, H, N, int(z["seed"]))
    if kind == "bsc":
        from prosper_amd.em.camodels.bsc_et import BSC_ET as cls
        m = cls(D, H, Hp, gamma)
    elif kind == "mca":
        from prosper_amd.em.camodels.mca_et import MCA_ET as cls
        m = cls(D, H, Hp, gamma)
    elif kind == "mmca":
        from prosper_amd.em.camodels.mmca_et import MMCA_ET as cls
        m = cls(D, H, Hp, gamma)
    elif kind == "dsc":
        from schedule_inputs import DSC_STATES
        from prosper_amd.em.camodels.dsc_et import DSC_ET
        m = DSC_ET(D, H, Hp, gamma, states=DSC_STATES.copy())
    elif kind == "tsc":
        from prosper_amd.em.camodels.tsc_et import TSC_ET
        m = TSC_ET(D, H, Hp, gamma)
    else:
        from prosper_amd.em.camodels.gsc_et import GSC
        m = GSC(D, H, Hp, gamma, sigma_sq_type="scalar")
    if not fast:
        m.speculate_estep = False
        if hasattr(m, "defer_stats"):
            m.defer_stats = False
    an = LinearAnnealing(steps)
    an["T"] = [(0, 2.), (.7, 1.)]
    an["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    an["anneal_prior"] = False
    names = tuple(p0.keys()) + ("L", "N_use")
    h = dlog.set_handler(names, StoreInMemory)
    try:
        em = EM(model=m, anneal=an, data={"y": y}, lparams={k: np.array(v, copy=True) for k, v in p0.items()})
        em.run()
    finally:
        dlog.remove_handler(h)
    return m, {k: np.array(h.tables[k]) for k in names if k in h.tables}


@pytest.mark.parametrize("name,kind,fast", [("bsc_c2", "bsc", True), ("bsc_c2", "bsc", False), ("gsc_c4", "gsc", True),
                                            ("mca_c5", "mca", True), ("mca_c5", "mca", False), ("mmca", "mmca", True),
                                            ("dsc", "dsc", True), ("tsc", "tsc", True)])
def test_em_run_follows_the_reference_on_its_own_schedule(name, kind, fast):
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    path = os.path.join(HERE, "golden", "schedule_%s.npz" % name)
    z = np.load(path)
    steps, N = int(z["steps"]), int(z["N"])
    m, got = _run(kind, z, fast)
    if "N_use" in got and len(z["N_use"]) == steps:
        np.testing.assert_array_equal(got["N_use"].astype(np.int64), z["N_use"].astype(np.int64))
    if len(z["L"]) == steps:
        ok = np.isfinite(z["L"])
        assert ok.sum() >= steps - 2, "the fixture is meant to stay in the reference's representable range"
        np.testing.assert_allclose(got["L"][ok], z["L"][ok], rtol=1e-10)
    # (measured for BSC / GSC / MCA: 4e-15, 7e-13, 3e-11 of the largest entry after 50 steps)
    tol = {"bsc": 1e-11, "gsc": 1e-9, "mca": 1e-7, "mmca": 1e-7, "dsc": 1e-9, "tsc": 1e-9}[kind]
    for k in ("pi", "sigma", "mu", "sigma_sq"):
        if k in z.files:
            np.testing.assert_allclose(got[k], z[k], rtol=tol, atol=tol * float(np.abs(z[k]).max()), err_msg=k)
    dev = {}
    for i, step in enumerate(z["keep"]):            # the matrices after two of the steps (megabytes each)
        for k in ("W", "psi_sq"):
            if k in z.files:
                ref = z[k][i]
                dev[k, int(step)] = float(np.abs(got[k][int(step)] - ref).max() / np.abs(ref).max())
                assert dev[k, int(step)] <= tol, "%s after step %d: %.2e of its largest entry" % (k, step, dev[k, int(step)])
    print("%s (fast=%s): max |dev| / max |ref| %s; N_use[-1] %d" % (name, fast, dev, int(got["N_use"][-1]) if "N_use" in got else N))
    if fast and kind == "bsc":
        assert m.spec_hits >= steps - 6, m.spec_hits          # the ramp and the truncation steps stay on the fast path
