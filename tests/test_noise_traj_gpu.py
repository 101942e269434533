"""EM.run with parameter noise AND partial data against the reference's own trajectory (tests/golden/noise_traj_*.npz,
make_golden.py::noise_trajectory): ``noisify_params`` (em/__init__.py:63-107) and ``select_partial_data``
(camodels/__init__.py:124-152) draw from NumPy's global stream -- the drop-in consumes it in the same order, so that with the
same seed every step sees the same noisy parameters and the same subset of the datapoints."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.mark.parametrize("kind", ["bsc", "mca", "gsc", "mmca", "dsc", "tsc"])
def test_noise_and_partial_data_follow_the_reference_stream(kind):
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    from schedule_inputs import schedule_inputs
    from prosper_amd.em.annealing import LinearAnnealing
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    g = np.load(os.path.join(HERE, "golden", "noise_traj_%s.npz" % kind))
    D, H, Hp, gamma, N, steps, seed = (int(g[k]) for k in ("D", "H", "Hprime", "gamma", "N", "steps", "seed"))
    y, p0 = schedule_inputs(kind, D, H, N, seed)
    if kind == "bsc":
        from prosper_amd.em.camodels.bsc_et import BSC_ET
        m = BSC_ET(D, H, Hp, gamma)
    elif kind == "mca":
        from prosper_amd.em.camodels.mca_et import MCA_ET
        m = MCA_ET(D, H, Hp, gamma)
    elif kind == "mmca":
        from prosper_amd.em.camodels.mmca_et import MMCA_ET
        m = MMCA_ET(D, H, Hp, gamma)
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
    an = LinearAnnealing(steps)
    an["T"] = [(0, 1.6), (.7, 1.)]
    an["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    an["anneal_prior"] = False
    an["partial"] = [(0, .6), (.5, 1.)]
    an["W_noise"] = [(0, .08), (.6, 0.)]
    an["pi_noise"] = [(0, .002), (.6, 0.)]
    an["sigma_noise" if kind != "gsc" else "sigma_sq_noise"] = [(0, .03), (.6, 0.)]
    np.random.seed(1000 + seed)
    lp = {k: np.array(v, copy=True) for k, v in p0.items()}
    h = dlog.set_handler(("N_use",), StoreInMemory)
    hist = {k: [] for k in p0}
    try:
        while not an.finished:                  # (EM.run's body, em/__init__.py:163-178, keeping every step's parameters)
            lp = m.step(an, lp, {"y": y})
            an.next()
            for k in hist:
                hist[k].append(np.array(lp[k], copy=True))
    finally:
        dlog.remove_handler(h)
    if len(g["N_use"]):
        assert np.array_equal(np.array(h.tables["N_use"]).astype(np.int64), g["N_use"].astype(np.int64))
    tol = {"bsc": 1e-9, "mca": 1e-7, "gsc": 1e-7, "mmca": 1e-7, "dsc": 1e-9, "tsc": 1e-9}[kind]
    for k in hist:
        ref = g[k]
        got = np.stack(hist[k])
        np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * float(np.abs(ref).max()), err_msg=k)
