"""The oracles on the reference's own 50-step schedule at the dimensions of BASELINE configs 2, 4 and 5, against the
trajectories the reference itself produced (tests/golden/schedule_*.npz, minted by make_golden.py::schedule_trajectory;
inputs from seeds: tests/golden/schedule_inputs.py).  Pins the checker over a whole annealed run -- temperature ramp, data
truncation from the second step on -- not just over single steps.  CPU only; the GPU twin is test_schedule_golden_gpu.py."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _schedule(steps):
    from prosper_amd.em.annealing import LinearAnnealing
    an = LinearAnnealing(steps)
    an["T"] = [(0, 2.), (.7, 1.)]
    an["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    an["anneal_prior"] = False
    return an


@pytest.mark.parametrize("name,kind,upto", [("bsc_c2", "bsc", 11), ("gsc_c4", "gsc", 11), ("mca_c5", "mca", 50),
                                            ("mmca", "mmca", 11), ("dsc", "dsc", 50), ("tsc", "tsc", 50)])
def test_oracle_follows_the_reference_schedule(name, kind, upto):
    from schedule_inputs import schedule_inputs
    from oracle import bsc_oracle as B, gsc_oracle as G, mca_oracle as M, mmca_oracle as MM, dsc_oracle as DS, tsc_oracle as TS
    from schedule_inputs import DSC_STATES
    z = np.load(os.path.join(HERE, "golden", "schedule_%s.npz" % name))
    D, H, Hp, gamma, N, steps = (int(z[k]) for k in ("D", "H", "Hprime", "gamma", "N", "steps"))
    y, p = schedule_inputs(kind, D, H, N, int(z["seed"]))
    O = {"bsc": B, "gsc": G, "mca": M, "mmca": MM, "dsc": DS, "tsc": TS}[kind]
    if kind == "dsc":
        model = DS.make_model(D, H, Hp, gamma, DSC_STATES.copy())
    else:
        model = {"gsc": G, "tsc": TS}.get(kind, B).make_model(D, H, Hp, gamma)
    if kind == "bsc":
        p = dict(p, mu=np.zeros(D))
    an = _schedule(steps)
    keep = [int(k) for k in z["keep"]]
    tol = {"bsc": 1e-11, "gsc": 1e-9, "mca": 1e-7, "mmca": 1e-7, "dsc": 1e-9, "tsc": 1e-9}[kind]
    for it in range(upto):
        A = B.Anneal(T=an["T"], Ncut_factor=an["Ncut_factor"], anneal_prior=False)
        out = O.em_step(A, model, p, y, stats_fn=B.m_step_stats_vec, vec=True) if kind == "bsc" else O.em_step(A, model, p, y)
        p = out[0] if isinstance(out, tuple) else out
        an.next()
        for k in ("pi", "sigma", "mu", "sigma_sq"):
            if k in z.files:
                np.testing.assert_allclose(np.asarray(p[k]), z[k][it], rtol=tol, atol=tol * float(np.abs(z[k][it]).max()),
                                           err_msg="%s after step %d" % (k, it))
        if it in keep:
            for k in ("W", "psi_sq"):
                if k in z.files:
                    ref = z[k][keep.index(it)]
                    d = float(np.abs(np.asarray(p[k]) - ref).max() / np.abs(ref).max())
                    assert d <= tol, "%s after step %d: %.2e of its largest entry" % (k, it, d)
