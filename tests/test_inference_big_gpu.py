"""CAModel.inference (camodels/__init__.py:256-375) on thousands of datapoints against the reference's own output
(tests/golden/{bsc,mca,gsc}_inference_big.npz, minted by make_golden.py::inference_big_case; inputs from seeds,
tests/golden/schedule_inputs.py): plain top-K and the adaptive H' / gamma growth -- a fifth to a third of the datapoints are
re-run with a larger state table.  The 60-datapoint goldens of the per-model test files cover the same code; this one covers
its ties and exits at size."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.mark.parametrize("kind", ["bsc", "mca", "gsc"])
@pytest.mark.parametrize("tag", ["plain", "capped"])
def test_inference_at_size_matches_the_reference(kind, tag):
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    from schedule_inputs import schedule_inputs
    from prosper_amd.em.annealing import LinearAnnealing
    g = np.load(os.path.join(HERE, "golden", "%s_inference_big.npz" % kind))
    D, H, Hp, gamma, N = (int(g[k]) for k in ("D", "H", "Hprime", "gamma", "N"))
    y, p0 = schedule_inputs(kind, D, H, N, int(g["seed"]))
    if kind == "bsc":
        from prosper_amd.em.camodels.bsc_et import BSC_ET
        m = BSC_ET(D, H, Hp, gamma)
    elif kind == "mca":
        from prosper_amd.em.camodels.mca_et import MCA_ET
        m = MCA_ET(D, H, Hp, gamma)
    else:
        from prosper_amd.em.camodels.gsc_et import GSC
        m = GSC(D, H, Hp, gamma, sigma_sq_type="scalar")
    anneal = LinearAnnealing(1)
    anneal["T"] = [(0, 1.)]
    anneal["anneal_prior"] = False
    kw = dict(topK=4, adaptive=False) if tag == "plain" else dict(topK=3, adaptive=True, Hprime_max=Hp + 1, gamma_max=gamma + 1)
    res = m.inference(anneal, {k: np.array(v, copy=True) for k, v in p0.items()}, {"y": y}, **kw)
    assert (m.Hprime, m.gamma) == (Hp, gamma)
    assert np.array_equal(res["gamma"], g[tag + "_gamma"]) and np.array_equal(res["Hprime"], g[tag + "_Hprime"])
    np.testing.assert_allclose(res["p"], g[tag + "_p"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(res["m"], g[tag + "_m"], rtol=1e-8, atol=1e-12)
    # the states: equal, except where two of a datapoint's top-K have the same probability (argsort's order there is the
    # reference's accident)
    diff = np.nonzero((res["s"] != g[tag + "_s"]).any(axis=2))
    for n, k in zip(*diff):
        same = np.isclose(g[tag + "_p"][n], g[tag + "_p"][n, k], rtol=1e-12, atol=0).sum()
        assert same >= 2, "datapoint %d: state %d differs without a tie" % (n, k)
    assert len(diff[0]) <= N // 100
    if tag == "capped":
        assert (g["capped_Hprime"] > Hp).sum() > N // 10
