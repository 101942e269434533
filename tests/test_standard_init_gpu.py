"""``standard_init`` of every model against the reference's own output from a seeded NumPy stream
(tests/golden/standard_init_all.npz, make_golden.py::standard_init_cases; BSC has had bsc_init_c1.npz since round 1): the data
moments come from the device (pm_col_moments_f64), the random draws from NumPy in upstream's order."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.mark.parametrize("tag", ["mca", "mmca", "dsc", "tsc", "gsc_scalar", "gsc_diagonal", "gsc_full"])
def test_standard_init_matches_the_reference(tag):
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    from schedule_inputs import schedule_inputs, DSC_STATES
    g = np.load(os.path.join(HERE, "golden", "standard_init_all.npz"))
    kind = tag.split("_")[0]
    y, _ = schedule_inputs(kind, 40, 16, 300, 500)
    if kind == "mca":
        from prosper_amd.em.camodels.mca_et import MCA_ET
        m = MCA_ET(40, 16, 5, 3)
    elif kind == "mmca":
        from prosper_amd.em.camodels.mmca_et import MMCA_ET
        m = MMCA_ET(40, 16, 5, 3)
    elif kind == "dsc":
        from prosper_amd.em.camodels.dsc_et import DSC_ET
        m = DSC_ET(40, 16, 5, 3, states=DSC_STATES.copy())
    elif kind == "tsc":
        from prosper_amd.em.camodels.tsc_et import TSC_ET
        m = TSC_ET(40, 16, 5, 3)
    else:
        from prosper_amd.em.camodels.gsc_et import GSC
        m = GSC(40, 16, 5, 3, sigma_sq_type=tag.split("_")[1])
    np.random.seed(77)
    init = m.standard_init({"y": y.copy()})
    keys = [k[len(tag) + 1:] for k in g.files if k.startswith(tag + "_")]
    assert sorted(keys) == sorted(init.keys())
    for k in keys:
        np.testing.assert_allclose(np.asarray(init[k], dtype=np.float64), g[tag + "_" + k], rtol=1e-12, atol=1e-13, err_msg=k)
