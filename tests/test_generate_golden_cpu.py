"""``generate_data`` (host path) of BSC, MCA, MMCA, DSC and GSC against the reference's own output from a seeded NumPy stream
(tests/golden/generate_data_all.npz, make_golden.py::generate_data_cases): the host generators consume the stream in upstream's
order (per datapoint: latents, then noise).  No GPU: the device generators (``device=True``) are statistical, tests/test_generate_gpu.py."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.mark.parametrize("kind", ["bsc", "mca", "mmca", "dsc", "gsc"])
def test_generate_data_matches_the_reference_stream(kind):
    from schedule_inputs import schedule_inputs, DSC_STATES
    g = np.load(os.path.join(HERE, "golden", "generate_data_all.npz"))
    _, p0 = schedule_inputs(kind, 24, 10, 8, 600)
    if kind == "bsc":
        from prosper_amd.em.camodels.bsc_et import BSC_ET
        m = BSC_ET(24, 10, 4, 3)
    elif kind == "mca":
        from prosper_amd.em.camodels.mca_et import MCA_ET
        m = MCA_ET(24, 10, 4, 3)
    elif kind == "mmca":
        from prosper_amd.em.camodels.mmca_et import MMCA_ET
        m = MMCA_ET(24, 10, 4, 3)
    elif kind == "dsc":
        from prosper_amd.em.camodels.dsc_et import DSC_ET
        m = DSC_ET(24, 10, 4, 3, states=DSC_STATES.copy())
    else:
        from prosper_amd.em.camodels.gsc_et import GSC
        m = GSC(24, 10, 4, 3, sigma_sq_type="scalar")
    np.random.seed(91)
    data = m.generate_data({k: np.array(v, copy=True) for k, v in p0.items()}, 40)
    keys = [k[len(kind) + 1:] for k in g.files if k.startswith(kind + "_")]
    assert sorted(keys) == sorted(data.keys())
    for k in keys:
        np.testing.assert_allclose(np.asarray(data[k], dtype=np.float64), g[kind + "_" + k].astype(np.float64), rtol=1e-12, atol=1e-13,
                                   err_msg=k)
