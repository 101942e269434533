"""Every documented limit of the HIP path fails LOUDLY -- `HipError` (PM_ERANGE from the library, or
the host's own range check) -- and never returns a wrong answer or falls back to a CPU path.  The reference itself asserts
only H' <= H and gamma <= H' (camodels/__init__.py:90-91)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class _An(dict):
    crit_params = []

    def __missing__(self, k):
        return 0.0

    def as_dict(self):
        return dict(self)


def _data(D, H, N, seed=0, positive=False):
    rng = np.random.RandomState(seed)
    W = rng.normal(size=(D, H))
    if positive:
        W = np.abs(W) + 0.1
    y = (rng.random_sample((N, H)) < 2.0 / H) @ W.T + rng.normal(size=(N, D))
    return W, y


def _bsc(D, H, Hp, g):
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    W, y = _data(D, H, 40)
    return BSC_ET(D, H, Hp, g), {"W": W, "pi": 2.0 / H, "sigma": 1.0}, y


def _mca(D, H, Hp, g, cls="MCA_ET"):
    import importlib
    mod = importlib.import_module("prosper_amd.em.camodels." + ("mca_et" if cls == "MCA_ET" else "mmca_et"))
    W, y = _data(D, H, 40, positive=True)
    return getattr(mod, cls)(D, H, Hp, g), {"W": W, "pi": 2.0 / H, "sigma": 1.0}, y


def _dsc(D, H, Hp, g, states):
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    W, y = _data(D, H, 40)
    K = len(states)
    pi = np.full(K, 0.5 / (K - 1))
    pi[list(states).index(0.0)] = 0.5
    return DSC_ET(D, H, Hp, g, states=np.array(states)), {"W": W, "pi": pi, "sigma": 1.0}, y


def _tsc(D, H, Hp, g):
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    W, y = _data(D, H, 40)
    return TSC_ET(D, H, Hp, g), {"W": W, "pi": 2.0 / H, "sigma": 1.0}, y


def _gsc(D, H, Hp, g):
    from prosper_amd.em.camodels.gsc_et import GSC
    W, y = _data(D, H, 40)
    return GSC(D, H, Hp, g, "scalar"), {"W": W, "pi": np.full(H, 2.0 / H), "mu": np.ones(H), "psi_sq": np.eye(H),
                                        "sigma_sq": 1.0}, y


BEYOND = [
    ("BSC H > 1024", lambda: _bsc(16, 1100, 4, 2)),
    ("BSC H' > 16", lambda: _bsc(16, 40, 17, 2)),
    ("MCA H > 512", lambda: _mca(16, 600, 4, 2)),
    ("MCA H' > 16", lambda: _mca(16, 40, 17, 2)),
    ("MCA E-step D > 1024", lambda: _mca(1100, 12, 4, 2)),
    ("MMCA H > 512", lambda: _mca(16, 600, 4, 2, "MMCA_ET")),
    ("DSC more than 8 latent values", lambda: _dsc(16, 12, 3, 2, [-4., -3., -2., -1., 0., 1., 2., 3., 4.])),
    ("DSC H > 512", lambda: _dsc(16, 600, 4, 2, [-1., 0., 1.])),
    ("TSC 2 H > 512", lambda: _tsc(16, 300, 4, 2)),
    ("GSC H > 512", lambda: _gsc(16, 600, 4, 2)),
    ("GSC gamma > 8", lambda: _gsc(16, 12, 10, 9)),
]


@pytest.mark.parametrize("what,make", BEYOND, ids=[b[0] for b in BEYOND])
def test_beyond_a_documented_limit_raises(what, make):
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    from prosper_amd import _lib
    with pytest.raises(_lib.HipError):
        model, params, y = make()             # (some limits are checked by the constructor)
        model.step(_An(T=1.0), params, {"y": y})


WITHIN = [
    ("BSC H = 1024 (generic row kernels, blocked inverse)", lambda: _bsc(16, 1024, 4, 2)),
    ("BSC H' = 16", lambda: _bsc(24, 40, 16, 2)),
    ("MCA H' = 16", lambda: _mca(24, 40, 16, 2)),
    ("MCA D = 1024", lambda: _mca(1024, 12, 4, 2)),
    ("DSC 8 latent values", lambda: _dsc(16, 12, 3, 2, [-3., -2., -1., 0., 1., 2., 3., 4.])),
    ("TSC 2 H = 512", lambda: _tsc(16, 256, 4, 2)),
    ("GSC gamma = 8", lambda: _gsc(16, 12, 9, 8)),
]


@pytest.mark.parametrize("what,make", WITHIN, ids=[w[0] for w in WITHIN])
def test_at_a_documented_limit_runs(what, make):
    """... and AT the limit a step runs and returns finite parameters (parity at these shapes: the models' own tests)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    model, params, y = make()
    new = model.step(_An(T=1.0), params, {"y": y})
    assert np.isfinite(np.asarray(new["W"])).all()
