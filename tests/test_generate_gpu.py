"""Device-side data generation (``generate_data(..., device=True)``): the draws come from
the device's Philox streams, so they cannot match the reference's NumPy stream value by value; they are checked
statistically (moments of the latents, exact superposition of the returned latents, noise level) and for
reproducibility under a seed.  The host path keeps the reference's stream (golden-tested elsewhere)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    return torch.device("cuda", 0)


N = 60000


def _noise_ok(resid, sigma):
    assert abs(resid.mean()) < 5 * sigma / np.sqrt(resid.size)
    assert abs(resid.std() / sigma - 1.0) < 0.01


def test_bsc_and_reproducibility(dev):
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H = 96, 40
    rng = np.random.RandomState(0)
    p = {"W": rng.normal(size=(D, H)), "pi": 0.07, "sigma": 0.8}
    m = BSC_ET(D, H, 5, 3)
    a = m.generate_data(p, N, device=True, seed=11)
    b = m.generate_data(p, N, device=True, seed=11)
    c = m.generate_data(p, N, device=True, seed=12)
    ya, sa = np.asarray(a["y"]), np.asarray(a["s"])
    assert np.array_equal(ya, np.asarray(b["y"])) and not np.array_equal(ya, np.asarray(c["y"]))
    assert sa.shape == (N, H) and ya.shape == (N, D)
    assert abs(sa.mean() - 0.07) < 5 * np.sqrt(0.07 * 0.93 / sa.size)
    _noise_ok(ya - sa.astype(float) @ p["W"].T, 0.8)
    # the handles feed the hot path directly
    new = m.step(type("A", (dict,), {"crit_params": [], "__missing__": lambda s, k: 0.0, "as_dict": lambda s: dict(s)})(T=1.0),
                 dict(p), {"y": a["y"]})
    assert np.isfinite(new["W"]).all()


def test_mca_max_superposition(dev):
    from prosper_amd.em.camodels.mca_et import MCA_ET
    D, H = 64, 24
    rng = np.random.RandomState(1)
    p = {"W": np.abs(rng.normal(size=(D, H))) * 2 + 0.1, "pi": 0.1, "sigma": 0.5}
    d = MCA_ET(D, H, 5, 3).generate_data(p, N, device=True, seed=3)
    y, s = np.asarray(d["y"]), np.asarray(d["s"]).astype(bool)
    clean = np.where(s[:, None, :], p["W"][None, :, :], -np.inf).max(axis=2).clip(min=0.0)
    assert abs(s.mean() - 0.1) < 5 * np.sqrt(0.09 / s.size)
    _noise_ok(y - clean, 0.5)


def test_mmca_largest_magnitude(dev):
    from prosper_amd.em.camodels.mmca_et import MMCA_ET
    D, H = 48, 20
    rng = np.random.RandomState(2)
    p = {"W": rng.normal(size=(D, H)) * 3, "pi": 0.12, "sigma": 0.7}
    d = MMCA_ET(D, H, 5, 3).generate_data(p, 20000, device=True, seed=4)
    y, s = np.asarray(d["y"]), np.asarray(d["s"]).astype(float)
    t0 = s[:, :, None] * p["W"].T[None, :, :]
    idx = np.abs(t0).argmax(axis=1)
    clean = np.take_along_axis(t0, idx[:, None, :], axis=1)[:, 0, :]
    _noise_ok(y - clean, 0.7)


def test_dsc_and_tsc_latent_values(dev):
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    D, H = 40, 16
    rng = np.random.RandomState(3)
    W = rng.normal(size=(D, H)) * 2
    states, pi = np.array([-2., 0., 1., 3.]), np.array([0.05, 0.8, 0.1, 0.05])
    d = DSC_ET(D, H, 4, 2, states=states).generate_data({"W": W, "pi": pi, "sigma": 0.6}, N, device=True, seed=5)
    y, s = np.asarray(d["y"]), np.asarray(d["s"])
    for v, pk in zip(states, pi):
        assert abs((s == v).mean() - pk) < 5 * np.sqrt(pk * (1 - pk) / s.size)
    _noise_ok(y - s @ W.T, 0.6)
    t = TSC_ET(D, H, 4, 2).generate_data({"W": W, "pi": 0.2, "sigma": 0.9}, N, device=True, seed=6)
    y, s = np.asarray(t["y"]), np.asarray(t["s"])
    assert set(np.unique(s)) <= {-1.0, 0.0, 1.0}
    assert abs((s == 1).mean() - 0.1) < 0.002 and abs((s == -1).mean() - 0.1) < 0.002
    _noise_ok(y - s @ W.T, 0.9)


def test_gsc_slab_moments(dev):
    from prosper_amd.em.camodels.gsc_et import GSC
    D, H = 32, 12
    rng = np.random.RandomState(4)
    Q = 0.2 * rng.normal(size=(H, H))
    p = {"W": rng.normal(size=(D, H)), "pi": np.full(H, 0.15), "mu": rng.normal(size=H) + 1.0,
         "psi_sq": np.diag(rng.uniform(0.5, 1.5, size=H)) + Q @ Q.T, "sigma_sq": 0.49}
    d = GSC(D, H, 4, 3, 'scalar').generate_data(p, N, device=True, seed=7)
    y, s, z = np.asarray(d["y"]), np.asarray(d["s"]).astype(bool), np.asarray(d["z"])
    live = (s * np.arange(H)[None, :]).sum(axis=1) != 0            # the upstream quirk: those rows stay zero
    assert (y[~live] == 0).all() and (z[~live] == 0).all()
    assert (z[~s] == 0).all()
    for h in (1, 5):
        zh = z[s[:, h] & live, h]
        assert abs(zh.mean() - p["mu"][h]) < 5 * np.sqrt(p["psi_sq"][h, h] / zh.size)
        assert abs(zh.var() / p["psi_sq"][h, h] - 1.0) < 0.05
    both = s[:, 2] & s[:, 7] & live
    cov = np.cov(z[both, 2], z[both, 7])[0, 1]
    assert abs(cov - p["psi_sq"][2, 7]) < 0.1
    _noise_ok((y - z @ p["W"].T)[live], 0.7)
