"""GSC (scalar, diagonal and full sigma_sq) parity on the GPU: HIP path vs golden vectors minted from the reference
(tests/golden/gsc_step_*.npz, mapped back from the reference's bucket order to datapoint order)
and vs the oracle.  float64 kernels; the H x H inverse in the M-step amplifies rounding by the
conditioning of sum xpt_szsz, hence 1e-7 on the parameters (BASELINE asks 1e-4)."""
import glob
import os

import numpy as np
import pytest

from conftest import golden, GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class _An(dict):
    crit_params = []

    def __missing__(self, k):
        return 0.0

    def as_dict(self):
        return dict(self)


def _cases():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "gsc_step_*.npz")))


def _params(g):
    sig = g["sigma_sq"]
    return {"W": g["W"].copy(), "pi": g["pi"].copy(), "mu": g["mu"].copy(), "psi_sq": g["psi_sq"].copy(),
            "sigma_sq": float(sig) if sig.ndim == 0 else sig.copy()}


def _kind(g):
    return str(g["sigma_type"]) if "sigma_type" in g else "scalar"


@pytest.mark.parametrize("case", _cases())
def test_gsc_step_matches_reference_golden(case):
    assert torch.cuda.is_available()
    from prosper_amd.em.camodels.gsc_et import GSC
    g = golden(case)
    m = GSC(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]), _kind(g))
    assert np.array_equal(m.state_matrix, g["state_matrix"])
    an = _An(T=float(g["T"]))
    params = _params(g)
    data = m.select_Hprimes(params, {"y": g["y"]})
    suff = m.E_step(an, params, data)
    assert np.array_equal(np.asarray(data["candidates"]).astype(np.int64), g["candidates"])
    np.testing.assert_allclose(np.asarray(suff["xpt_s"]), g["xpt_s"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(np.asarray(suff["xpt_sz"]), g["xpt_sz"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(suff["xpt_ss"].sum(axis=0).cpu().numpy(), g["sum_xpt_ss"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(suff["xpt_szsz"].sum(axis=0).cpu().numpy(), g["sum_xpt_szsz"], rtol=1e-9, atol=1e-10)
    new = m.M_step(an, params, suff, data)
    for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
        ref = g[k + "_new"]
        np.testing.assert_allclose(new[k], ref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=k)
    # candidates on their own, and candidates handed in
    assert np.array_equal(np.asarray(m.candidates(_params(g), {"y": g["y"]})), g["candidates"])
    suff2 = m.E_step(an, _params(g), {"y": g["y"], "candidates": g["candidates"]})
    np.testing.assert_allclose(np.asarray(suff2["xpt_sz"]), g["xpt_sz"], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("D,H,Hp,gamma,N,T", [(256, 128, 6, 3, 1000, 1.0), (60, 50, 5, 4, 333, 1.3), (20, 10, 3, 2, 70, 1.0)])
def test_gsc_step_matches_oracle(D, H, Hp, gamma, N, T):
    from oracle import gsc_oracle as G
    from prosper_amd.em.camodels.gsc_et import GSC
    rng = np.random.RandomState(D + H + N)
    gt = {"W": rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.5), "psi_sq": np.eye(H),
          "sigma_sq": 1.0}
    y, _, _ = G.generate_gsc_data(gt, N, rng)
    Q = 0.05 * rng.normal(size=(H, H))
    params = {"W": gt["W"] + 0.1 * rng.normal(size=(D, H)), "pi": np.clip(gt["pi"] * rng.uniform(0.8, 1.3, size=H), 0.01, 0.9),
              "mu": gt["mu"] + 0.1 * rng.normal(size=H), "psi_sq": np.diag(rng.uniform(0.7, 1.4, size=H)) + Q @ Q.T,
              "sigma_sq": 1.2}
    model = G.make_model(D, H, Hp, gamma)
    an = G.Anneal(T=T)
    ref, log = G.em_step(an, model, {k: np.array(v, copy=True) for k, v in params.items()}, y)
    m = GSC(D, H, Hp, gamma, "scalar")
    new = m.step(_An(T=T), {k: np.array(v, copy=True) for k, v in params.items()}, {"y": y})
    cond = np.linalg.cond(log["suff"]["xpt_szsz"].sum(0))
    tol = max(1e-8, 50 * cond * np.finfo(float).eps)
    for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
        np.testing.assert_allclose(new[k], ref[k], rtol=10 * tol, atol=tol * max(1.0, np.abs(ref[k]).max()), err_msg=k)


def test_gsc_unknown_noise_type_raises():
    from prosper_amd import _lib
    from prosper_amd.em.camodels.gsc_et import GSC
    m = GSC(8, 4, 3, 2, "banded")
    with pytest.raises(_lib.HipError):
        m.select_Hprimes({}, {"y": np.zeros((2, 8))})


@pytest.mark.parametrize("kind", ["diagonal", "full"])
@pytest.mark.parametrize("D,H,Hp,gamma,N,T", [(96, 40, 5, 3, 600, 1.0), (30, 12, 4, 4, 257, 1.25)])
def test_gsc_noise_types_match_oracle(kind, D, H, Hp, gamma, N, T):
    """Diagonal / full noise covariance: Sigma^-1-weighted scores, Gram matrix and norms through the same
    kernel (sigma^2 = 1), per-type sigma_sq update (gsc_et.py:677-701)."""
    from oracle import gsc_oracle as G
    from prosper_amd.em.camodels.gsc_et import GSC
    rng = np.random.RandomState(D + H + N)
    gt = {"W": rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.5), "psi_sq": np.eye(H),
          "sigma_sq": 1.0}
    y, _, _ = G.generate_gsc_data(gt, N, rng)
    Q = 0.05 * rng.normal(size=(H, H))
    sig = rng.uniform(0.7, 1.6, size=D)
    if kind == "full":
        Qs = 0.1 * rng.normal(size=(D, D))
        sig = np.diag(sig) + Qs @ Qs.T
    params = {"W": gt["W"] + 0.1 * rng.normal(size=(D, H)), "pi": np.clip(gt["pi"] * rng.uniform(0.8, 1.3, size=H), 0.01, 0.9),
              "mu": gt["mu"] + 0.1 * rng.normal(size=H), "psi_sq": np.diag(rng.uniform(0.7, 1.4, size=H)) + Q @ Q.T,
              "sigma_sq": sig}
    model = G.make_model(D, H, Hp, gamma)
    ref, log = G.em_step(G.Anneal(T=T), model, {k: np.array(v, copy=True) for k, v in params.items()}, y)
    m = GSC(D, H, Hp, gamma, kind)
    p = {k: np.array(v, copy=True) for k, v in params.items()}
    data = m.select_Hprimes(p, {"y": y})
    suff = m.E_step(_An(T=T), p, data)
    assert np.array_equal(np.asarray(data["candidates"]).astype(np.int64), log["candidates"])
    np.testing.assert_allclose(np.asarray(suff["xpt_s"]), log["suff"]["xpt_s"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(np.asarray(suff["xpt_sz"]), log["suff"]["xpt_sz"], rtol=1e-8, atol=1e-12)
    new = m.M_step(_An(T=T), p, suff, data)
    cond = np.linalg.cond(log["suff"]["xpt_szsz"].sum(0))
    tol = max(1e-8, 50 * cond * np.finfo(float).eps)
    for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
        np.testing.assert_allclose(new[k], ref[k], rtol=tol, atol=tol * max(1.0, np.abs(ref[k]).max()), err_msg=k)
    assert np.shape(new["sigma_sq"]) == ((D,) if kind == "diagonal" else (D, D))
