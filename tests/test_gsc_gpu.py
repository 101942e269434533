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
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
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


@pytest.mark.parametrize("case", _cases())
def test_gsc_compute_lpj_matches_reference_golden(case):
    """GSC.compute_lpj (gsc_et.py:811-944): log-joints of [null ; singletons ; multi-cause states] without annealing,
    prior odds included, in datapoint order -- against what the reference's own compute_lpj returned (the ``logpj``
    of the step goldens, scalar / diagonal / full noise, config-4 dimensions)."""
    from prosper_amd.em.camodels.gsc_et import GSC
    g = golden(case)
    m = GSC(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]), _kind(g))
    logpj, cand = m.compute_lpj(_An(T=float(g["T"])), _params(g), {"y": g["y"]})
    assert np.array_equal(np.asarray(cand), g["candidates"])
    lp = np.asarray(logpj)
    assert lp.shape == g["logpj"].shape and lp.dtype == np.float64
    np.testing.assert_allclose(lp, g["logpj"], rtol=1e-9, atol=1e-8)


def test_gsc_data_clusters_and_reference_order():
    """``my_data['data_clusters']`` after select_Hprimes is the reference's bucketing by candidate set (gsc_et.py:731-747),
    built on first access; with ``reference_order`` the E-step's rows follow the clusters as upstream (gsc_et.py:572-573)."""
    from prosper_amd.em.camodels.gsc_et import GSC
    g = golden("gsc_step_small.npz")
    y, cands = g["y"], g["candidates"]
    expect = {}
    for n in range(y.shape[0]):
        expect.setdefault(str(cands[n]), []).append(n)
    order = np.concatenate([np.array(v) for v in expect.values()])
    m = GSC(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    data = m.select_Hprimes(_params(g), {"y": y.copy()})
    cl = data["data_clusters"]
    assert list(cl.keys()) == list(expect.keys()) and len(cl) == len(expect)
    for k, c in cl.items():
        assert c["ind"] == expect[k] and np.array_equal(c["hprimes"], cands[expect[k][0]])
        assert np.array_equal(c["data"], y[expect[k]])
    suff = m.E_step(_An(T=float(g["T"])), _params(g), data)                      # datapoint order (the default)
    np.testing.assert_allclose(np.asarray(suff["xpt_s"]), g["xpt_s"], rtol=1e-8, atol=1e-12)
    m2 = GSC(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    m2.reference_order = True
    d2 = m2.select_Hprimes(_params(g), {"y": y.copy()})
    s2 = m2.E_step(_An(T=float(g["T"])), _params(g), d2)
    assert np.array_equal(np.asarray(d2["y"]), y[order])
    assert np.array_equal(np.asarray(d2["candidates"]).astype(np.int64), cands[order])
    np.testing.assert_allclose(np.asarray(s2["xpt_s"]), g["xpt_s"][order], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(np.asarray(s2["xpt_sz"]), g["xpt_sz"][order], rtol=1e-8, atol=1e-12)
    new = m2.M_step(_An(T=float(g["T"])), _params(g), s2, d2)                    # consistent: same update either way
    np.testing.assert_allclose(new["W"], g["W_new"], rtol=1e-7, atol=1e-9)


def test_gsc_component_scores_match_reference():
    """GSC.component_scores (gsc_et.py:752-809) against the reference's own output."""
    from prosper_amd.em.camodels.gsc_et import GSC
    g = golden("gsc_inference.npz")
    m = GSC(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    got = np.asarray(m.component_scores(_params(g), {"y": g["y"]}))
    np.testing.assert_allclose(got, g["component_scores"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("tag,kw", [("plain", dict(topK=5, adaptive=False)), ("adaptive", dict(topK=4, adaptive=True)),
                                    ("capped", dict(topK=3, adaptive=True, Hprime_max=5, gamma_max=3, logprob=True))])
def test_gsc_inference_matches_reference(tag, kw, capsys):
    """CAModel.inference (camodels/__init__.py:256-375) through GSC's own compute_lpj: top-K states bit for bit,
    marginals and probabilities; the adaptive run grows gamma to 5 and H' to 7 (g x g systems up to 5 x 5)."""
    from prosper_amd.em.camodels.gsc_et import GSC
    g = golden("gsc_inference.npz")
    m = GSC(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    res = m.inference(_An(T=1.0), _params(g), {"y": g["y"]}, **kw)
    assert (m.Hprime, m.gamma) == (int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(res["gamma"], g[tag + "_gamma"]) and np.array_equal(res["Hprime"], g[tag + "_Hprime"])
    assert res["s"].dtype == np.int8 and np.array_equal(res["s"], g[tag + "_s"])
    np.testing.assert_allclose(res["p"], g[tag + "_p"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(res["m"], g[tag + "_m"], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("D,H,Hp,gamma,N,T", [(256, 128, 6, 3, 1000, 1.0), (60, 50, 5, 4, 333, 1.3), (20, 10, 3, 2, 70, 1.0),
                                              (64, 200, 4, 2, 101, 1.0), (40, 40, 8, 2, 90, 1.1)])   # 16 latents per lane; H' = 8
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


@pytest.mark.parametrize("tag", ["scalar", "full"])
def test_gsc_compute_posterior_hprime_matches_reference(tag):
    """GSC.compute_posterior_hprime (gsc_et.py:260-398) on one data cluster against the reference's own output: non-symmetric
    psi_sq (after an M-step), T = 1.3, scalar and full noise; also with the candidates handed over in another order (the
    columns of the state matrix index positions in THAT order)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    from prosper_amd.em.camodels.gsc_et import GSC
    g = golden("gsc_posterior_hprime.npz")
    D, H, Hp, gamma, T = int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]), float(g["T"])
    sig = g[tag + "_sigma_sq"]
    params = {"W": g[tag + "_W"], "pi": g[tag + "_pi"], "mu": g[tag + "_mu"], "psi_sq": g[tag + "_psi_sq"],
              "sigma_sq": float(sig) if sig.ndim == 0 else sig}
    assert np.abs(params["psi_sq"] - params["psi_sq"].T).max() > 0
    m = GSC(D, H, Hp, gamma, tag)
    got = m.compute_posterior_hprime(_An(T=T), params, {"y": g[tag + "_y"], "candidates": g[tag + "_cand"]})
    for k in ("pstr_s", "pstr_ss", "pstr_sz", "pstr_szsz", "post_nfac_n"):
        ref = g[tag + "_" + k]
        assert got[k].shape == ref.shape
        np.testing.assert_allclose(got[k], ref, rtol=1e-8, atol=1e-12 * np.abs(ref).max(), err_msg=k)
    # the same set of candidates in reversed order: every state's latents change, the set of states does not
    rev = g[tag + "_cand"][::-1].copy()
    got2 = m.compute_posterior_hprime(_An(T=T), params, {"y": g[tag + "_y"], "candidates": rev})
    np.testing.assert_allclose(got2["post_nfac_n"], g[tag + "_post_nfac_n"], rtol=1e-8)
    np.testing.assert_allclose(got2["pstr_sz"], g[tag + "_pstr_sz"], rtol=1e-8, atol=1e-12 * np.abs(g[tag + "_pstr_sz"]).max())


def test_gsc_dead_latent_column_of_clamped_weights():
    """A latent whose prior odds underflow (pi_h = 1e-300) has nothing but `tiny`-clamped posterior weights, as upstream
    (gsc_et.py:354-356: exp(beta lp) < tiny -> tiny) -- its one-cause state and every multi-cause state that contains it,
    whether it is among a datapoint's candidates (selection ignores the prior) or not.  Its column of xpt_s / xpt_sz is a
    small multiple of tiny * nf for every datapoint: the kernel must reproduce exactly that -- these entries are all the
    column sums of a dead latent consist of -- and the sums over datapoints must carry them."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    from oracle import gsc_oracle as G
    from prosper_amd.em.camodels.gsc_et import GSC
    D, H, Hp, gamma, N = 40, 12, 4, 3, 300
    rng = np.random.RandomState(77)
    gt = {"W": rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.5), "psi_sq": np.eye(H), "sigma_sq": 1.0}
    y, _, _ = G.generate_gsc_data(gt, N, rng)
    dead = 7
    params = {"W": gt["W"] + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
              "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
    params["pi"][dead] = 1e-300                                # log odds -690.8: every state with this latent underflows
    model = G.make_model(D, H, Hp, gamma)
    cand_ref = G.select_hprimes(params, y, Hp)
    assert (cand_ref == dead).any() and not (cand_ref == dead).any(axis=1).all()       # a candidate for some datapoints only
    suff = G.e_step(G.Anneal(T=1.0), model, params, y, cand_ref)
    tiny = np.finfo(np.float64).tiny
    col = suff["xpt_s"][:, dead]
    # (a few tiny) * nf: ~1e-290 where the datapoint has states that do not underflow; where ALL of them do, every state
    # weighs `tiny` and the clamped weights are all there is (entries of order 1 / K)
    assert (col > 0).all() and np.median(col) < 1e-200
    m = GSC(D, H, Hp, gamma, "scalar")
    data = m.select_Hprimes({k: np.array(v, copy=True) for k, v in params.items()}, {"y": y})
    got = m.E_step(_An(T=1.0), {k: np.array(v, copy=True) for k, v in params.items()}, data)
    assert np.array_equal(np.asarray(data["candidates"]).astype(np.int64), cand_ref)
    xs, xsz = np.asarray(got["xpt_s"]), np.asarray(got["xpt_sz"])
    np.testing.assert_allclose(xs, suff["xpt_s"], rtol=1e-8, atol=0)                  # incl. the dead column, RELATIVE
    np.testing.assert_allclose(xsz, suff["xpt_sz"], rtol=1e-8, atol=0)
    np.testing.assert_allclose(got["xpt_ss"].sum(axis=0).cpu().numpy(), suff["xpt_ss"].sum(0), rtol=1e-9, atol=0)
    np.testing.assert_allclose(got["xpt_szsz"].sum(axis=0).cpu().numpy(), suff["xpt_szsz"].sum(0), rtol=1e-8, atol=1e-305)
    np.testing.assert_allclose(got["_sums"][0].cpu().numpy(), suff["xpt_s"].sum(0), rtol=1e-9, atol=0)
    # the update runs (the H x H inverse of a matrix with a ~1e-300 pivot is the reference's own LAPACK call on the host)
    new = m.step(_An(T=1.0), {k: np.array(v, copy=True) for k, v in params.items()}, {"y": y})
    assert np.isfinite(new["pi"]).all() and new["pi"][dead] == 5e-5          # clipped from below (gsc_et.py:640-645)


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


# ------------------------------------------------------------------------- config-4 properties
def test_config4_properties():
    """BASELINE config 4 dims (D=256 H=128 H'=6 gamma=3) at N = 30000 (every XCD holds datapoints, so the
    per-XCD accumulator copies and their fold are exercised): properties that need no oracle -- candidates are
    the H' best component scores, moments are additive over shards (what the all-reduce relies on), the
    sums the kernel reports equal the sums of what it wrote, posteriors are probabilities."""
    from prosper_amd.em.camodels.gsc_et import GSC
    dev = torch.device("cuda", 0)
    D, H, Hp, gamma, N = 256, 128, 6, 3, 30000
    gen = torch.Generator(device=dev).manual_seed(4)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
    S = (torch.rand(N, H, generator=gen, device=dev) < 2.0 / H).to(torch.float64)
    Z = S * (1.5 + torch.randn(N, H, generator=gen, device=dev, dtype=torch.float64))
    Y = Z @ W_gt.t() + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    rng = np.random.RandomState(4)
    p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
         "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
    an = _An(T=1.0)
    m = GSC(D, H, Hp, gamma, 'scalar')
    data = m.select_Hprimes(p, {"y": Y})
    ss = m.E_step(an, p, data)
    xs, xsz = ss["xpt_s"].tensor, ss["xpt_sz"].tensor
    sum_ss, sum_zz = ss["xpt_ss"].sum(axis=0), ss["xpt_szsz"].sum(axis=0)
    cand = data["candidates"].tensor.long()
    assert cand.shape == (N, Hp) and (cand[:, 1:] > cand[:, :-1]).all()          # sorted by index, distinct

    # (1) posteriors are probabilities; the diagonal of sum xpt_ss is the column sum of xpt_s (s_h^2 = s_h)
    assert (xs >= 0).all() and (xs <= 1 + 1e-12).all() and torch.isfinite(xsz).all()
    torch.testing.assert_close(torch.diagonal(sum_ss), xs.sum(0), rtol=1e-11, atol=1e-9)
    torch.testing.assert_close(ss["_sums"][0], xs.sum(0), rtol=1e-11, atol=1e-9)
    torch.testing.assert_close(ss["_sums"][1], xsz.sum(0), rtol=1e-11, atol=1e-9)
    # (2) second moments: symmetric, positive semi-definite, off-diagonal mass only between co-candidates
    torch.testing.assert_close(sum_ss, sum_ss.t(), rtol=0, atol=0)
    # (sum xpt_szsz is accumulated with both triangles as they are -- from the second EM step on psi_sq, and with it this
    # matrix, is not symmetric, gsc_et.py:660-675; with the symmetric psi_sq of this test it is, up to rounding)
    torch.testing.assert_close(sum_zz, sum_zz.t(), rtol=1e-10, atol=1e-10)
    assert torch.linalg.eigvalsh(0.5 * (sum_zz + sum_zz.t())).min().item() > -1e-8 * sum_zz.abs().max().item()
    co = torch.zeros(H, H, dtype=torch.bool, device=dev)
    co[cand[:, :, None].expand(N, Hp, Hp).reshape(-1), cand[:, None, :].expand(N, Hp, Hp).reshape(-1)] = True
    assert (sum_ss[~co] == 0).all() and (sum_zz[~co] == 0).all()
    # (3) outside the candidates a latent only has its singleton state: tiny but positive posterior
    assert (xs > 0).all()
    # (4) additivity over shards
    acc_ss, acc_zz = torch.zeros_like(sum_ss), torch.zeros_like(sum_zz)
    half = N // 2 + 5
    for sl in (slice(0, half), slice(half, N)):
        m2 = GSC(D, H, Hp, gamma, 'scalar')
        d2 = m2.select_Hprimes(p, {"y": Y[sl].contiguous()})
        s2 = m2.E_step(an, p, d2)
        # per datapoint up to the rounding of its scores (another GEMM tiling for another shard length; the weights
        # are exp(log-joint) with |log-joint| ~ 1e3, so one ulp in a score is ~1e-12 relative in a weight)
        torch.testing.assert_close(s2["xpt_s"].tensor, xs[sl], rtol=1e-9, atol=1e-300)
        acc_ss += s2["xpt_ss"].sum(axis=0)
        acc_zz += s2["xpt_szsz"].sum(axis=0)
    torch.testing.assert_close(acc_ss, sum_ss, rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(acc_zz, sum_zz, rtol=1e-10, atol=1e-10)
    # (5) the update keeps the parameters sane and near the generating ones
    new = m.M_step(an, p, ss, data)
    assert np.isfinite(new["W"]).all() and (new["pi"] > 0).all() and (new["pi"] < 1).all() and new["sigma_sq"] > 0
    assert np.abs(new["W"] - W_gt.cpu().numpy()).mean() < 0.2 and abs(new["pi"].mean() * H - 2.0) < 0.5


@pytest.mark.parametrize("H,D,learn", [(24, 48, 15), (128, 256, 15), (10, 7, 5), (10, 7, 0)])
def test_gsc_mstep_finish_kernel(H, D, learn):
    """pm_gsc_mstep_finish_f64 -- pi clip, mu, psi_sq, scalar sigma_sq and the next E-step's tables on the device --
    against the host formulas of GSC._update / GSC._tables_for (gsc_et.py:640-713, 260-398)."""
    import ctypes
    from prosper_amd import _lib
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(H + learn)
    N = 5000.0
    sum_s = rs.uniform(0.0, 400.0, H)
    sum_s[0] = 0.01                                   # clipped from below
    sum_s[1] = N                                      # ... and from above
    sum_sz = sum_s * rs.normal(1.4, 0.2, H)
    sq = lambda: (lambda B: B @ B.T)(rs.normal(size=(H, H + 3)))
    sum_ss, sum_zz, xs_xsz, xsz_xsz = sq() + np.diag(sum_s), sq(), rs.normal(size=(H, H)), sq()
    ss_inv = np.linalg.inv(sum_ss + 1e-5 * np.eye(H))
    W = rs.normal(size=(D, H))
    G = W.T @ W
    sum_yy = float(np.trace(xsz_xsz @ G) * 1.3 + 100.0)
    old = {"pi": rs.uniform(0.01, 0.2, H), "mu": rs.normal(size=H), "psi_sq": sq() / H + np.eye(H), "sigma_sq": 1.7}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))).to(dev)
    oldv = t(np.concatenate([old["pi"], old["mu"], old["psi_sq"].reshape(-1), [old["sigma_sq"]]]))
    params = torch.zeros(2 * H + H * H + 1, dtype=torch.float64, device=dev)
    tables = torch.zeros(9 * H, dtype=torch.float64, device=dev)
    args = [t(x) for x in (xs_xsz, xsz_xsz, sum_ss, sum_zz, ss_inv, sum_s, sum_sz, [sum_yy], G)]
    _lib.call("pm_gsc_mstep_finish_f64", *[ctypes.c_void_p(a.data_ptr()) for a in args], ctypes.c_void_p(oldv.data_ptr()),
              ctypes.c_double(N), D, H, learn, ctypes.c_void_p(params.data_ptr()), ctypes.c_void_p(tables.data_ptr()),
              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    got, tab = params.cpu().numpy(), tables.cpu().numpy().reshape(9, H)
    pi = np.clip(sum_s / N, 5e-5, 1 - 5e-5) if learn & 1 else old["pi"]
    mu = sum_sz / (sum_s + np.finfo(np.float64).eps) if learn & 2 else old["mu"]
    psi = ((np.outer(mu, mu) * sum_ss + sum_zz - 2 * (mu[:, None] * xs_xsz)) * ss_inv + 1e-5 * np.eye(H)) if learn & 4 \
        else old["psi_sq"]
    s2 = (sum_yy - np.einsum('ij,ji->', xsz_xsz, G)) / N / D + 1e-5 if learn & 8 else old["sigma_sq"]
    np.testing.assert_allclose(got[:H], pi, rtol=1e-14)
    np.testing.assert_allclose(got[H:2 * H], mu, rtol=1e-14)
    np.testing.assert_allclose(got[2 * H:2 * H + H * H].reshape(H, H), psi, rtol=1e-12, atol=1e-12 * np.abs(psi).max())
    np.testing.assert_allclose(got[-1], s2, rtol=1e-11)
    psid, Gd = np.diag(psi), np.diag(G)
    lam = Gd / s2 + 1. / psid
    want = np.stack([-(np.log(psid) + np.log(lam)) - mu * mu * Gd / s2, 2. * mu / s2, Gd * mu, 1. / (lam * s2 * s2),
                     1. / (lam * s2), 1. / lam, mu, np.log(pi) - np.log(1 - pi), np.full(H, 1. / s2)])
    # the list threshold of the next E-step (pm_gsc_estep_lists_f64): 2^-57 / N of the smallest column sum (round 6: the
    # all-ranks N the kernel is given, not a fixed 2^-75 that assumed N <= 2^18)
    want[8, 1] = float(np.abs(sum_sz).min()) * 2.0 ** -57 / max(float(N), 1.0)
    thr_got, tab[8, 1] = tab[8, 1], want[8, 1]
    np.testing.assert_allclose(thr_got, want[8, 1], rtol=1e-14)
    # ... and the threshold below which an entry of a datapoint's pair blocks is not sent (gsc_estep_kernel, thr_p)
    want[8, 2] = min(float(np.abs(sum_s).min()), float(np.abs(np.diag(sum_zz)).min())) * 2.0 ** -57 / max(float(N), 1.0)
    thr_got, tab[8, 2] = tab[8, 2], want[8, 2]
    np.testing.assert_allclose(thr_got, want[8, 2], rtol=1e-14)
    np.testing.assert_allclose(tab, want, rtol=1e-10, atol=1e-12)


def test_gsc_em_loop_speculation_is_transparent():
    """The M-step leaves the next step's W^T, Gram matrix and scores on the device (GSC._speculate), finishes its
    H x H algebra there (pm_gsc_mstep_finish_f64) and, inside an EM loop, launches the next E-step itself.  That must
    not change a trajectory -- whether the caller feeds the returned parameters straight back or edits them first."""
    from prosper_amd.em.camodels.gsc_et import GSC
    D, H, Hp, gamma, N = 48, 24, 4, 3, 700
    rng = np.random.RandomState(11)
    W_gt = rng.normal(size=(D, H))
    S = rng.random_sample((N, H)) < 2.0 / H
    y = (S * (1.5 + rng.normal(size=(N, H)))) @ W_gt.T + rng.normal(size=(N, D))
    p0 = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
          "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
    an = _An(T=1.0)
    runs = []
    for spec in (True, False):
        m = GSC(D, H, Hp, gamma, 'scalar')
        m.speculate = m.speculate_estep = spec        # (whatever PM_SPECULATE* say in the environment)
        p = {k: np.array(v, copy=True) for k, v in p0.items()}
        used = []
        for it in range(8):
            if it == 3:                                   # the caller edits W: the speculated scores must be dropped
                p["W"] = p["W"] * (1.0 + 1e-3 * np.cos(np.arange(D * H).reshape(D, H)))
            if it == 4:                                   # an in-place edit of the very array the M-step returned
                p["W"][0, 0] += 0.01
            p = m.step(an, p, {"y": y})
            used.append(m._par.get("scores", "absent"))
        runs.append(p)
        if spec:
            assert m._seed is not None and m._seed["W_host"] is not p["W"] and np.array_equal(m._seed["W_host"], p["W"])
            # the M-step also finishes on the device and launches the next E-step itself: adopted at steps 2, 5, 6, 7
            # (step 1 follows the first step: no flat schedule yet; 3 and 4 follow an edit)
            # (... unless the device rejected the warm start of the general matrix sum xpt_szsz at one of them -- early EM
            # steps move it a lot --: that step's inverse then comes from the host, as upstream, and nothing is launched
            # from the device's result)
            assert 3 <= m.spec_hits <= 4, m.spec_hits
        else:
            assert m.spec_hits == 0
    for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
        np.testing.assert_allclose(runs[0][k], runs[1][k], rtol=1e-8, atol=1e-11, err_msg=k)


def _ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("M,Nc,K,count", [(512, 128, 3000, 2999), (512, 128, 3000, 0), (256, 256, 700, 13), (128, 128, 64, 64)])
def test_gemm_tn_acc_rows_matches_numpy(M, Nc, K, count):
    """pm_gemm_tn_acc_rows_f64: C += A[rows]^T B[rows] over a device-side row list with a device-side length (ragged ends
    read the zero row), A and B views of one buffer as in GSC's M-step."""
    import ctypes
    from prosper_amd import _lib
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(M + K + count)
    big = rng.normal(size=(K + 1, M + 64))
    big[K] = 0.0
    rows = rng.permutation(K)[:max(count, 1)].astype(np.int32)
    C0 = rng.normal(size=(M, Nc))
    b = torch.from_numpy(big).to(dev)
    c = torch.from_numpy(C0.copy()).to(dev)
    r = torch.from_numpy(np.concatenate([rows, np.full(K - len(rows), -7, np.int32)])).to(dev)   # (garbage behind the list)
    cnt = torch.tensor([count], dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    Bv = b[:, M - Nc:M] if M > Nc else b[:, :Nc]                # B: the last Nc columns of A's block (GSC: xsz inside [Y|xs|xsz])
    _lib.call("pm_gemm_tn_acc_rows_f64", _ptr(b), b.stride(0), _ptr(Bv), b.stride(0), _ptr(c), Nc, M, Nc, _ptr(r), _ptr(cnt),
              K, K, st)
    sel = rows[:count]
    A_np = big[sel][:, :M]
    B_np = big[sel][:, M - Nc:M] if M > Nc else big[sel][:, :Nc]
    want = C0 + A_np.T @ B_np
    np.testing.assert_allclose(c.cpu().numpy(), want, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(want).max()))


def test_wp_sparse_transposed_matches_numpy():
    """pm_wp_sparse_t_f64: C (D x H) += Y^T V with V's rows as lists; rows with an empty list contribute nothing."""
    import ctypes
    from prosper_amd import _lib
    dev = torch.device("cuda", 0)
    N, D, H = 3001, 512, 128
    rng = np.random.RandomState(5)
    Y = rng.normal(size=(N + 1, D))
    idx = np.full((N, 16), 0xFFFF, dtype=np.uint16)
    val = rng.normal(size=(N, 16))
    V = np.zeros((N, H))
    for n in range(N):
        k = 0 if n % 5 == 0 else rng.randint(1, 17)
        hs = rng.permutation(H)[:k]
        idx[n, :k] = hs
        V[n, hs] = val[n, :k]
    C0 = rng.normal(size=(D, H))
    c = torch.from_numpy(C0.copy()).to(dev)
    y, i_d, v_d = torch.from_numpy(Y).to(dev), torch.from_numpy(idx.view(np.int16)).to(dev), torch.from_numpy(val).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.call("pm_wp_sparse_t_f64", _ptr(i_d), _ptr(v_d), _ptr(y), D, _ptr(c), H, N, H, D, st)
    want = C0 + Y[:N].T @ V
    np.testing.assert_allclose(c.cpu().numpy(), want, rtol=1e-12, atol=1e-12 * np.abs(want).max())


@pytest.mark.parametrize("dead", [False, True])
def test_gsc_sparse_moments_equal_the_dense_contraction(dead):
    """Inside an EM loop the M-step launches the next E-step with lists (pm_gsc_estep_lists_f64) and contracts
    [Y | xs | xsz]^T xsz as sparse product + gathered dense GEMM.  Same trajectory as the one dense GEMM, to rounding;
    the split really happens (some rows listed, some dense); a latent whose every weight is the `tiny` clamp pulls the
    threshold to nothing -- every row dense -- and the result still agrees."""
    from prosper_amd.em.camodels.gsc_et import GSC
    from prosper_amd.em.camodels._device import KernelTimer
    D, H, Hp, gamma, N = 256, 128, 6, 3, 6000
    rng = np.random.RandomState(41)
    W_gt = rng.normal(size=(D, H))
    S = rng.random_sample((N, H)) < 2.0 / H
    y = (S * (1.5 + rng.normal(size=(N, H)))) @ W_gt.T + rng.normal(size=(N, D))
    p0 = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
          "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
    learn = ('W', 'pi', 'mu', 'psi_sq', 'sigma_sq')
    if dead:
        p0["pi"][7] = 1e-300
        learn = ('W', 'mu', 'psi_sq', 'sigma_sq')        # pi stays: latent 7 stays dead through the loop
    an = _An(T=1.0)
    runs, counts = [], None
    for sparse, pairs in ((True, False), (False, False), (True, True)):
        m = GSC(D, H, Hp, gamma, 'scalar', to_learn=list(learn))
        m.sparse_moments = sparse
        m.list_pairs = pairs          # (True: the H x H blocks of listed datapoints from their lists, pm_gsc_list_pairs_f64)
        p = {k: np.array(v, copy=True) for k, v in p0.items()}
        traj = []
        for it in range(7):
            if sparse and it == 6:
                m.timer = KernelTimer()
            p = m.step(an, p, {"y": y})
            traj.append({k: np.array(p[k], copy=True) for k in p0})
        runs.append(traj)
        if sparse:
            names = set(m.timer.summary())
            m.timer = None
            assert m.spec_hits >= 3 and (("stats_pairs" in names) == pairs)
            res = m._resident(y)
            lists = [l for l in res.get("gsc_lists", []) if l is not None]
            assert lists, "the list form of the E-step never ran"
            assert "stats_sparse" in names, names
            empty = (lists[0][0][:, 0].cpu().numpy().view(np.uint16) == 0xFFFF)
            counts = (int(empty.sum()), N)
    for other in (runs[0], runs[2]):
        for a, b in zip(other, runs[1]):
            for k in p0:
                np.testing.assert_allclose(a[k], b[k], rtol=1e-9, atol=1e-11 * max(1.0, np.abs(b[k]).max()), err_msg=k)
    # (the "dead" latent still collects 1 / (H + S) of every datapoint whose weights are ALL the clamp, so its column sum --
    # and with it the threshold -- stays ordinary: its 1e-243 entries elsewhere are dropped, as they should be)
    n_dense, n_all = counts
    assert 0.05 * n_all < n_dense < 0.5 * n_all, counts


def test_gsc_lists_with_zero_threshold_keep_every_row_dense():
    """Threshold 0 (what a vanishing column sum of xpt_sz gives): every weight is at least the `tiny` clamp, so every row has
    H entries above it -- all rows land in the dense list, all lists are empty, and the gathered GEMM alone reproduces the
    dense contraction."""
    import ctypes
    from prosper_amd import _lib
    from prosper_amd.em.camodels.gsc_et import GSC
    D, H, Hp, gamma, N = 256, 128, 6, 3, 1500
    rng = np.random.RandomState(43)
    W_gt = rng.normal(size=(D, H))
    S = rng.random_sample((N, H)) < 2.0 / H
    y = (S * (1.5 + rng.normal(size=(N, H)))) @ W_gt.T + rng.normal(size=(N, D))
    p = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
         "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
    m = GSC(D, H, Hp, gamma, 'scalar')
    res = m._resident(y)
    par = m._tables_for(p, res)
    A = m._gemm_nt(res["Y"], par["Wst"], m._buf("scores", (N, H)), "scores_gemm")
    for thr, all_dense in ((0.0, True), (1e-20, False)):
        tdev = torch.zeros(9 * H, dtype=torch.float64, device=m.device)
        tdev[:8 * H] = par["tables"].reshape(-1)[:8 * H]
        tdev[8 * H] = 1.0 / par["s2"]
        tdev[8 * H + 1] = thr
        cand, xs, xsz, stats = m._launch_estep(res, A, par["G"], par["psi_d"], par["yn"], tdev, 0.0, 1.0, None, lists=True)
        nz_idx, nz_val, rows, cnt, big = stats._pm_lists
        n_dense = int(cnt.item())
        listed = nz_idx[:, 0].cpu().numpy().view(np.uint16) != 0xFFFF
        assert n_dense + int(listed.sum()) == N and sorted(rows[:n_dense].cpu().numpy().tolist()) == np.nonzero(~listed)[0].tolist()
        assert (n_dense == N) if all_dense else (0 < n_dense < N // 2)
        want = (big[:N].t() @ xsz).cpu().numpy()
        got = torch.zeros((D + 2 * H, H), dtype=torch.float64, device=m.device)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.call("pm_wp_sparse_t_f64", _ptr(nz_idx), _ptr(nz_val), _ptr(big), big.stride(0), _ptr(got), H, N, H, D + 2 * H, st)
        if all_dense:
            assert float(got.abs().max()) == 0.0
        _lib.call("pm_gemm_tn_acc_rows_f64", _ptr(big), big.stride(0), _ptr(xsz), big.stride(0), _ptr(got), H, D + 2 * H, H,
                  _ptr(rows), _ptr(cnt), N, N, st)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-11, atol=1e-13 * np.abs(want).max())


def test_gsc_pair_block_threshold_drops_nothing_visible():
    """With tables from the device (sigma_sq = 0) the E-step kernel sends an entry of a datapoint's xpt_ss / xpt_szsz blocks only
    if it exceeds tables[8 H + 2] = 2^-75 of the smallest diagonal entry of the previous step's sums.  Against the same pass
    with threshold 0: identical moments and column sums, block sums equal to within N * threshold -- and a threshold large enough
    to matter really drops entries (the switch is live)."""
    from prosper_amd.em.camodels.gsc_et import GSC
    D, H, Hp, gamma, N = 256, 128, 6, 3, 4000
    rng = np.random.RandomState(47)
    W_gt = rng.normal(size=(D, H))
    S = rng.random_sample((N, H)) < 2.0 / H
    y = (S * (1.5 + rng.normal(size=(N, H)))) @ W_gt.T + rng.normal(size=(N, D))
    p = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
         "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
    m = GSC(D, H, Hp, gamma, 'scalar')
    res = m._resident(y)
    par = m._tables_for(p, res)
    A = m._gemm_nt(res["Y"], par["Wst"], m._buf("scores", (N, H)), "scores_gemm")
    outs = {}
    for thr in (0.0, None, 1e-3):
        tdev = torch.zeros(9 * H, dtype=torch.float64, device=m.device)
        tdev[:8 * H] = par["tables"].reshape(-1)[:8 * H]
        tdev[8 * H] = 1.0 / par["s2"]
        if thr is None:       # what pm_gsc_mstep_finish_f64 would leave: from this very pass's sums
            st0 = outs[0.0][3]
            diag = torch.minimum(st0[2 * H * H:2 * H * H + H].abs(),
                                 (st0[H * H:2 * H * H].view(H, H).diagonal() + st0[2 * H * H + 2 * H:2 * H * H + 3 * H]).abs())
            thr = float(diag.min()) * 2.0 ** -75
            assert thr > 0.0
        tdev[8 * H + 2] = thr
        cand, xs, xsz, stats = m._launch_estep(res, A, par["G"], par["psi_d"], par["yn"], tdev, 0.0, 1.0, None)
        outs[thr] = (cand.clone(), xs.clone(), xsz.clone(), stats[:2 * H * H + 3 * H].clone())
    (k0, kt, kbig) = sorted(outs)
    a, b, c = outs[k0], outs[kt], outs[kbig]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    # column sums, singleton diagonal: untouched (atomics in another order of arrival: rounding only)
    np.testing.assert_allclose(b[3][2 * H * H:].cpu().numpy(), a[3][2 * H * H:].cpu().numpy(), rtol=1e-13)
    # (same entries in another order of arrival: the atomics' rounding, 1e-16 of an entry's own scale, is there in both)
    diff = (a[3][:2 * H * H] - b[3][:2 * H * H]).abs().cpu().numpy()
    scale = a[3][:2 * H * H].abs().cpu().numpy()
    assert (diff <= N * kt + 1e-13 * scale).all(), float((diff - 1e-13 * scale).max())
    dropped = (a[3][:2 * H * H] - c[3][:2 * H * H]).abs().max()
    assert float(dropped) > 1e-4, float(dropped)


def test_config4_full_shard_against_oracle():
    """BASELINE config 4 at its real size -- D=256 H=128 H'=6 gamma=3, N = 200 000 -- through the shipped launches.
    The oracle runs on ~400 sampled rows -- and, round 6, on every row -- : candidates identical, posterior moments xpt_s / xpt_sz to 1e-9; the sampled rows
    as a shard of their own reproduce the oracle's whole EM step (device M-step tail included)."""
    from oracle import gsc_oracle as G
    from prosper_amd.em.camodels.gsc_et import GSC
    dev = torch.device("cuda", 0)
    D, H, Hp, gamma, N = 256, 128, 6, 3, 200_000
    gen = torch.Generator(device=dev).manual_seed(44)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
    Y = torch.empty(N, D, dtype=torch.float64, device=dev)
    for lo in range(0, N, 50_000):
        S = (torch.rand(50_000, H, generator=gen, device=dev) < 2.0 / H).to(torch.float64)
        Z = S * (1.5 + torch.randn(50_000, H, generator=gen, device=dev, dtype=torch.float64))
        Y[lo:lo + 50_000] = Z @ W_gt.t() + torch.randn(50_000, D, generator=gen, device=dev, dtype=torch.float64)
    rng = np.random.RandomState(44)
    p = {"W": W_gt.cpu().numpy() + 0.1 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
         "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
    an = _An(T=1.0)
    m = GSC(D, H, Hp, gamma, 'scalar')
    cp = lambda q: {k: np.array(v, copy=True) for k, v in q.items()}
    data = m.select_Hprimes(cp(p), {"y": Y})
    ss = m.E_step(an, cp(p), data)
    rows = np.unique(np.concatenate([rng.randint(0, N, size=300), np.arange(0, 48), np.arange(N - 48, N)]))
    idx = torch.from_numpy(rows).to(dev)
    y_s = Y[idx].cpu().numpy()
    model = G.make_model(D, H, Hp, gamma)
    cand_ref = G.select_hprimes(p, y_s, Hp)
    assert np.array_equal(data["candidates"].tensor[idx].cpu().numpy().astype(np.int64), cand_ref)
    suff = G.e_step(G.Anneal(T=1.0), model, p, y_s, cand_ref)
    np.testing.assert_allclose(ss["xpt_s"].tensor[idx].cpu().numpy(), suff["xpt_s"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ss["xpt_sz"].tensor[idx].cpu().numpy(), suff["xpt_sz"], rtol=1e-9, atol=1e-12)
    # round 6: the candidates of EVERY datapoint, and the moments of every datapoint too (the oracle's E-step walks the data
    # clusters in Python: ~0.25 ms per row on the GPU boxes' hosts)
    bad = 0
    for lo in range(0, N, 20_000):
        c_ref = G.select_hprimes(p, Y[lo:lo + 20_000].cpu().numpy(), Hp)
        bad += int((data["candidates"].tensor[lo:lo + 20_000].cpu().numpy().astype(np.int64) != c_ref).any(axis=1).sum())
    assert bad == 0, "%d of %d datapoints with other candidates than the oracle's" % (bad, N)
    worst = 0.0
    full = {"xpt_s": np.empty((N, H)), "xpt_sz": np.empty((N, H)), "xpt_ss": np.zeros((1, H, H)), "xpt_szsz": np.zeros((1, H, H))}
    y_host = np.empty((N, D))
    # (the first 100 000 rows in the suite; PM_FULL_PARITY=1: all 200 000, and the whole shard's M-step against the oracle)
    N_all = N if os.environ.get("PM_FULL_PARITY") == "1" else 100_000
    y_host[N_all:] = Y[N_all:].cpu().numpy()
    for lo in range(0, N_all, 2000):      # (2000 rows at a time: the oracle materialises the (n, H, H) second moments)
        y_m = y_host[lo:lo + 2000] = Y[lo:lo + 2000].cpu().numpy()
        c_m = data["candidates"].tensor[lo:lo + 2000].cpu().numpy().astype(np.int64)
        suff_m = G.e_step(G.Anneal(T=1.0), model, p, y_m, c_m)
        for k in ("xpt_s", "xpt_sz"):
            got_m = ss[k].tensor[lo:lo + 2000].cpu().numpy()
            worst = max(worst, float(np.max(np.abs(got_m - suff_m[k]) / (1e-12 + 1e-9 * np.abs(suff_m[k])))))
            full[k][lo:lo + 2000] = suff_m[k]
        full["xpt_ss"][0] += suff_m["xpt_ss"].sum(axis=0)
        full["xpt_szsz"][0] += suff_m["xpt_szsz"].sum(axis=0)
    assert worst <= 1.0, "posterior moments: %.2f times the tolerance (rtol 1e-9, atol 1e-12)" % worst
    new = m.M_step(an, cp(p), ss, data)
    assert np.isfinite(new["W"]).all() and (new["pi"] > 0).all() and new["sigma_sq"] > 0
    # ... and the M-step of the WHOLE shard -- list pass, sparse product + gathered GEMM, pair atomics under their threshold,
    # device-side inverses and finish kernel -- against the oracle's update from its own moments of all 200 000 rows (the
    # (N, H, H) moments enter the update only as sums: accumulated chunk by chunk)
    ref_full = None
    if N_all == N:
        ref_full = G.m_step(model, cp(p), full, y_host)
        tol_f = max(1e-8, 50 * np.linalg.cond(full["xpt_szsz"][0]) * np.finfo(float).eps)
        for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
            np.testing.assert_allclose(new[k], ref_full[k], rtol=10 * tol_f, atol=tol_f * max(1.0, np.abs(ref_full[k]).max()),
                                       err_msg="full shard: " + k)
    # ... and the SECOND step of an EM loop on the whole shard -- the pass the first M-step launched itself: lists, the sparse
    # product + the gathered GEMM over the dense rows -- against the oracle's second step from the oracle's first
    # (on the first 50 000 rows in the suite; PM_FULL_PARITY=1: all 200 000 -- run and passing)
    N2 = N if os.environ.get("PM_FULL_PARITY") == "1" else 50_000
    Y2, y2 = Y[:N2], y_host[:N2]
    first = {k: (v[:N2] if k in ("xpt_s", "xpt_sz") else np.zeros_like(v)) for k, v in full.items()}
    for lo in range(0, N2, 2000):
        c_m = data["candidates"].tensor[lo:lo + 2000].cpu().numpy().astype(np.int64)
        suff_m = G.e_step(G.Anneal(T=1.0), model, p, y2[lo:lo + 2000], c_m)
        first["xpt_ss"][0] += suff_m["xpt_ss"].sum(axis=0)
        first["xpt_szsz"][0] += suff_m["xpt_szsz"].sum(axis=0)
    ref_first = ref_full if N2 == N else G.m_step(model, cp(p), first, y2)
    m2 = GSC(D, H, Hp, gamma, 'scalar')
    m2._predict_anneal(an)                 # (as if a step at this annealing point had gone before: the look-ahead is trusted)
    p1 = m2.step(an, cp(p), {"y": Y2})
    p2 = m2.step(an, p1, {"y": Y2})
    assert m2.spec_hits == 1
    full2 = {"xpt_s": np.empty((N2, H)), "xpt_sz": np.empty((N2, H)), "xpt_ss": np.zeros((1, H, H)), "xpt_szsz": np.zeros((1, H, H))}
    for lo in range(0, N2, 2000):
        y_m = y2[lo:lo + 2000]
        suff_m = G.e_step(G.Anneal(T=1.0), model, ref_first, y_m, G.select_hprimes(ref_first, y_m, Hp))
        for k in ("xpt_s", "xpt_sz"):
            full2[k][lo:lo + 2000] = suff_m[k]
        full2["xpt_ss"][0] += suff_m["xpt_ss"].sum(axis=0)
        full2["xpt_szsz"][0] += suff_m["xpt_szsz"].sum(axis=0)
    ref2 = G.m_step(model, cp(ref_first), full2, y2)
    tol_2 = 10 * max(1e-8, 50 * np.linalg.cond(full2["xpt_szsz"][0]) * np.finfo(float).eps)       # (two steps of error growth)
    for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
        np.testing.assert_allclose(p2[k], ref2[k], rtol=10 * tol_2, atol=tol_2 * max(1.0, np.abs(ref2[k]).max()),
                                   err_msg="full shard, second step: " + k)
    ref, log = G.em_step(G.Anneal(T=1.0), model, cp(p), y_s)
    got = GSC(D, H, Hp, gamma, 'scalar').step(an, cp(p), {"y": y_s})
    tol = max(1e-8, 50 * np.linalg.cond(log["suff"]["xpt_szsz"].sum(0)) * np.finfo(float).eps)
    for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
        np.testing.assert_allclose(got[k], ref[k], rtol=10 * tol, atol=tol * max(1.0, np.abs(ref[k]).max()), err_msg=k)
