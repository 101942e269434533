"""Pin the oracle (oracle/bsc_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, minted by tests/golden/make_golden.py from /root/reference)."""
import numpy as np
import pytest

from conftest import golden, bsc_step_cases, rank_deficient
from oracle import bsc_oracle as O

RTOL = 1e-10   # CPU restatement vs reference


def _model(g):
    m = O.make_model(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(m["SM"], g["state_matrix"])
    assert np.array_equal(m["state_abs"], g["state_abs"])
    return m


def _anneal(g):
    return O.Anneal(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))


@pytest.mark.parametrize("case", bsc_step_cases())
@pytest.mark.parametrize("flavour", ["loop", "vec"])
def test_bsc_step_matches_reference(case, flavour):
    g = golden(case)
    m, an = _model(g), _anneal(g)
    mu = g["mu"]
    select = O.select_hprimes_loop if flavour == "loop" else O.select_hprimes_vec
    estep = O.e_step_loop if flavour == "loop" else O.e_step_vec
    stats = O.m_step_stats_loop if flavour == "loop" else O.m_step_stats_vec
    cand = select(g["W"], g["y"], m["Hprime"])
    assert np.array_equal(cand, g["candidates"])
    logpj = estep(an, g["W"], float(g["pi"]), float(g["sigma"]), mu, g["y"], cand, m["SM"], m["state_abs"])
    np.testing.assert_allclose(logpj, g["logpj"], rtol=1e-11, atol=1e-10)
    new, log = O.m_step(an, m, g["W"], float(g["pi"]), float(g["sigma"]), mu, g["y"], cand, g["logpj"],
                        to_learn=tuple(str(s) for s in g["to_learn"]), stats_fn=stats)
    assert log["N_use"] == int(g["N_use"]) == int(g["N"])
    np.testing.assert_allclose(log["L"], g["L"], rtol=1e-12)
    if "Wq" in g:      # config-2 fixtures: the statistics the reference handed to lstsq (bsc_et.py:373-380)
        np.testing.assert_allclose(log["stats"]["Wq"], g["Wq"], rtol=1e-10, atol=1e-12 * np.abs(g["Wq"]).max())
        np.testing.assert_allclose(log["stats"]["Wp"], g["Wp"], rtol=1e-10, atol=1e-12 * np.abs(g["Wp"]).max())
    if not rank_deficient(g) or flavour == "loop":     # loop: the reference's own summation order, bit for bit
        np.testing.assert_allclose(new["W"], g["W_new"], rtol=RTOL, atol=1e-9 * np.abs(g["W_new"]).max())
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=RTOL)
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=RTOL)
    np.testing.assert_allclose(new["mu"], g["mu_new"], rtol=1e-9, atol=1e-10)


def test_state_matrix_counts():
    # config 2 has S = 28+56+70 = 154, config 1 has S = 20
    assert O.generate_state_matrix(8, 4)[1] == 154
    assert O.generate_state_matrix(5, 3)[1] == 20
    assert O.generate_state_matrix(4, 4)[1] == 2 ** 4 - 4 - 1


def test_sharded_statistics_equal_single_shard():
    g = golden("bsc_step_c1_anneal_cut.npz")
    m, an = _model(g), _anneal(g)
    N = g["y"].shape[0]
    shards = [np.arange(0, N // 3), np.arange(N // 3, N)]
    args = (an, m, g["W"], float(g["pi"]), float(g["sigma"]), g["mu"], g["y"], g["candidates"], g["logpj"])
    a, _ = O.m_step(*args, stats_fn=O.m_step_stats_vec)
    b, _ = O.m_step(*args, stats_fn=O.m_step_stats_vec, shards=shards)
    np.testing.assert_allclose(a["W"], b["W"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(a["sigma"], b["sigma"], rtol=1e-13)


def test_trajectory_c1_matches_reference():
    """BASELINE config 1: 20 EM steps on bars data, parameters after every step."""
    g = golden("bsc_traj_c1.npz")
    m = O.make_model(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    steps = int(g["steps"])
    params = {"W": g["W0"].copy(), "pi": float(g["pi0"]), "sigma": float(g["sigma0"])}
    for t in range(steps):
        # LinearAnnealing(20): T [(0,2),(.7,1)], Ncut_factor [(0,0),(2/3,1)]
        T = 2.0 + (1.0 - 2.0) * min(t, 14) / 14.0
        nc = min(t, 13) / 13.0
        an = O.Anneal(T=T, Ncut_factor=nc, anneal_prior=False)
        params, log = O.em_step(an, m, params, g["y"], stats_fn=O.m_step_stats_vec, vec=True)
        np.testing.assert_allclose(log["L"], g["L"][t], rtol=1e-9, err_msg="step %d" % t)
        assert log["N_use"] == int(g["N_use"][t])
        np.testing.assert_allclose(params["W"], g["W"][t], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(params["pi"], g["pi"][t], rtol=1e-8)
        np.testing.assert_allclose(params["sigma"], g["sigma"][t], rtol=1e-8)


def test_init_and_generate_match_reference_rng_stream():
    g = golden("bsc_init_c1.npz")
    rng = np.random.RandomState(int(g["seed_data"]))
    y, s = O.generate_bsc_data(g["W_gt"], float(g["pi_gt"]), float(g["sigma_gt"]), int(g["N"]), rng)
    assert np.array_equal(s, g["s"])
    np.testing.assert_allclose(y, g["y"], rtol=1e-13, atol=1e-13)
    init = O.standard_init(g["y"], int(g["H"]), np.random.RandomState(int(g["seed_init"])))
    np.testing.assert_allclose(init["W"], g["W0"], rtol=1e-13)
    np.testing.assert_allclose(init["sigma"], g["sigma0"], rtol=1e-13)
    assert init["pi"] == float(g["pi0"])


# ----------------------------------------------------------------------------- MCA (mca_et.py)
def _mca_cases():
    import glob, os
    from conftest import GOLDEN
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "mca_step_*.npz")))


@pytest.mark.parametrize("case", _mca_cases())
@pytest.mark.parametrize("flavour", ["loop", "vec"])
def test_mca_step_matches_reference(case, flavour):
    from oracle import mca_oracle as M
    g = golden(case)
    model = M.make_model(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(model["SM"], g["state_matrix"])
    an = M.Anneal(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]))
    params = M.check_params({"W": g["W"], "pi": float(g["pi"]), "sigma": float(g["sigma"])})
    vec = flavour == "vec"
    cand = (M.select_hprimes_vec if vec else M.select_hprimes_loop)(params["W"], g["y"], model["Hprime"])
    assert np.array_equal(cand, g["candidates"])
    logpj = (M.e_step_vec if vec else M.e_step_loop)(an, params["W"], params["pi"], params["sigma"], g["y"], cand,
                                                     model["SM"], model["state_abs"])
    np.testing.assert_allclose(logpj, g["logpj"], rtol=1e-10, atol=1e-9)
    new, log = M.m_step(an, model, params["W"], params["pi"], params["sigma"], g["y"], cand, g["logpj"], vec=vec)
    assert log["N_use"] == int(g["N_use"])
    np.testing.assert_allclose(new["W"], g["W_new"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=1e-10)
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=1e-10)
    np.testing.assert_allclose(new["Q"], g["Q"], rtol=1e-11)


# ----------------------------------------------------------------------------- MMCA (mmca_et.py)
def _mmca_cases():
    import glob, os
    from conftest import GOLDEN
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "mmca_step_*.npz")))


@pytest.mark.parametrize("case", _mmca_cases())
@pytest.mark.parametrize("flavour", ["loop", "vec"])
def test_mmca_step_matches_reference(case, flavour):
    from oracle import mmca_oracle as M
    g = golden(case)
    model = M.make_model(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(model["SM"], g["state_matrix"])
    an = M.Anneal(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]))
    np.testing.assert_array_equal(M.generate_from_hidden(g["W_gt"], g["s"]), g["y_clean"])
    params = M.check_params({"W": g["W"], "pi": float(g["pi"]), "sigma": float(g["sigma"])})
    assert np.abs(params["W"]).min() >= M.TOL and (np.abs(g["W"]) < M.TOL).any()
    vec = flavour == "vec"
    cand = (M.select_hprimes_vec if vec else M.select_hprimes_loop)(params["W"], g["y"], model["Hprime"])
    if vec:     # Gram form: rounding may swap near-ties; the sets must agree wherever the distances are distinct
        d = ((params["W"].T[None] - g["y"][:, None, :]) ** 2).sum(axis=2)
        for n in np.where((cand != g["candidates"]).any(axis=1))[0]:
            np.testing.assert_allclose(d[n, cand[n]], d[n, g["candidates"][n]], rtol=1e-10)
        cand = g["candidates"]
    else:
        assert np.array_equal(cand, g["candidates"])
    logpj = (M.e_step_vec if vec else M.e_step_loop)(an, params["W"], params["pi"], params["sigma"], g["y"], cand,
                                                     model["SM"], model["state_abs"])
    np.testing.assert_allclose(logpj, g["logpj"], rtol=1e-10, atol=1e-9)
    new, log = M.m_step(an, model, params["W"], params["pi"], params["sigma"], g["y"], cand, g["logpj"], vec=vec)
    assert log["N_use"] == int(g["N_use"])
    np.testing.assert_allclose(new["W"], g["W_new"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=1e-10)
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=1e-10)
    np.testing.assert_allclose(new["Q"], g["Q"], rtol=1e-11)


# ----------------------------------------------------------------------------- DSC (dsc_et.py)
def _dsc_cases():
    import glob, os
    from conftest import GOLDEN
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "dsc_step_*.npz")))


@pytest.mark.parametrize("case", _dsc_cases())
@pytest.mark.parametrize("flavour", ["loop", "vec"])
def test_dsc_step_matches_reference(case, flavour):
    from oracle import dsc_oracle as M
    g = golden(case)
    model = M.make_model(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]), g["states"])
    assert np.array_equal(model["SM"], g["state_matrix"])
    assert np.array_equal(model["SSM"], g["single_state_matrix"])
    assert np.array_equal(model["state_abs"], g["state_abs"])
    an = M.Anneal(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
    W, pi, sigma = g["W"], g["pi"], float(g["sigma"])
    vec = flavour == "vec"
    cand = (M.select_hprimes_vec if vec else M.select_hprimes_loop)(model, W, pi, sigma, g["y"])
    if vec:   # Gram form: rounding may swap near-ties only
        best = M.select_scores_vec(model, W, pi, sigma, g["y"])
        for n in np.where((cand != g["candidates"]).any(axis=1))[0]:
            np.testing.assert_allclose(best[n, cand[n]], best[n, g["candidates"][n]], rtol=1e-10)
        cand = g["candidates"]
    else:
        assert np.array_equal(cand, g["candidates"])
    logpj = (M.e_step_vec if vec else M.e_step_loop)(an, model, W, pi, sigma, g["y"], cand)
    np.testing.assert_allclose(logpj, g["logpj"], rtol=1e-10, atol=1e-9)
    new, log = M.m_step(an, model, W, pi, sigma, g["y"], cand, g["logpj"], vec=vec)
    assert log["N_use"] == int(g["N_use"])
    np.testing.assert_allclose(log["L"], float(g["L"]), rtol=1e-12)
    np.testing.assert_allclose(log["prior_mass"], float(g["prior_mass"]), rtol=1e-12)
    cond = np.linalg.cond(log["stats"]["Wq"])
    np.testing.assert_allclose(new["W"], g["W_new"], rtol=0, atol=max(1e-9, 1e-14 * cond) * np.abs(g["W_new"]).max())
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=1e-10)
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=1e-10)
    assert new["Q"] == float(g["Q"]) == 0.0


# ----------------------------------------------------------------------------- TSC (tsc_et.py)
def _tsc_cases():
    import glob, os
    from conftest import GOLDEN
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "tsc_step_*.npz")))


@pytest.mark.parametrize("case", _tsc_cases())
@pytest.mark.parametrize("flavour", ["loop", "vec"])
def test_tsc_step_matches_reference(case, flavour):
    from oracle import tsc_oracle as M
    g = golden(case)
    model = M.make_model(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(model["SM"], g["state_matrix"]) and np.array_equal(model["SSM"], g["single_state_matrix"])
    assert model["no_states"] == int(g["no_states"]) and np.array_equal(model["state_abs"], g["state_abs"])
    an = M.Anneal(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
    W, pi, sigma = g["W"], float(g["pi"]), float(g["sigma"])
    vec = flavour == "vec"
    cand = (M.select_hprimes_vec if vec else M.select_hprimes_loop)(model, W, pi, sigma, g["y"])
    if vec:   # Gram form: rounding may swap near-ties only
        R = M.select_scores_vec(model, W, g["y"])
        H = model["H"]
        for n in np.where((cand != g["candidates"]).any(axis=1))[0]:
            best = np.maximum(R[n, :H], R[n, H:])
            np.testing.assert_allclose(np.sort(best[cand[n]]), np.sort(best[g["candidates"][n]]), rtol=1e-9)
        cand = g["candidates"]
    else:
        assert np.array_equal(cand, g["candidates"])
    logpj = (M.e_step_vec if vec else M.e_step_loop)(an, model, W, pi, sigma, g["y"], cand)
    np.testing.assert_allclose(logpj, g["logpj"], rtol=1e-10, atol=1e-9)
    new, log = M.m_step(an, model, W, pi, sigma, g["y"], cand, g["logpj"], vec=vec)
    assert log["N_use"] == int(g["N_use"])
    np.testing.assert_allclose(log["L"], float(g["L"]), rtol=1e-12)
    cond = np.linalg.cond(log["stats"]["Wq"])
    np.testing.assert_allclose(new["W"], g["W_new"], rtol=0, atol=max(1e-9, 1e-14 * cond) * np.abs(g["W_new"]).max())
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=1e-10)
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=1e-10)
    assert (M.last_position_mask(g["candidates"]).all(axis=1).mean() < 1.0) == \
        bool(np.any([len(set(r)) < len(r) for r in g["candidates"]]))


# ----------------------------------------------------------------------------- GSC (gsc_et.py)
def _gsc_cases():
    import glob, os
    from conftest import GOLDEN
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "gsc_step_*.npz")))


def _gsc_params(g):
    sig = g["sigma_sq"]
    return {"W": g["W"], "pi": g["pi"], "mu": g["mu"], "psi_sq": g["psi_sq"],
            "sigma_sq": float(sig) if sig.ndim == 0 else sig}


@pytest.mark.parametrize("case", _gsc_cases())
def test_gsc_step_matches_reference(case):
    from oracle import gsc_oracle as G
    g = golden(case)
    model = G.make_model(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(model["SM"], g["state_matrix"])
    params = _gsc_params(g)
    an = G.Anneal(T=float(g["T"]))
    cand = G.select_hprimes(params, g["y"], model["Hprime"])
    assert np.array_equal(cand, g["candidates"])
    np.testing.assert_allclose(G.compute_lpj(model, params, g["y"], cand), g["logpj"], rtol=1e-9, atol=1e-8)
    suff = G.e_step(an, model, params, g["y"], cand)
    np.testing.assert_allclose(suff["xpt_s"], g["xpt_s"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(suff["xpt_sz"], g["xpt_sz"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(suff["xpt_ss"].sum(0), g["sum_xpt_ss"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(suff["xpt_szsz"].sum(0), g["sum_xpt_szsz"], rtol=1e-9, atol=1e-10)
    new = G.m_step(model, params, suff, g["y"])
    for k in ("W", "pi", "mu", "psi_sq", "sigma_sq"):
        ref = g[k + "_new"]
        np.testing.assert_allclose(new[k], ref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=k)
