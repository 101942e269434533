"""DSC parity on the GPU: HIP path (through the C ABI) vs golden vectors minted from the reference
(tests/golden/dsc_step_*.npz) and vs the oracle at sizes it finishes in seconds.  float64 kernels:
held to 1e-8 on W/pi/sigma/L (BASELINE asks 1e-4)."""
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
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "dsc_step_*.npz")))


def _check_candidates(cand, ref, best):
    """Best-first ranking of the per-latent best singleton log-joint; rows may differ from the reference's
    only where those scores tie to rounding (the device ranks the Gram form)."""
    cand = np.asarray(cand)
    bad = np.where((cand != ref).any(axis=1))[0]
    if bad.size:
        np.testing.assert_allclose(np.take_along_axis(best[bad], cand[bad], 1),
                                   np.take_along_axis(best[bad], ref[bad], 1), rtol=1e-9)
    return bad.size


@pytest.mark.parametrize("case", _cases())
def test_dsc_step_matches_reference_golden(case):
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    from oracle import dsc_oracle as M
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    g = golden(case)
    D, H, Hp, gamma = int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"])
    m = DSC_ET(D, H, Hp, gamma, states=g["states"])
    assert np.array_equal(m.state_matrix, g["state_matrix"])
    assert np.array_equal(m.single_state_matrix, g["single_state_matrix"])
    assert np.array_equal(m.state_abs, g["state_abs"])
    an = _An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
    params = {"W": g["W"].copy(), "pi": g["pi"].copy(), "sigma": float(g["sigma"])}
    h = dlog.set_handler(("N_use", "L", "prior_mass"), StoreInMemory)
    try:
        params = m.check_params(params)
        data = m.select_Hprimes(params, {"y": g["y"]})
        model = M.make_model(D, H, Hp, gamma, g["states"])
        best = M.select_scores_vec(model, params["W"], params["pi"], params["sigma"], g["y"])
        assert _check_candidates(data["candidates"], g["candidates"], best) == 0
        ss = m.E_step(an, params, data)
        new = m.M_step(an, params, ss, data)
    finally:
        dlog.remove_handler(h)
    np.testing.assert_allclose(np.asarray(ss["logpj"]), g["logpj"], rtol=1e-10, atol=1e-9)
    assert int(h.tables["N_use"][0]) == int(g["N_use"])
    np.testing.assert_allclose(h.tables["L"][0], float(g["L"]), rtol=1e-10)
    np.testing.assert_allclose(h.tables["prior_mass"][0], float(g["prior_mass"]), rtol=1e-12)
    np.testing.assert_allclose(new["W"], g["W_new"], rtol=0, atol=1e-8 * np.abs(g["W_new"]).max())
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=1e-9)
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=1e-9)
    assert new["Q"] == 0.0 and new["W"].shape == (D, H)
    # foreign NumPy inputs take the same kernels
    new2 = m.M_step(an, params, {"logpj": g["logpj"]}, {"y": g["y"], "candidates": g["candidates"]})
    np.testing.assert_allclose(new2["W"], g["W_new"], rtol=0, atol=1e-8 * np.abs(g["W_new"]).max())
    np.testing.assert_allclose(new2["pi"], g["pi_new"], rtol=1e-9)


@pytest.mark.parametrize("D,H,Hp,gamma,N,T,ncut,states", [
    (256, 128, 6, 3, 1500, 1.0, 0.0, [-1., 0., 1.]),
    (100, 70, 5, 4, 333, 1.6, 0.6, [0., 1., 2.]),
    (40, 20, 3, 2, 65, 1.0, 1.0, [-2., -1., 0., 1., 2.]),
    (64, 256, 8, 3, 3000, 1.0, 0.0, [-1., 0., 1.]),
    (48, 24, 10, 2, 130, 1.0, 0.0, [-1., 0., 1.])])       # H' > 8: the 16-wide instantiation of the row kernels
def test_dsc_step_matches_oracle(D, H, Hp, gamma, N, T, ncut, states):
    from oracle import dsc_oracle as M
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    rng = np.random.RandomState(D + H + N)
    states = np.array(states)
    K = len(states)
    pi_gt = np.where(states == 0, 1 - 2.0 / H, (2.0 / H) / (K - 1))      # ~2 active latents per datapoint
    W_gt = rng.normal(size=(D, H)) * 2.0
    s = rng.choice(states, size=(N, H), p=pi_gt)
    y = s @ W_gt.T + rng.normal(size=(N, D))
    pi0 = pi_gt * rng.uniform(0.8, 1.25, size=K)
    params = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": pi0 / pi0.sum(), "sigma": 1.1}
    model = M.make_model(D, H, Hp, gamma, states)
    an = M.Anneal(T=T, Ncut_factor=ncut, anneal_prior=(T != 1.0))
    m = DSC_ET(D, H, Hp, gamma, states=states)
    dan = _An(T=T, Ncut_factor=ncut, anneal_prior=(T != 1.0))
    data = m.select_Hprimes(params, {"y": y})
    best = M.select_scores_vec(model, params["W"], params["pi"], params["sigma"], y)
    ref_cand = M.select_hprimes_vec(model, params["W"], params["pi"], params["sigma"], y)
    _check_candidates(data["candidates"], ref_cand, best)
    cand = np.asarray(data["candidates"])
    logpj = M.e_step_vec(an, model, params["W"], params["pi"], params["sigma"], y, cand)
    ss = m.E_step(dan, params, data)
    np.testing.assert_allclose(np.asarray(ss["logpj"]), logpj, rtol=1e-10, atol=1e-9)
    ref, log = M.m_step(an, model, params["W"], params["pi"], params["sigma"], y, cand, logpj, vec=True)
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    h = dlog.set_handler(("N_use", "L"), StoreInMemory)
    try:
        new = m.M_step(dan, params, ss, data)
    finally:
        dlog.remove_handler(h)
    assert int(h.tables["N_use"][0]) == log["N_use"]
    np.testing.assert_allclose(h.tables["L"][0], log["L"], rtol=1e-10)
    cond = np.linalg.cond(log["stats"]["Wq"])
    np.testing.assert_allclose(new["W"], ref["W"], rtol=0, atol=max(1e-8, 1e-13 * cond) * np.abs(ref["W"]).max())
    np.testing.assert_allclose(new["pi"], ref["pi"], rtol=1e-9)
    np.testing.assert_allclose(new["sigma"], ref["sigma"], rtol=1e-9)


def test_dsc_em_recovers_parameters():
    """EM through the reference-shaped step loop on ternary data: the free energy rises and sigma approaches
    the generating value."""
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    D, H, Hp, gamma, N = 48, 12, 5, 3, 6000
    rng = np.random.RandomState(3)
    states = np.array([-1., 0., 1.])
    W_gt = rng.normal(size=(D, H)) * 3.0
    s = rng.choice(states, size=(N, H), p=[0.1, 0.8, 0.1])
    y = s @ W_gt.T + rng.normal(size=(N, D))
    m = DSC_ET(D, H, Hp, gamma, states=states)
    p = {"W": W_gt + 0.5 * rng.normal(size=(D, H)), "pi": np.array([0.15, 0.7, 0.15]), "sigma": 2.0}
    h = dlog.set_handler(("L",), StoreInMemory)
    try:
        for _ in range(25):
            p = m.step(_An(T=1.0), p, {"y": y})
    finally:
        dlog.remove_handler(h)
    L = [float(v) for v in h.tables["L"]]
    assert L[-1] > L[0] and all(b > a - 1e-9 for a, b in zip(L[5:], L[6:]))
    assert p["sigma"] < 1.9 and np.isfinite(p["W"]).all()
    np.testing.assert_allclose(p["pi"], [0.1, 0.8, 0.1], atol=0.05)


@pytest.mark.parametrize("tag,kw", [("plain", dict(topK=5, adaptive=False)), ("adaptive", dict(topK=4, adaptive=True)),
                                    ("capped", dict(topK=3, adaptive=True, Hprime_max=5, gamma_max=3, logprob=True))])
def test_dsc_inference_matches_reference(tag, kw, capsys):
    """DSC_ET.inference (dsc_et.py:927-1059) against the reference's golden outputs, adaptive re-runs
    (with their 1-D state_abs prior) included."""
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    g = golden("dsc_inference.npz")
    D, H, Hp, gamma = int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"])
    m = DSC_ET(D, H, Hp, gamma, states=g["states"])
    S0, abs0 = m.no_states, m.state_abs.copy()
    res = m.inference(_An(T=1.0), {"W": g["W"].copy(), "pi": g["pi"].copy(), "sigma": float(g["sigma"])},
                      {"y": g["y"]}, **kw)
    assert (m.Hprime, m.gamma, m.no_states) == (Hp, gamma, S0) and np.array_equal(m.state_abs, abs0)
    assert np.array_equal(res["gamma"], g[tag + "_gamma"]) and np.array_equal(res["Hprime"], g[tag + "_Hprime"])
    assert res["s"].dtype == np.int8 and np.array_equal(res["s"], g[tag + "_s"])
    np.testing.assert_allclose(res["p"], g[tag + "_p"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(res["m"], g[tag + "_m"], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("model,T,D,H,Hp,gamma,N", [("dsc", 1.0, 96, 128, 6, 3, 5003), ("tsc", 1.0, 96, 128, 6, 3, 5003),
                                                  ("dsc", 3.0, 40, 20, 4, 3, 333), ("tsc", 2.0, 40, 50, 5, 2, 1000),
                                                  ("dsc", 1.0, 64, 200, 9, 2, 9000)])
def test_estep_pass_with_mstep_statistics_matches_two_passes(model, T, D, H, Hp, gamma, N):
    """Inside `step` with no data truncation ahead the E-step kernel also produces the M-step's row statistics
    (pm_dsc_estep_mstats_f64: posterior weights from the exponentials of its log-sum-exp).  Same statistics buffer, E[s]
    rows and parameters as the two-pass form (pm_dsc_estep_f64 + pm_dsc_mstep_rows_nz_f64), which the goldens pin."""
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    rng = np.random.RandomState(D + H + N)
    W_gt = 2.0 * rng.normal(size=(D, H))
    u = rng.random_sample((N, H))
    y = ((u < 1.5 / H).astype(float) - (u > 1 - 1.5 / H)) @ W_gt.T + rng.normal(size=(N, D))
    W0 = W_gt + 0.1 * rng.normal(size=(D, H))
    out = {}
    for fuse in (True, False):
        if model == "dsc":
            m = DSC_ET(D, H, Hp, gamma, states=np.array([-1., 0., 1.]))
            p = {"W": W0, "pi": np.array([1.5 / H, 1 - 3.0 / H, 1.5 / H]), "sigma": 1.0}
        else:
            m = TSC_ET(D, H, Hp, gamma)
            p = {"W": W0, "pi": 3.0 / H, "sigma": 1.0}
        m.fuse_mstats = fuse
        names = []
        orig = m._call
        m._call = lambda label, name, *a, _o=orig, _n=names: (_n.append(name), _o(label, name, *a))[1]
        new = m.step(_An(T=T), dict(p), {"y": y})
        st = m._ws["dsc_stats" if model == "dsc" else "tsc_stats"].cpu().numpy().copy()
        out[fuse] = (new, st, m._ws["expect"].cpu().numpy().copy(), names)
    a, b = out[True], out[False]
    from prosper_amd import _lib
    K, flags = (3, 0) if model == "dsc" else (3, 3)
    can = bool(_lib.load().pm_dsc_estep_mstats_supported(H, Hp, m.no_states if model == "dsc" else m.state_matrix.shape[0], K, flags))
    assert can                                # (every shape of this test fits; where the layout does not: two passes, silently)
    if can:
        assert "pm_dsc_estep_mstats_f64" in a[3] and "pm_dsc_mstep_rows_nz_f64" not in a[3] and "pm_dsc_estep_f64" not in a[3]
    else:
        assert "pm_dsc_estep_mstats_f64" not in a[3] and "pm_dsc_estep_f64" in a[3]
    assert "pm_dsc_estep_mstats_f64" not in b[3] and "pm_dsc_estep_f64" in b[3]
    # E[s] rows: weights below e^-37 of the row maximum (fused) / e^-60 of the evidence (two passes) are dropped
    np.testing.assert_allclose(a[2], b[2], rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(a[1][:-1], b[1][:-1], rtol=1e-9, atol=1e-11 * np.abs(b[1]).max())
    for k in ("W", "pi", "sigma"):
        np.testing.assert_allclose(a[0][k], b[0][k], rtol=1e-9, atol=1e-12)
    # ... and a truncation step takes the two-pass form on its own
    m.fuse_mstats = True
    names.clear()
    m.step(_An(T=T, Ncut_factor=0.5), dict(p), {"y": y})
    assert "pm_dsc_estep_f64" in names and "pm_dsc_estep_mstats_f64" not in names


@pytest.mark.parametrize("model", ["dsc", "tsc"])
def test_em_loop_speculation_is_transparent(model):
    """The M-step leaves the next step's W^T, Gram matrix and scores on the device (DeviceCAModel._seed_next).  A
    trajectory must not depend on it -- also when the caller replaces W, or edits the returned array in place."""
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    D, H, Hp, gamma, N = 40, 20, 4, 3, 600
    rng = np.random.RandomState(21)
    W_gt = rng.normal(size=(D, H)) * 2.0
    u = rng.random_sample((N, H))
    S = (u < 1.0 / H).astype(float) - (u > 1 - 1.0 / H).astype(float)
    y = S @ W_gt.T + rng.normal(size=(N, D))
    W0 = W_gt + 0.1 * rng.normal(size=(D, H))
    runs = []
    for spec in (True, False):
        if model == "dsc":
            m = DSC_ET(D, H, Hp, gamma, states=np.array([-1., 0., 1.]))
            p = {"W": W0.copy(), "pi": np.array([1.0 / H, 1 - 2.0 / H, 1.0 / H]), "sigma": 1.0}
        else:
            m = TSC_ET(D, H, Hp, gamma)
            p = {"W": W0.copy(), "pi": 2.0 / H, "sigma": 1.0}
        m.speculate = spec
        taken = []
        take = m._take_seed
        m._take_seed = lambda W, res: taken.append(take(W, res)) or taken[-1]
        hit = []
        for it in range(5):
            if it == 3:
                p["W"] = p["W"] * (1.0 + 1e-3 * np.cos(np.arange(D * H).reshape(D, H)))
            if it == 4:
                p["W"][0, 0] += 0.01
            before = len(taken)
            p = m.step(_An(T=1.0), p, {"y": y})
            hit.append(any(t is not None for t in taken[before:]))      # (the seed is only looked at when there is one)
        # steps 1 and 2 ran on the seeded parameters, 0 (nothing seeded yet), 3 and 4 (W edited) must not.  (Step 2's
        # seed is void when the device rejected the warm start of step 1's inverse: W then comes from the repeated,
        # refined solve -- DeviceCAModel._solve_accurate -- not from the solution the seed was computed from.)
        assert hit[:2] + hit[3:] == ([False, True, False, False] if spec else [False] * 4) and (spec or not hit[2])
        runs.append(p)
    for k in ("W", "pi", "sigma"):
        np.testing.assert_allclose(runs[0][k], runs[1][k], rtol=1e-8, atol=1e-11, err_msg=k)


@pytest.mark.parametrize("model,T", [("dsc", 1.0), ("tsc", 1.0), ("dsc", 60.0)])
def test_sparse_wp_from_nonzero_lists(model, T, monkeypatch):
    """DSC / TSC M-step: the per-datapoint pass leaves the non-zeros of every E[s] row as a list as well and
    Wp = E[s]^T Y (dsc_et.py:703-735) is accumulated from the lists (pm_wp_sparse_f64); rows with more than 16 non-zeros
    (hot temperature) are counted and the dense product runs instead, decided on the device.  Same statistics and
    parameters as the dense product; the lists hold exactly the non-zeros of the dense rows."""
    from prosper_amd import _lib
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    D, H, Hp, gamma, N = 96, 128, 6, 3, 5000
    rng = np.random.RandomState(11)
    W_gt = 2.0 * rng.normal(size=(D, H))
    u = rng.random_sample((N, H))
    y = ((u < 1.5 / H).astype(float) - (u > 1 - 1.5 / H)) @ W_gt.T + rng.normal(size=(N, D))
    W0 = W_gt + 0.1 * rng.normal(size=(D, H))
    out = {}
    for sparse in ("1", "0"):
        if model == "dsc":
            m = DSC_ET(D, H, Hp, gamma, states=np.array([-1., 0., 1.]))
            p = {"W": W0, "pi": np.array([1.5 / H, 1 - 3.0 / H, 1.5 / H]), "sigma": 1.0}
        else:
            m = TSC_ET(D, H, Hp, gamma)
            p = {"W": W0, "pi": 3.0 / H, "sigma": 1.0}
        m.sparse_wp = sparse == "1"
        names = []
        orig = m._call
        m._call = lambda label, name, *a, _o=orig, _n=names: (_n.append(name), _o(label, name, *a))[1]
        new = m.step(_An(T=T), dict(p), {"y": y})
        st = m._ws["dsc_stats" if model == "dsc" else "tsc_stats"].cpu().numpy().copy()
        out[sparse] = (new, st, m._ws["expect"].cpu().numpy().copy(), names)
        if sparse == "1":
            idx = m._ws["nz_idx"].cpu().numpy().view(np.uint16).astype(np.int64)
            val = m._ws["nz_val"].cpu().numpy()
    a, b = out["1"], out["0"]
    assert "pm_wp_sparse_f64" in a[3] and "pm_gemm_tn_acc_gated_f64" in a[3] and "pm_wp_sparse_f64" not in b[3]
    E = a[2]
    nnz = (E != 0).sum(axis=1)
    assert int(a[1][-1]) == int((nnz > 16).sum())
    assert (int(a[1][-1]) > 0) == (T > 10)
    ok = nnz <= 16
    assert np.array_equal((idx != 0xFFFF).sum(axis=1)[ok], nnz[ok])
    rebuilt = np.zeros_like(E)
    rows = np.repeat(np.arange(N), 16).reshape(N, 16)
    sel = (idx != 0xFFFF) & ok[:, None]
    rebuilt[rows[sel], idx[sel]] = val[sel]
    assert np.array_equal(rebuilt[ok], E[ok])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-12, atol=1e-300)
    sa, sb = a[1].copy(), b[1].copy()
    sa[-1] = sb[-1] = 0.0
    np.testing.assert_allclose(sa, sb, rtol=1e-9, atol=1e-11 * np.abs(sb).max())
    Wp_ref = E.T @ y
    np.testing.assert_allclose(a[1][:H * D].reshape(H, D), Wp_ref, rtol=1e-10, atol=1e-11 * np.abs(Wp_ref).max())
    np.testing.assert_allclose(a[0]["W"], b[0]["W"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("model", ["dsc", "tsc"])
def test_full_shard_against_oracle_rows(model):
    """DSC / TSC at the bench's dimensions (D=256, H=128, H'=6, gamma=3) on a shard large enough that every workgroup of
    the sixteen-lane row kernels walks several groups of datapoints (N = 70000 > 16 x the resident grid): candidates,
    log-joints and E[s] rows of 400 sampled datapoints against the oracle on those rows alone, and the M-step's
    statistics against the dense product of the device's own E[s]."""
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    D, H, Hp, gamma, N = 256, 128, 6, 3, 70000
    rng = np.random.RandomState(5)
    W_gt = 2.0 * rng.normal(size=(D, H))
    u = rng.random_sample((N, H))
    y = ((u < 1.0 / H).astype(float) - (u > 1 - 1.0 / H)) @ W_gt.T + rng.normal(size=(N, D))
    W0 = W_gt + 0.1 * rng.normal(size=(D, H))
    states = np.array([-1., 0., 1.])
    if model == "dsc":
        from oracle import dsc_oracle as M
        m = DSC_ET(D, H, Hp, gamma, states=states)
        pi = np.array([1.0 / H, 1 - 2.0 / H, 1.0 / H])
        om = M.make_model(D, H, Hp, gamma, states)
    else:
        from oracle import tsc_oracle as M
        m = TSC_ET(D, H, Hp, gamma)
        pi = 2.0 / H
        om = M.make_model(D, H, Hp, gamma)
    params = {"W": W0, "pi": pi, "sigma": 1.0}
    an = _An(T=1.0)
    data = m.select_Hprimes(params, {"y": y})
    ss = m.E_step(an, params, data)
    idx = np.sort(rng.choice(N, 400, replace=False))
    idx[-1] = N - 1                                             # the last datapoint of the last (ragged) group too
    cand = np.asarray(data["candidates"][idx])
    oan = M.Anneal(T=1.0, Ncut_factor=0.0, anneal_prior=False)
    ref_cand = M.select_hprimes_vec(om, W0, pi, 1.0, y[idx])
    if model == "dsc":
        _check_candidates(cand, ref_cand, M.select_scores_vec(om, W0, pi, 1.0, y[idx]))
    else:
        best = M.select_scores_vec(om, W0, y[idx])
        for r in range(len(idx)):       # (TSC candidates may repeat a latent; equal scores may swap)
            np.testing.assert_allclose(np.sort(best[r, cand[r]]), np.sort(best[r, ref_cand[r]]), rtol=1e-9)
    logpj = M.e_step_vec(oan, om, W0, pi, 1.0, y[idx], cand)
    got = np.asarray(ss["logpj"][idx])
    np.testing.assert_allclose(got, logpj, rtol=1e-10, atol=1e-9)
    # round 6: EVERY row of the shard -- candidates (equal up to ties of the ranked scores) and all log-joints
    worst, moved = 0.0, 0
    cand_all, lp_all = np.asarray(data["candidates"]), ss["logpj"]
    for lo in range(0, N, 4096):
        y_c, c_c = y[lo:lo + 4096], cand_all[lo:lo + 4096]
        ref_c = M.select_hprimes_vec(om, W0, pi, 1.0, y_c)
        if model == "dsc":
            moved += _check_candidates(c_c, ref_c, M.select_scores_vec(om, W0, pi, 1.0, y_c))
        else:
            best_c = M.select_scores_vec(om, W0, y_c)
            np.testing.assert_allclose(np.sort(np.take_along_axis(best_c, c_c, 1), axis=1),
                                       np.sort(np.take_along_axis(best_c, ref_c, 1), axis=1), rtol=1e-9)
        lp_c = M.e_step_vec(oan, om, W0, pi, 1.0, y_c, c_c)
        got_c = np.asarray(lp_all[lo:lo + 4096])
        worst = max(worst, float(np.max(np.abs(got_c - lp_c) / (1e-9 + 1e-10 * np.abs(lp_c)))))
    assert worst <= 1.0, "log-joints: %.2f times the tolerance (rtol 1e-10, atol 1e-9)" % worst
    assert moved <= N // 1000, moved
    new = m.M_step(an, params, ss, data)
    E = m._ws["expect"].cpu().numpy()
    q = np.exp(logpj - np.logaddexp.reduce(logpj, axis=1, keepdims=True))
    stats = m._ws["dsc_stats" if model == "dsc" else "tsc_stats"].cpu().numpy()
    Wp_ref = E.T @ y
    np.testing.assert_allclose(stats[:H * D].reshape(H, D), Wp_ref, rtol=1e-10, atol=1e-11 * np.abs(Wp_ref).max())
    assert np.isfinite(new["W"]).all() and E.shape == (N, H) and q.shape[0] == 400
    # E[s] rows of the sampled datapoints: the same kernels on those rows alone (one group per workgroup; that case is
    # held to the oracle by test_*_step_matches_oracle)
    m2 = type(m)(D, H, Hp, gamma, states=states) if model == "dsc" else type(m)(D, H, Hp, gamma)
    d2 = {"y": y[idx], "candidates": cand}
    ss2 = m2.E_step(an, params, d2)
    np.testing.assert_allclose(np.asarray(ss2["logpj"]), got, rtol=1e-12, atol=1e-12)
    m2.M_step(an, params, ss2, d2)
    np.testing.assert_allclose(E[idx], m2._ws["expect"].cpu().numpy(), rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("tsc,H,Hp,K,N", [(0, 128, 6, 3, 5001), (0, 50, 5, 5, 777), (0, 300, 4, 3, 403), (0, 10, 3, 2, 70),
                                          (1, 128, 6, 0, 5001), (1, 32, 4, 0, 333), (1, 256, 8, 0, 150)])
def test_fused_selection_matches_the_two_launch_form(tsc, H, Hp, K, N):
    """pm_xsc_select_f64 (ranking values formed in registers and ranked in one pass) against pm_dsc_select_scores_f64 /
    pm_tsc_select_scores_f64 + pm_bsc_select_estep_f64 (+ the modulo of tsc_et.py:210): the same candidates in the same
    order; a row may differ only where two ranking values agree to rounding (the two kernels may round a fused multiply-add
    differently)."""
    import ctypes
    from prosper_amd import _lib
    dev = torch.device("cuda", 0)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.RandomState(H + 3 * Hp + K)
    A = torch.from_numpy(rng.normal(size=(N, H)) * 30).to(dev)
    W = rng.normal(size=(H, 40))
    G = torch.from_numpy(W @ W.T).to(dev)
    yn = torch.zeros(N, dtype=torch.float64, device=dev)
    assert _lib.load().pm_xsc_select_supported(H, Hp, tsc)
    new = torch.full((N, Hp), -1, dtype=torch.int32, device=dev)
    if tsc:
        R = torch.empty((N, 2 * H), dtype=torch.float64, device=dev)
        _lib.call("pm_tsc_select_scores_f64", p(A), H, p(G), N, H, p(R), 2 * H, st)
        gd = torch.zeros((2 * H, 2 * H), dtype=torch.float64, device=dev)
        old = torch.empty((N, Hp), dtype=torch.int32, device=dev)
        _lib.call("pm_bsc_select_estep_f64", p(R), 2 * H, p(gd), p(yn), None, None, None, None, None, 0, 3, None, N, 2 * H,
                  Hp, 1 | 8, p(old), None, 0, None, st)
        old = torch.remainder(old, H)
        _lib.call("pm_xsc_select_f64", p(A), H, p(G), None, N, H, Hp, p(new), st)
    else:
        P = _lib.DscParams()
        vals = np.concatenate([[0.0], rng.permutation(np.arange(1, K))[:K - 1] * rng.choice([-1.0, 1.0], size=K - 1)])
        P.K, P.K0 = K, 0
        pi = rng.dirichlet(np.ones(K))
        for k in range(K):
            P.values[k] = float(vals[k])
            P.logpi[k] = float(np.log(pi[k]))
        P.pre1 = -0.37
        R = torch.empty((N, H), dtype=torch.float64, device=dev)
        _lib.call("pm_dsc_select_scores_f64", p(A), H, p(G), ctypes.byref(P), N, H, p(R), H, st)
        old = torch.empty((N, Hp), dtype=torch.int32, device=dev)
        _lib.call("pm_bsc_select_estep_f64", p(R), H, p(G), p(yn), None, None, None, None, None, 0, 3, None, N, H, Hp,
                  1 | 4 | 8, p(old), None, 0, None, st)
        _lib.call("pm_xsc_select_f64", p(A), H, p(G), ctypes.byref(P), N, H, Hp, p(new), st)
    old, new, Rh = old.cpu().numpy(), new.cpu().numpy(), R.cpu().numpy()
    assert new.min() >= 0 and new.max() < H
    bad = np.nonzero((old != new).any(axis=1))[0]
    assert len(bad) <= max(1, N // 1000), (len(bad), N)
    for n in bad:          # only where ranking values tie to rounding
        r = np.sort(Rh[n])
        gaps = np.abs(np.diff(r)) / np.maximum(np.abs(r[1:]), 1e-300)
        assert gaps.min() < 1e-12, (n, old[n], new[n])
