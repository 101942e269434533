"""MCA parity on the GPU: HIP path (through the C ABI) vs golden vectors minted from the reference
(tests/golden/mca_step_*.npz) and vs the oracle at sizes it finishes in seconds.  float64 kernels:
held to 1e-8 on W/pi/sigma/Q (BASELINE asks 1e-4)."""
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
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "mca_step_*.npz")))


@pytest.mark.parametrize("case", _cases())
def test_mca_step_matches_reference_golden(case):
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    from prosper_amd.em.camodels.mca_et import MCA_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    g = golden(case)
    m = MCA_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    an = _An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]))
    params = {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}
    h = dlog.set_handler(("N_use",), StoreInMemory)
    try:
        params = m.check_params(params)
        data = m.select_Hprimes(params, {"y": g["y"]})
        ss = m.E_step(an, params, data)
        new = m.M_step(an, params, ss, data)
    finally:
        dlog.remove_handler(h)
    assert np.array_equal(np.asarray(data["candidates"]), g["candidates"])
    np.testing.assert_allclose(np.asarray(ss["logpj"]), g["logpj"], rtol=1e-10, atol=1e-9)
    assert int(h.tables["N_use"][0]) == int(g["N_use"])
    np.testing.assert_allclose(new["W"], g["W_new"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=1e-9)
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=1e-9)
    np.testing.assert_allclose(new["Q"], g["Q"], rtol=1e-10)
    assert new["W"].shape == (int(g["D"]), int(g["H"]))
    # foreign NumPy inputs take the same kernels
    new2 = m.M_step(an, params, {"logpj": g["logpj"]}, {"y": g["y"], "candidates": g["candidates"]})
    np.testing.assert_allclose(new2["W"], g["W_new"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(new2["Q"], g["Q"], rtol=1e-10)


@pytest.mark.parametrize("D,H,Hp,gamma,N,T,ncut", [(256, 128, 8, 3, 1500, 1.0, 0.0), (100, 70, 5, 4, 333, 1.6, 0.6),
                                                    (40, 20, 3, 2, 65, 1.0, 1.0), (700, 24, 4, 3, 90, 1.2, 0.0),
                                                    (1000, 16, 10, 2, 50, 1.0, 0.4),
                                                    (200, 40, 14, 2, 300, 1.0, 0.0),     # H' > 12: the M-step's V[16][2] tile,
                                                    (130, 30, 16, 2, 200, 1.3, 0.5)])    # ... slabs of 128 dimensions (round 4)
def test_mca_step_matches_oracle(D, H, Hp, gamma, N, T, ncut):
    from oracle import mca_oracle as M
    from prosper_amd.em.camodels.mca_et import MCA_ET
    rng = np.random.RandomState(D + H + N)
    W_gt = np.abs(rng.normal(size=(D, H))) * 2.0 + 0.1
    y, _ = M.generate_mca_data(W_gt, 2.0 / H, 1.0, N, rng)
    params = {"W": W_gt * (1 + 0.2 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.4 / H, "sigma": 1.1}
    model = M.make_model(D, H, Hp, gamma)
    an = M.Anneal(T=T, Ncut_factor=ncut)
    ref, log = M.em_step(an, model, dict(params), y, vec=True)
    m = MCA_ET(D, H, Hp, gamma)
    new = m.step(_An(T=T, Ncut_factor=ncut), dict(params), {"y": y})
    np.testing.assert_allclose(new["W"], ref["W"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(new["pi"], ref["pi"], rtol=1e-9)
    np.testing.assert_allclose(new["sigma"], ref["sigma"], rtol=1e-9)
    if np.isfinite(ref["Q"]):
        np.testing.assert_allclose(new["Q"], ref["Q"], rtol=1e-10)
    else:
        # the reference's log(sum(exp(logpj))) (mca_et.py:327) underflows to -inf once logpj < -745 (large D);
        # the device keeps the stabilised log-evidence (documented deviation)
        from scipy.special import logsumexp
        lp = log["logpj"]
        lb = logsumexp(lp / T, axis=1)
        keep = lb >= np.sort(lb)[-log["N_use"]]
        lAi = (H * np.log(1. - ref["pi"])) - ((D / 2) * np.log(2 * np.pi)) - (D * np.log(ref["sigma"]))
        np.testing.assert_allclose(new["Q"], lAi * log["N_use"] + logsumexp(lp[keep], axis=1).sum(), rtol=1e-10)


def test_mca_generate_data_rng_stream():
    """Host-side generate_data keeps the reference's RNG order (one random(H) per datapoint)."""
    from oracle import mca_oracle as M
    from prosper_amd.em.camodels.mca_et import MCA_ET
    rng = np.random.RandomState(4)
    W = np.abs(rng.normal(size=(12, 6))) + 0.1
    m = MCA_ET(12, 6, 3, 2)
    np.random.seed(9)
    d = m.generate_data({"W": W, "pi": 0.3, "sigma": 0.5}, 20)
    y, s = M.generate_mca_data(W, 0.3, 0.5, 20, np.random.RandomState(9))
    assert np.array_equal(d["s"], s)
    np.testing.assert_allclose(d["y"], y, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("cls_name", ["MCA_ET", "MMCA_ET"])
@pytest.mark.parametrize("D,H,Hp,gamma,N", [(256, 128, 8, 3, 700), (60, 33, 5, 4, 301), (512, 40, 4, 2, 130)])
def test_fused_estep_statistics_match_two_pass(cls_name, D, H, Hp, gamma, N):
    """pm_mca_estep_mstats_f64 (E-step + M-step statistics in one pass, running-maximum accumulation)
    against pm_mca_estep_f64 + pm_mca_mstep_rows_f64."""
    import importlib
    mod = importlib.import_module("prosper_amd.em.camodels." + ("mca_et" if cls_name == "MCA_ET" else "mmca_et"))
    cls = getattr(mod, cls_name)
    rng = np.random.RandomState(D + N)
    signed = cls_name == "MMCA_ET"
    W_gt = rng.normal(size=(D, H)) * 3.0 if signed else np.abs(rng.normal(size=(D, H))) * 2.0 + 0.1
    s = rng.random_sample((N, H)) < 2.0 / H
    y = np.zeros((N, D))
    for n in range(N):
        if s[n].any():
            t0 = W_gt.T[s[n]]
            y[n] = t0[np.argmax(np.abs(t0), axis=0), np.arange(D)] if signed else t0.max(axis=0)
    y += rng.normal(size=(N, D))
    p0 = {"W": W_gt * (1 + 0.1 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.4 / H, "sigma": 1.1}
    outs = []
    for fuse in (True, False):
        m = cls(D, H, Hp, gamma)
        m.fuse_em = fuse
        p = m.check_params({k: (v.copy() if hasattr(v, "copy") else v) for k, v in p0.items()})
        data = m.select_Hprimes(p, {"y": y})
        ss = m.E_step(_An(T=1.3), p, data)
        assert (ss["logpj"].fused is not None) == fuse
        new = m.M_step(_An(T=1.3), p, ss, data)
        outs.append((np.asarray(ss["logpj"]), new))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-12, atol=1e-10)
    for k in ("W", "pi", "sigma", "Q"):
        np.testing.assert_allclose(outs[0][1][k], outs[1][1][k], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("ncut", [0.0, 0.5])
def test_mca_state_set_in_any_order_with_single_candidate_states(ncut):
    """The row sums of the MCA kernels keep a prefix of the previous state's sum in registers (round 5): correct for ANY
    state set, not only itertools.combinations order -- shuffled states and single-candidate rows in the state matrix
    (a caller may edit `state_matrix`, as with the reference's attribute) against the oracle with the same matrix, on the
    fused pass (ncut = 0) and the two-pass kernels (truncation)."""
    from oracle import mca_oracle as M
    from prosper_amd.em.camodels.mca_et import MCA_ET
    D, H, Hp, gamma, N = 96, 24, 6, 3, 257
    rng = np.random.RandomState(11)
    W_gt = np.abs(rng.normal(size=(D, H))) * 2.0 + 0.1
    y, _ = M.generate_mca_data(W_gt, 2.0 / H, 1.0, N, rng)
    params = {"W": W_gt * (1 + 0.2 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.4 / H, "sigma": 1.1}
    model = M.make_model(D, H, Hp, gamma)
    SM = np.asarray(model["SM"]).copy()
    singles = np.zeros((3, Hp), dtype=SM.dtype)
    singles[0, 0] = singles[1, 3] = singles[2, Hp - 1] = 1
    SM = np.concatenate([SM, singles])[rng.permutation(SM.shape[0] + 3)]
    assert SM.shape[0] % 2 == 0 and SM.shape[0] < 64          # (the odd / multi-batch cases: test_mca_step_matches_oracle)
    SM = SM[:-1]                                              # ... an odd count here as well
    model = dict(model, SM=SM, state_abs=SM.sum(axis=1), S=SM.shape[0])
    ref, log = M.em_step(M.Anneal(T=1.0, Ncut_factor=ncut), model, dict(params), y, vec=True)
    m = MCA_ET(D, H, Hp, gamma)
    m.state_matrix, m.no_states, m.state_abs, m._masks_dev = SM.astype(np.uint8), SM.shape[0], SM.sum(axis=1), None
    new = m.step(_An(T=1.0, Ncut_factor=ncut), dict(params), {"y": y})
    np.testing.assert_allclose(new["W"], ref["W"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(new["pi"], ref["pi"], rtol=1e-9)
    np.testing.assert_allclose(new["sigma"], ref["sigma"], rtol=1e-9)
    np.testing.assert_allclose(new["Q"], ref["Q"], rtol=1e-10)


# ------------------------------------------------------------------------- config-5 properties
def test_config5_properties():
    """BASELINE config 5 dims (D=256 H=128 H'=8 gamma=3) at N = 6000, fused E+M pass (per-XCD copies of Wp/Wq and
    their fold): candidates are the H' smallest distances, log-evidences are the log-sum-exp of the rows, the
    packed statistics are additive over shards, the fused and the two-pass route agree."""
    from prosper_amd.em.camodels.mca_et import MCA_ET
    dev = torch.device("cuda", 0)
    D, H, Hp, gamma, N = 256, 128, 8, 3, 6000
    gen = torch.Generator(device=dev).manual_seed(5)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64).abs() * 2 + 0.1
    S = torch.rand(N, H, generator=gen, device=dev) < 2.0 / H
    Y = torch.where(S[:, None, :], W_gt[None, :, :].expand(N, D, H),
                    torch.zeros((), dtype=torch.float64, device=dev)).max(dim=2).values
    Y = Y + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    W0 = (W_gt * (1 + 0.1 * (2 * torch.rand(D, H, generator=gen, device=dev, dtype=torch.float64) - 1))).cpu().numpy()
    p = {"W": W0, "pi": 2.0 / H, "sigma": 1.0}
    an = _An(T=1.2)

    def run(rows, fuse):
        m = MCA_ET(D, H, Hp, gamma)
        m.fuse_em = fuse
        data = m.select_Hprimes(p, {"y": Y[rows].contiguous()})
        ss = m.E_step(an, p, data)
        fz = ss["logpj"].fused
        assert (fz is not None) == fuse
        buf = fz["stats"] if fuse else None                 # M_step completes (G1 GEMM) and all-reduces it in place
        new = m.M_step(an, p, ss, data)
        if buf is None:
            buf = m._ws["mca_stats"]
        n_doc = 3 * H * D + H + 4
        assert (buf[n_doc:] == 0).all()                     # per-XCD scratch: folded and cleared
        return data, ss, new, buf[:n_doc].clone(), m

    data, ss, new, stats_all, m = run(slice(0, N), True)
    cand = data["candidates"].tensor.long()
    Wt = torch.from_numpy(np.ascontiguousarray(W0.T)).to(dev)
    # (1) candidates: the H' latents with the smallest sum_d max(W_hd - y_d, 0)... as upstream (mca_et.py:90-111)
    sim = torch.clamp(Wt[None, :, :] - Y[:256, None, :], min=0).sum(-1)
    top = torch.topk(sim, Hp, dim=1, largest=False).indices
    assert (torch.sort(top, 1).values == torch.sort(cand[:256], 1).values).float().mean().item() > 0.999
    # (2) every row of log-joints is finite; its evidence is what the kernel stored
    logpj = ss["logpj"].tensor
    assert logpj.shape == (N, 1 + H + 28 + 56) and torch.isfinite(logpj).all()
    # (3) the documented statistics are additive over shards; the scratch tail is folded and cleared
    half = N // 2 + 7
    acc = torch.zeros_like(stats_all)
    for sl in (slice(0, half), slice(half, N)):
        acc += run(sl, True)[3]
    torch.testing.assert_close(acc[:3 * H * D + H], stats_all[:3 * H * D + H], rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(acc[-4:], stats_all[-4:], rtol=1e-11, atol=1e-9)
    # (4) fused and two-pass agree on the update
    _, _, new2, stats2, _ = run(slice(0, N), False)
    for k in ("W", "pi", "sigma", "Q"):
        np.testing.assert_allclose(new[k], new2[k], rtol=1e-9, atol=1e-12)
    assert np.isfinite(new["W"]).all() and 0 < new["pi"] < 1 and new["sigma"] > 0


def test_config5_full_shard_against_oracle():
    """BASELINE config 5 at one GPU's real share -- D=256 H=128 H'=8 gamma=3, N = 100 000 of the 800 000 -- through
    the shipped launches (selection scores, fused E-step + M-statistics pass).  Datapoints are independent given the
    parameters, so the vectorised oracle runs on ~500 sampled rows: candidate sets identical, log-joints to 1e-10;
    the sampled rows as a shard of their own reproduce the oracle's whole EM step."""
    from oracle import mca_oracle as M
    from prosper_amd.em.camodels.mca_et import MCA_ET
    dev = torch.device("cuda", 0)
    D, H, Hp, gamma, N = 256, 128, 8, 3, 100_000
    gen = torch.Generator(device=dev).manual_seed(55)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64).abs() * 2 + 0.1
    Y = torch.empty(N, D, dtype=torch.float64, device=dev)
    for lo in range(0, N, 20_000):
        S = torch.rand(20_000, H, generator=gen, device=dev) < 2.0 / H
        Y[lo:lo + 20_000] = torch.where(S[:, None, :], W_gt[None, :, :].expand(20_000, D, H),
                                        torch.zeros((), dtype=torch.float64, device=dev)).max(dim=2).values
    Y += torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    W0 = (W_gt * (1 + 0.1 * (2 * torch.rand(D, H, generator=gen, device=dev, dtype=torch.float64) - 1))).cpu().numpy()
    params = {"W": W0, "pi": 2.0 / H, "sigma": 1.0}
    an = _An(T=1.0)
    m = MCA_ET(D, H, Hp, gamma)
    data = m.select_Hprimes(dict(params), {"y": Y})
    ss = m.E_step(an, dict(params), data)
    assert ss["logpj"].fused is not None                      # the one-pass E-step + M-statistics kernel ran
    rng = np.random.RandomState(5)
    rows = np.unique(np.concatenate([rng.randint(0, N, size=400), np.arange(0, 64), np.arange(N - 64, N)]))
    idx = torch.from_numpy(rows).to(dev)
    y_s = Y[idx].cpu().numpy()
    model = M.make_model(D, H, Hp, gamma)
    cand_ref = M.select_hprimes_vec(W0, y_s, Hp)
    cand = data["candidates"].tensor[idx].cpu().numpy()
    assert np.array_equal(np.sort(cand, 1), np.sort(cand_ref, 1))
    lp_ref = M.e_step_vec(M.Anneal(T=1.0), W0, params["pi"], params["sigma"], y_s, cand, model["SM"], model["state_abs"])
    np.testing.assert_allclose(ss["logpj"].tensor[idx].cpu().numpy(), lp_ref, rtol=1e-10, atol=1e-9)
    new = m.M_step(an, dict(params), ss, data)
    assert np.isfinite(new["W"]).all() and 0 < new["pi"] < 1 and new["sigma"] > 0
    # the sample as its own shard: a whole EM step against the oracle
    ref, _ = M.em_step(M.Anneal(T=1.0), model, dict(params), y_s, vec=True)
    got = MCA_ET(D, H, Hp, gamma).step(an, dict(params), {"y": y_s})
    np.testing.assert_allclose(got["W"], ref["W"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose([got["pi"], got["sigma"]], [ref["pi"], ref["sigma"]], rtol=1e-9)
    # ... and EVERY row of the shard (round 6; 500 sampled rows before: the GPU boxes' hosts run the vectorised oracle at
    # ~3 k datapoints/s)
    worst, bad = 0.0, 0
    every = np.arange(0, N if os.environ.get("PM_FULL_PARITY") == "1" else N // 2)      # (half of them in the suite)
    for lo in range(0, len(every), 2048):
        r = every[lo:lo + 2048]
        ix = torch.from_numpy(r).to(dev)
        y_c = Y[ix].cpu().numpy()
        c_ref = M.select_hprimes_vec(W0, y_c, Hp)
        c_got = data["candidates"].tensor[ix].cpu().numpy()
        bad += int((np.sort(c_got, 1) != np.sort(c_ref, 1)).any(axis=1).sum())
        lp_c = M.e_step_vec(M.Anneal(T=1.0), W0, params["pi"], params["sigma"], y_c, c_got, model["SM"], model["state_abs"])
        got_c = ss["logpj"].tensor[ix].cpu().numpy()
        worst = max(worst, float(np.max(np.abs(got_c - lp_c) / (1e-9 + 1e-10 * np.abs(lp_c)))))
    assert bad == 0, "%d of %d datapoints with another candidate set than the oracle's" % (bad, len(every))
    assert worst <= 1.0, "log-joints: %.2f times the tolerance (rtol 1e-10, atol 1e-9)" % worst


def test_config5_full_shard_truncation_step_against_oracle():
    """A whole EM step of config 5 at one GPU's share (N = 100 000) on an annealed, data-truncating point of the reference's
    schedule (T = 1.4 -> rho = 3.5, Ncut_factor = 0.5) against the oracle on ALL rows (round 6): ``CAModel.step`` -- the pass
    with the uniform-exponent power that leaves per-datapoint records, the cut selected on the device, the LDS-privatised
    scatter -- returns the oracle's W, pi, sigma and Q.  The oracle's statistics are sums over the kept datapoints: its
    m_step runs on them chunk by chunk (the cut itself from the log-denominators of all rows, mca_et.py:237-262)."""
    from oracle import mca_oracle as M
    from prosper_amd.em.camodels.mca_et import MCA_ET
    dev = torch.device("cuda", 0)
    # (40 000 rows in the suite, 28 s; PM_FULL_PARITY=1: the whole share of 100 000, 69 s -- run and passing)
    D, H, Hp, gamma, N = 256, 128, 8, 3, (100_000 if os.environ.get("PM_FULL_PARITY") == "1" else 40_000)
    rng = np.random.RandomState(56)
    W_gt = np.abs(rng.normal(size=(D, H))) * 2 + 0.1
    y = np.empty((N, D))
    for lo in range(0, N, 10_000):
        s = rng.random_sample((10_000, H)) < 2.0 / H
        y[lo:lo + 10_000] = np.where(s[:, None, :], W_gt[None, :, :], 0.0).max(axis=2) + rng.normal(size=(10_000, D))
    params = M.check_params({"W": W_gt * (1 + 0.1 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.0 / H, "sigma": 1.0})
    T, ncut = 1.4, 0.5
    an = _An(T=T, Ncut_factor=ncut)
    m = MCA_ET(D, H, Hp, gamma)
    got = m.step(an, {k: (v.copy() if hasattr(v, "copy") else v) for k, v in params.items()}, {"y": torch.from_numpy(y).to(dev)})
    assert m.defer_stats
    model = M.make_model(D, H, Hp, gamma)
    oan = M.Anneal(T=T, Ncut_factor=ncut)
    K = 1 + H + model["SM"].shape[0]
    cand = np.empty((N, Hp), dtype=np.int64)
    logpj = np.empty((N, K))
    for lo in range(0, N, 2048):
        cand[lo:lo + 2048] = M.select_hprimes_vec(params["W"], y[lo:lo + 2048], Hp)
        logpj[lo:lo + 2048] = M.e_step_vec(oan, params["W"], params["pi"], params["sigma"], y[lo:lo + 2048], cand[lo:lo + 2048],
                                           model["SM"], model["state_abs"])
    beta = 1.0 / T
    corr = beta * logpj.max(axis=1)
    denoms = np.log(np.exp(beta * logpj - corr[:, None]).sum(axis=1)) + corr
    A_pg, B_pg = M.pi_gamma_factors(params["pi"], H, gamma)
    N_use = int(N * (1 - (1 - A_pg) * ncut))
    keep = np.nonzero(denoms >= np.sort(denoms, kind="mergesort")[-N_use])[0]
    N_use = len(keep)
    assert 0.5 * N < N_use < N
    tot, ld = None, 0.0
    flat = M.Anneal(T=T, Ncut_factor=0.0)                       # (the kept rows' statistics; the cut is applied above)
    for lo in range(0, N_use, 2048):
        r = keep[lo:lo + 2048]
        _, log = M.m_step(flat, model, params["W"], params["pi"], params["sigma"], y[r], cand[r], logpj[r], vec=True)
        tot = log["stats"] if tot is None else {k: tot[k] + log["stats"][k] for k in tot}
        ld += np.log(np.exp(logpj[r]).sum(axis=1)).sum()
    Wp, Wq = tot["Wp"].copy(), tot["Wq"].copy()                 # (the update: mca_et.py:333-377)
    tiny = np.finfo(np.float64).tiny
    Wp[Wq < tiny] = 0.0
    Wq[Wq < tiny] = tiny
    W_ref = (Wp / Wq).T
    pi_ref = A_pg / B_pg * params["pi"] * tot["pi"] / N_use
    sigma_ref = np.sqrt(tot["sigma"] / D / N_use)
    Q_ref = ((H * np.log(1. - pi_ref)) - ((D / 2) * np.log(2 * np.pi)) - (D * np.log(sigma_ref))) * N_use + ld
    np.testing.assert_allclose(got["W"], W_ref, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose([got["pi"], got["sigma"]], [pi_ref, sigma_ref], rtol=1e-9)
    np.testing.assert_allclose(got["Q"], Q_ref, rtol=1e-10)


@pytest.mark.parametrize("tag,kw", [("plain", dict(topK=5, adaptive=False)), ("adaptive", dict(topK=4, adaptive=True)),
                                    ("capped", dict(topK=3, adaptive=True, Hprime_max=5, gamma_max=3, logprob=True))])
def test_mca_inference_matches_reference(tag, kw, capsys):
    """CAModel.inference (camodels/__init__.py:256-375) of MCA_ET -- compute_lpj = select_Hprimes + E_step on the HIP
    path -- against the reference's own output: top-K states bit for bit, probabilities and marginals; the adaptive
    run regenerates the state table up to the H' / gamma the golden records."""
    from prosper_amd.em.camodels.mca_et import MCA_ET
    g = golden("mca_inference.npz")
    m = MCA_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    an = _An(T=1.0)
    res = m.inference(an, {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}, {"y": g["y"]}, **kw)
    assert (m.Hprime, m.gamma) == (int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(res["gamma"], g[tag + "_gamma"]) and np.array_equal(res["Hprime"], g[tag + "_Hprime"])
    assert res["s"].dtype == np.int8 and np.array_equal(res["s"], g[tag + "_s"])
    np.testing.assert_allclose(res["p"], g[tag + "_p"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(res["m"], g[tag + "_m"], rtol=1e-7, atol=1e-12)


def test_mca_em_loop_seeded_selection_is_transparent():
    """The M-step ranks the next step's candidates on the device behind its download (MCA_ET._seed_select); the next
    select_Hprimes adopts them iff it gets the W that M-step returned.  Same trajectory as the loop that never seeds,
    whether the caller feeds W straight back, replaces it, or edits it in place."""
    from prosper_amd.em.camodels.mca_et import MCA_ET
    from oracle import mca_oracle as MO
    D, H, Hp, gamma, N = 64, 32, 5, 3, 900
    rng = np.random.RandomState(7)
    Wm = np.abs(rng.normal(size=(D, H))) * 2 + 0.1
    y, _ = MO.generate_mca_data(Wm, 2.0 / H, 1.0, N, rng)
    p0 = {"W": Wm * (1 + 0.1 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.0 / H, "sigma": 1.0}
    runs = []
    for spec in (True, False):
        m = MCA_ET(D, H, Hp, gamma)
        m.speculate = spec
        launched = []
        sel = m._select_on_device
        m._select_on_device = lambda res, Wt: launched.append(1) or sel(res, Wt)
        p = {k: np.array(v, copy=True) for k, v in p0.items()}
        per_step = []
        for it in range(6):
            if it == 3:
                p["W"] = p["W"] * (1.0 + 1e-3 * np.cos(np.arange(D * H).reshape(D, H)))
            if it == 4:
                p["W"][0, 0] *= 1.01
            before = len(launched)
            p = m.step(_An(T=1.0), p, {"y": y})
            p = {k: p[k] for k in ("W", "pi", "sigma")}
            per_step.append(len(launched) - before)
        runs.append(p)
        # seeding: one selection per step (the M-step's, for the next step) wherever the seed was adopted, two where
        # select_Hprimes had to rank for itself (step 0, the replaced W, the in-place edit)
        assert per_step == ([2, 1, 1, 2, 2, 1] if spec else [1] * 6), per_step
    for k in ("W", "pi", "sigma"):
        # (the statistics are sums of f64 atomics: two runs of the same loop agree to ~1e-12, not bit for bit)
        np.testing.assert_allclose(runs[0][k], runs[1][k], rtol=1e-9, err_msg=k)


@pytest.mark.parametrize("signed,rho", [(False, 21.0), (True, 6.0), (False, 3.7)])
def test_power_tables_on_the_device_match_numpy(signed, rho):
    """pm_mca_tables_f64 against the reference's host formulas (mca_et.py:218-227, mmca_et.py:250-260)."""
    import ctypes
    import torch
    from prosper_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    lib = _lib.load()
    rng = np.random.RandomState(3)
    H, D = 37, 300
    Wt = np.abs(rng.normal(size=(H, D))) * 3 + 1e-3
    if signed:
        Wt *= rng.choice([-1.0, 1.0], size=(H, D))
    dev = torch.device("cuda", 0)
    d_W = torch.from_numpy(Wt).to(dev)
    tabs = torch.empty((3, H, D), dtype=torch.float64, device=dev)
    wn = torch.empty((H,), dtype=torch.float64, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.pm_mca_tables_f64(p(d_W), H, D, ctypes.c_double(rho), p(tabs), p(wn),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    t = tabs.cpu().numpy()
    Wl = np.log(np.abs(Wt))
    assert np.array_equal(t[0], Wt)
    # (exp(rho log w): the logarithm's rounding enters rho |log w| times -- both sides carry that, differently)
    np.testing.assert_allclose(t[1], np.sign(Wt) * np.exp(rho * Wl), rtol=4e-14)
    np.testing.assert_allclose(t[2], np.exp((rho - 1.0) * Wl), rtol=4e-14)
    np.testing.assert_allclose(wn.cpu().numpy(), (Wt * Wt).sum(axis=1), rtol=1e-14)
