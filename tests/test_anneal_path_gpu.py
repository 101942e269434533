"""The reference's own annealing schedule on the fast path (round 6): every shipped schedule ramps T and Ncut_factor
(examples/barstests/bars-learning.py:77-80), so 49 of a run's 50 steps are data-truncation steps (bsc_et.py:247-258) and
the annealing point moves on most of them.  Parity of
  * the DEFERRED statistics -- the E-step pass leaves per-datapoint records, pm_bsc_defer_apply_f64 adds the ones above
    the cut, which never visits the host -- against the two-pass path (already held to the reference's goldens) and
    against the oracle;
  * the speculative next E-step across a moving schedule (_predict_anneal) against the loop that never speculates.
Tolerances as tests/test_bsc_gpu.py: float64 kernels, statistics 1e-9, parameters 1e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    return torch.device("cuda", 0)


class _An(dict):
    crit_params = []

    def __missing__(self, k):
        return 0.0

    def as_dict(self):
        return dict(self)


def _recorded(m):
    names = []
    orig = m._call
    m._call = lambda label, name, *a, _o=orig, _n=names: (_n.append(name), _o(label, name, *a))[1]
    return names


@pytest.mark.parametrize("D,H,Hp,gamma,N,T,ncut,sigma0", [
    (128, 256, 8, 4, 40000, 1.0, 0.6, 1.05),      # whole round of 128-row tiles + TAIL launch (on a 256-CU chip)
    (64, 200, 8, 3, 5000, 1.5, 0.3, 1.05),        # TAIL only, H not a multiple of 32, gamma = 3
    (512, 256, 8, 4, 6000, 50.0, 0.9, 1.05),      # hot: lists overflow, the dense product runs behind its gate
    (1024, 256, 8, 4, 3000, 1.0, 0.5, 0.3),       # every log-evidence below log(2^-1075): the reference keeps ALL (bsc_et.py:253)
    (256, 256, 8, 4, 33000, 1.0, 1.0, 1.05),      # Ncut_factor = 1: the plateau of the reference's schedule
    (128, 256, 6, 3, 40000, 1.0, 0.6, 1.05),      # H' = 6, 5, 7 on the 16-wavefront kernel (round 6: the layout of eight
    (64, 192, 5, 4, 36000, 1.3, 0.4, 1.05),       # candidate positions, the state set of H')
    (96, 200, 7, 4, 5000, 1.0, 0.8, 1.05),
])
def test_deferred_statistics_match_two_pass_and_oracle(dev, D, H, Hp, gamma, N, T, ncut, sigma0):
    from oracle import bsc_oracle as O
    from prosper_amd import _lib
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    rng = np.random.RandomState(N + H + D)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 3.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": sigma0}
    out = {}
    for defer in (True, False):
        m = BSC_ET(D, H, Hp, gamma)
        m.defer_stats = defer
        names = _recorded(m)
        h = dlog.set_handler(("L", "N_use"), StoreInMemory)
        try:
            new = m.step(_An(T=T, Ncut_factor=ncut), dict(params), {"y": y})
        finally:
            dlog.remove_handler(h)
        out[defer] = (new, m._ws["stats"].cpu().numpy().copy(), names, float(h.tables["L"][0]), int(h.tables["N_use"][0]))
    a, b = out[True], out[False]
    assert "pm_bsc_defer_apply_f64" in a[2] and "pm_bsc_estep_fused8_defer_f64" in a[2]
    assert not any(n.startswith("pm_bsc_mstep_rows") for n in a[2])
    assert "pm_bsc_defer_apply_f64" not in b[2] and "pm_bsc_mstep_rows16_nz_f64" in b[2]
    o_sc = _lib.load().pm_bsc_stats_offset_scalars(H, D)
    sa, sb = a[1].copy(), b[1].copy()
    if sigma0 < 1.0:
        assert a[4] == N                      # below the reference's underflow point nothing is cut
    else:
        assert a[4] < N
    assert a[4] == b[4]
    # (overflow counters differ by design: the fused pass drops terms below exp(-37) of the row's largest, the M-step's own
    # pass below exp(-50) -- both under every statistic's rounding -- and the deferred pass counts dropped datapoints too)
    assert T < 10 or (sa[o_sc + 3] > 0 and sb[o_sc + 3] > 0)
    sa[o_sc + 3] = sb[o_sc + 3] = 0.0
    # (the diagonal of the second moments: the whole-shard pass and the apply kernel leave it in qdiag in full and a zero
    # diagonal in the Wq block, the M-step's own pass splits it into the multi-cause part there and the singletons' in qdiag)
    lib = _lib.load()
    o_wq, o_qd, o_mus = lib.pm_bsc_stats_offset_wq(H, D), lib.pm_bsc_stats_offset_qdiag(H, D), lib.pm_bsc_stats_offset_mus(H, D)
    for st_ in (sa, sb):
        wq = st_[o_wq:o_qd].reshape(H, H)
        st_[o_qd:o_mus] += np.diag(wq)
        wq[np.arange(H), np.arange(H)] = 0.0
    np.testing.assert_allclose(sa, sb, rtol=1e-9, atol=1e-11 * np.abs(sb).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-11)
    for k in ("W", "pi", "sigma"):
        np.testing.assert_allclose(a[0][k], b[0][k], rtol=1e-8, atol=1e-10)
    # ... and the oracle's step on the same inputs
    ref, rlog = O.em_step(O.Anneal(T=T, Ncut_factor=ncut, anneal_prior=False), O.make_model(D, H, Hp, gamma), dict(params), y,
                          stats_fn=O.m_step_stats_vec, vec=True)
    assert a[4] == rlog["N_use"]
    if np.isfinite(rlog["L"]):       # (the reference's un-stabilised evidence sums underflow in the sigma0 = 0.3 case: L = -inf there)
        np.testing.assert_allclose(a[3], rlog["L"], rtol=1e-10)
    tol = max(1e-8, 20 * np.linalg.cond(rlog["stats"]["Wq"]) * np.finfo(float).eps)
    assert tol < 1e-4
    np.testing.assert_allclose(a[0]["W"], ref["W"], rtol=10 * tol, atol=tol * np.abs(ref["W"]).max())
    np.testing.assert_allclose(a[0]["pi"], ref["pi"], rtol=1e-9)
    np.testing.assert_allclose(a[0]["sigma"], ref["sigma"], rtol=1e-9)


def test_deferred_records_with_a_foreign_m_step_call(dev):
    """A pass launched for a truncation step whose M_step is then called WITHOUT truncation (a foreign caller): the records
    are added with the cut at -inf; and an M_step with other scalars than the pass was launched with drops them."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 64, 160, 8, 3, 3000
    rng = np.random.RandomState(5)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 3.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.05}
    ref = BSC_ET(D, H, Hp, gamma).step(_An(T=1.0, Ncut_factor=0.0), dict(params), {"y": y})
    for other_sigma in (False, True):
        m = BSC_ET(D, H, Hp, gamma)
        p = dict(params)
        m._in_step = True
        try:
            d = m.select_Hprimes(p, {"y": y})
            ss = m.E_step(_An(T=1.0, Ncut_factor=0.5), p, d)          # launches the deferred form
            assert ss["logpj"].mstats["nz"].get("defer") is not None
            names = _recorded(m)
            if other_sigma:
                # (the log-joints belong to sigma = 1.05; an M-step at 1.2 weighs them as the reference would: its own pass)
                new = m.M_step(_An(T=1.0, Ncut_factor=0.5), dict(p, sigma=1.2), ss, d)
                assert "pm_bsc_defer_apply_f64" not in names and "pm_bsc_mstep_rows16_nz_f64" in names
            else:
                new = m.M_step(_An(T=1.0, Ncut_factor=0.0), p, ss, d)
                assert "pm_bsc_defer_apply_f64" in names
        finally:
            m._in_step = False
        if not other_sigma:
            for k in ("W", "pi", "sigma"):
                np.testing.assert_allclose(new[k], ref[k], rtol=1e-8, atol=1e-10)
        else:
            assert np.isfinite(new["W"]).all()


@pytest.mark.parametrize("D,H,Hp,gamma,N,steps", [(64, 160, 8, 3, 6000, 16), (256, 256, 8, 4, 34000, 12), (25, 10, 5, 3, 500, 12)])
def test_em_run_on_the_reference_schedule_speculates_across_the_ramp(dev, D, H, Hp, gamma, N, steps):
    """EM.run over LinearAnnealing with the reference's T and Ncut_factor ramps: the M-step launches the next E-step with
    the NEXT annealing point (a pure function of the position, annealing.py:90-107) and E_step adopts it on all but the
    first two steps; same N_use, free energies and parameters as the loop that never speculates and never defers."""
    from prosper_amd.em import EM
    from prosper_amd.em.annealing import LinearAnnealing
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    rng = np.random.RandomState(N + H)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 2.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 2.0 / H, "sigma": 1.05}
    out = {}
    for fast in (True, False):
        m = BSC_ET(D, H, Hp, gamma)
        m.speculate = True
        m.speculate_estep = fast
        m.defer_stats = fast
        an = LinearAnnealing(steps)
        an['T'] = [(0, 2.), (.7, 1.)]
        an['Ncut_factor'] = [(0, 0.), (2. / 3, 1.)]
        an['anneal_prior'] = False
        h = dlog.set_handler(("L", "N_use", "W", "pi", "sigma"), StoreInMemory)
        try:
            em = EM(model=m, anneal=an, data={"y": y}, lparams=dict(params))
            em.run()
        finally:
            dlog.remove_handler(h)
        out[fast] = (em.lparams, np.array(h.tables["L"]), np.array(h.tables["N_use"]), m.spec_hits,
                     np.array(h.tables["sigma"]), np.array(h.tables["pi"]))
    a, b = out[True], out[False]
    assert b[3] == 0
    if BSC_ET(D, H, Hp, gamma)._fused():
        # (no history at step 0, the prediction is trusted from step 1 on; a rejected warm start of the inverse voids the
        # pass launched from its unrefined solution -- frequent on the 500-datapoint toy, rare at size)
        assert a[3] >= (steps - 3 if N > 1000 else steps // 2), a[3]
    assert len(a[2]) == steps
    np.testing.assert_array_equal(a[2], b[2])
    assert a[2][0] == N and a[2][-1] < N
    np.testing.assert_allclose(a[1], b[1], rtol=1e-10)
    np.testing.assert_allclose(a[4], b[4], rtol=1e-9)
    np.testing.assert_allclose(a[5], b[5], rtol=1e-9)
    np.testing.assert_allclose(a[0]["W"], b[0]["W"], rtol=1e-6, atol=1e-8)


def test_a_schedule_that_is_not_advanced_falls_back_to_the_flat_predictor(dev):
    """model.step called again and again at the SAME position of a LinearAnnealing on its ramp: the look-ahead predictor is
    wrong once (a dropped pass), then the flat one takes over."""
    from prosper_amd.em.annealing import LinearAnnealing
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 64, 160, 8, 3, 3000
    rng = np.random.RandomState(11)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 3.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    p = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.05}
    an = LinearAnnealing(20)
    an['T'] = [(0, 2.), (.7, 1.)]
    an['Ncut_factor'] = [(0, 0.), (2. / 3, 1.)]
    an.next(0.)
    an.next(0.)
    m = BSC_ET(D, H, Hp, gamma)
    m.speculate = m.speculate_estep = True
    for _ in range(6):
        p = m.step(an, p, {"y": y})
    assert m.spec_hits >= 3, m.spec_hits
    assert np.isfinite(p["W"]).all()


# ------------------------------------------------------------------------- MCA / MMCA
@pytest.mark.parametrize("cls,D,H,Hp,gamma,N,T,ncut", [
    ("MCA", 256, 128, 8, 3, 3000, 1.0, 0.6),       # config-5 dimensions: 64-column slices of the scatter kernel
    ("MCA", 256, 128, 8, 3, 3000, 1.37, 1.0),      # a temperature of the ramp: rho = 3.70..., the uniform-exponent power
    ("MCA", 100, 200, 5, 4, 700, 1.6, 0.3),        # H > 128: 32-column slices; D not a multiple of 64
    ("MCA", 40, 300, 3, 2, 400, 2.0, 0.5),         # H > 256: 16-column slices
    ("MMCA", 256, 128, 8, 3, 2000, 1.0, 0.6),      # signed W
    ("MMCA", 64, 40, 6, 3, 900, 1.5, 1.0),
])
def test_mca_deferred_statistics_match_two_pass_and_oracle(dev, cls, D, H, Hp, gamma, N, T, ncut):
    """MCA_ET / MMCA_ET.step on a data-truncation step: the fused pass leaves the Aid blocks as records
    (pm_mca_estep_mstats_defer_f64), pm_mca_defer_apply_f64 adds the ones above the device-resident cut -- same statistics
    and parameters as the M-step's own pass (pm_mca_mstep_rows_f64 behind a host-side cut), and as the oracle."""
    from prosper_amd.em.camodels.mca_et import MCA_ET
    from prosper_amd.em.camodels.mmca_et import MMCA_ET
    rng = np.random.RandomState(D + H + N)
    if cls == "MCA":
        from oracle import mca_oracle as M
        W_gt = np.abs(rng.normal(size=(D, H))) * 2.0 + 0.1
        y, _ = M.generate_mca_data(W_gt, 2.0 / H, 1.0, N, rng)
        params = {"W": W_gt * (1 + 0.2 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.4 / H, "sigma": 1.1}
        make = lambda: MCA_ET(D, H, Hp, gamma)
    else:
        W_gt = rng.normal(size=(D, H)) * 2.0
        s = rng.random_sample((N, H)) < 2.0 / H
        amax = np.abs(np.where(s[:, None, :], W_gt[None], 0.0))
        idx = amax.argmax(axis=2)
        y = np.take_along_axis(np.where(s[:, None, :], W_gt[None], 0.0), idx[:, :, None], axis=2)[:, :, 0] + rng.normal(size=(N, D))
        params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 2.4 / H, "sigma": 1.1}
        make = lambda: MMCA_ET(D, H, Hp, gamma)
    out = {}
    for defer in (True, False):
        m = make()
        m.defer_stats = defer
        names = _recorded(m)
        new = m.step(_An(T=T, Ncut_factor=ncut), dict(params), {"y": y})
        out[defer] = (new, names)
    a, b = out[True], out[False]
    assert "pm_mca_defer_apply_f64" in a[1] and "pm_mca_estep_mstats_defer_f64" in a[1] and "pm_mca_mstep_rows_f64" not in a[1]
    assert "pm_mca_defer_apply_f64" not in b[1] and "pm_mca_mstep_rows_f64" in b[1]
    for k in ("W", "pi", "sigma", "Q"):
        np.testing.assert_allclose(a[0][k], b[0][k], rtol=1e-9, atol=1e-11)
    if cls == "MCA":
        ref, log = M.em_step(M.Anneal(T=T, Ncut_factor=ncut), M.make_model(D, H, Hp, gamma), dict(params), y, vec=True)
        np.testing.assert_allclose(a[0]["W"], ref["W"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(a[0]["pi"], ref["pi"], rtol=1e-9)
        np.testing.assert_allclose(a[0]["sigma"], ref["sigma"], rtol=1e-9)
        if np.isfinite(ref["Q"]):
            np.testing.assert_allclose(a[0]["Q"], ref["Q"], rtol=1e-10)


def test_mca_em_run_on_the_reference_schedule(dev):
    """EM.run over the reference's schedule at config-5 dimensions: the default path (deferred statistics, power tables built
    for the NEXT step's rho behind the download) against the two-pass path."""
    from oracle import mca_oracle as M
    from prosper_amd.em import EM
    from prosper_amd.em.annealing import LinearAnnealing
    from prosper_amd.em.camodels.mca_et import MCA_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    D, H, Hp, gamma, N, steps = 256, 128, 8, 3, 2500, 12
    rng = np.random.RandomState(17)
    W_gt = np.abs(rng.normal(size=(D, H))) * 2.0 + 0.1
    y, _ = M.generate_mca_data(W_gt, 2.0 / H, 1.0, N, rng)
    params = {"W": W_gt * (1 + 0.2 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.4 / H, "sigma": 1.1}
    out = {}
    for fast in (True, False):
        m = MCA_ET(D, H, Hp, gamma)
        m.defer_stats = fast
        names = _recorded(m)
        an = LinearAnnealing(steps)
        an['T'] = [(0, 2.), (.7, 1.)]
        an['Ncut_factor'] = [(0, 0.), (2. / 3, 1.)]
        h = dlog.set_handler(("N_use", "Q", "sigma"), StoreInMemory)
        try:
            em = EM(model=m, anneal=an, data={"y": y}, lparams=dict(params))
            em.run()
        finally:
            dlog.remove_handler(h)
        out[fast] = (em.lparams, np.array(h.tables["N_use"]), np.array(h.tables["Q"]), np.array(h.tables["sigma"]), names)
    a, b = out[True], out[False]
    # the tables of steps 2 .. come from the M-step before them (pm_mca_tables_f64 once per step, no second build)
    assert a[4].count("pm_mca_tables_f64") <= steps + 2, a[4].count("pm_mca_tables_f64")
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-9)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    np.testing.assert_allclose(a[0]["W"], b[0]["W"], rtol=1e-6, atol=1e-8)
