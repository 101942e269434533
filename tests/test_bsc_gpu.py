"""Parity tests proper (need an MI355X): the HIP path, called through the C ABI, against
  * golden vectors minted from the reference (tests/golden/*.npz),
  * the oracle (oracle/bsc_oracle.py) on seeded inputs at sizes it finishes in seconds,
  * size-independent properties at BASELINE config-2 dimensions.
Tolerances: the kernels compute in float64 like the reference.  BASELINE.json asks for 1e-4
relative on W/pi/sigma/L; these tests hold the f64 path to 1e-8 (1e-6 after 20 EM steps)."""
import ctypes

import numpy as np
import pytest

from conftest import golden, bsc_step_cases, rank_deficient

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

RTOL_STEP = 1e-8
BASELINE_RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    return torch.device("cuda", 0)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------- dense kernels
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (200, 256, 1024), (333, 10, 25), (1, 1, 1), (130, 129, 18),
                                   (5000, 256, 1024), (64, 100, 48), (257, 12, 30),
                                   (35768, 256, 256),    # one whole round of tiles + a ragged one: the fused launch
                                   (33000, 200, 512),    # ... with edge tiles in both directions
                                   (9000, 400, 784), (700, 136, 64), (300, 72, 40)])   # column tiles with 1, 8 or 9 of their 16-column blocks inside N
def test_gemm_nt_matches_numpy(dev, M, N, K):
    from prosper_amd import _lib
    rng = np.random.RandomState(M + 7 * N + K)
    A, B = rng.normal(size=(M, K)), rng.normal(size=(N, K))
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    c = torch.full((M, N), float("nan"), dtype=torch.float64, device=dev)
    _lib.call("pm_gemm_nt_f64", _p(a), K, _p(b), K, _p(c), N, M, N, K, _stream())
    ref = A @ B.T
    np.testing.assert_allclose(c.cpu().numpy(), ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("M,N,K", [(256, 256, 1024), (128, 128, 256), (100, 37, 25), (1, 1, 1), (130, 129, 18), (17, 512, 1030),
                                   (300, 300, 64), (512, 256, 128)])
def test_gemm_nt_small_matches_numpy_and_is_deterministic(dev, M, N, K):
    """pm_gemm_nt_small_f64 (Gram matrices, H x H x D solve products): against NumPy; bit-identical from launch to launch
    (fixed summation order, no atomics -- what keeps ranks with the same W on the same bits); A == B gives a Gram matrix
    that is symmetric bit for bit."""
    from prosper_amd import _lib
    rng = np.random.RandomState(M + 7 * N + K)
    A, B = rng.normal(size=(M, K)), rng.normal(size=(N, K))
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    outs = []
    for _ in range(3):
        c = torch.full((M, N), float("nan"), dtype=torch.float64, device=dev)
        _lib.call("pm_gemm_nt_small_f64", _p(a), K, _p(b), K, _p(c), N, M, N, K, _stream())
        outs.append(c.cpu().numpy())
    ref = A @ B.T
    np.testing.assert_allclose(outs[0], ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    if M == N:
        g = torch.full((M, M), float("nan"), dtype=torch.float64, device=dev)
        _lib.call("pm_gemm_nt_small_f64", _p(a), K, _p(a), K, _p(g), M, M, M, K, _stream())
        G = g.cpu().numpy()
        assert np.array_equal(G, G.T)
        np.testing.assert_allclose(G, A @ A.T, rtol=1e-12, atol=1e-12 * np.abs(A @ A.T).max())


@pytest.mark.parametrize("M,N,K", [(100000, 128, 256), (70000, 256, 64), (65536 + 64 * 150 + 7, 128, 256)])
def test_gemm_nt_ragged_last_round_in_64_row_tiles(dev, M, N, K):
    """pm_gemm_nt_f64 on tall products whose last round of 128-row tiles would fill a quarter to 70 % of the resident
    slots and whose K is too short to split: that round runs in 64-row tiles (gemm_nt_f64_dma_kernel<false, 64>).  Rows
    across the seam between the two launches and the ragged end against torch."""
    from prosper_amd import _lib
    g = torch.Generator(device=dev).manual_seed(M + N)
    a = torch.randn(M, K, generator=g, device=dev, dtype=torch.float64)
    b = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    c = torch.full((M, N), float("nan"), dtype=torch.float64, device=dev)
    _lib.call("pm_gemm_nt_f64", _p(a), K, _p(b), K, _p(c), N, M, N, K, _stream())
    ref = a @ b.t()
    err = float((c - ref).abs().max())
    assert err <= 1e-11 * float(ref.abs().max()), err


def test_gemm_nt_layout_asymmetric(dev):
    """A = I against an asymmetric B catches a swapped row/column map of the MFMA tile."""
    from prosper_amd import _lib
    M = N = K = 192
    B = np.arange(N * K, dtype=np.float64).reshape(N, K)
    a, b = torch.eye(M, dtype=torch.float64, device=dev), torch.from_numpy(B).to(dev)
    c = torch.zeros((M, N), dtype=torch.float64, device=dev)
    _lib.call("pm_gemm_nt_f64", _p(a), K, _p(b), K, _p(c), N, M, N, K, _stream())
    assert np.array_equal(c.cpu().numpy(), B.T)


@pytest.mark.parametrize("M,N,K", [(256, 1024, 4096), (10, 25, 333), (128, 128, 16), (100, 48, 1000), (33, 7, 5),
                                   (256, 1024, 50001)])
def test_gemm_tn_acc_matches_numpy(dev, M, N, K):
    from prosper_amd import _lib
    rng = np.random.RandomState(M + N + K)
    A, B, C0 = rng.normal(size=(K, M)), rng.normal(size=(K, N)), rng.normal(size=(M, N))
    a, b, c = (torch.from_numpy(x).to(dev) for x in (A, B, C0))
    _lib.call("pm_gemm_tn_acc_f64", _p(a), M, _p(b), N, _p(c), N, M, N, K, _stream())
    ref = C0 + A.T @ B
    np.testing.assert_allclose(c.cpu().numpy(), ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max())


def test_row_sqnorm(dev):
    from prosper_amd import _lib
    Y = np.random.RandomState(0).normal(size=(1001, 77))
    y = torch.from_numpy(Y).to(dev)
    out = torch.empty(1001, dtype=torch.float64, device=dev)
    _lib.call("pm_row_sqnorm_f64", _p(y), 77, 1001, 77, _p(out), _stream())
    np.testing.assert_allclose(out.cpu().numpy(), (Y * Y).sum(1), rtol=1e-13)


@pytest.mark.parametrize("n", [1, 7, 32, 33, 100, 256])
def test_spd_inverse(dev, n):
    """pm_spd_inverse_f64 (the M-step's W solve, bsc_et.py:380) against LAPACK."""
    from prosper_amd import _lib
    rs = np.random.RandomState(n)
    B = rs.normal(size=(n, 3 * n + 5))
    A = B @ B.T
    dadd = rs.uniform(0.1, 1.0, size=n)
    U = np.triu(A) + np.tril(rs.normal(size=(n, n)), -1)         # the strict lower triangle must be ignored
    u = torch.from_numpy(U).to(dev)
    da = torch.from_numpy(dadd).to(dev)
    full = torch.zeros((n, n), dtype=torch.float64, device=dev)
    inv = torch.zeros((n, n), dtype=torch.float64, device=dev)
    piv = torch.zeros(2, dtype=torch.float64, device=dev)
    _lib.call("pm_spd_inverse_f64", _p(u), n, _p(da), n, _p(full), _p(inv), n, _p(piv), _stream())
    Af = A + np.diag(dadd)
    np.testing.assert_array_equal(full.cpu().numpy(), Af)
    ref = np.linalg.inv(Af)
    got = inv.cpu().numpy()
    np.testing.assert_array_equal(got, got.T)
    cond = np.linalg.cond(Af)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-14 * cond * np.abs(ref).max())
    d2 = np.diag(np.linalg.cholesky(Af)) ** 2                      # sweep pivots = squared Cholesky diagonal
    np.testing.assert_allclose(piv.cpu().numpy(), [d2.min(), d2.max()], rtol=1e-9)


def test_spd_inverse_batch(dev):
    """pm_spd_inverse_batch_f64: three matrices, one workgroup each, strided in and out (GSC's M-step uses two)."""
    from prosper_amd import _lib
    n, batch, pad = 40, 3, 7
    rs = np.random.RandomState(5)
    mats, dadd = [], rs.uniform(0.1, 1.0, size=(batch, n))
    for b in range(batch):
        B = rs.normal(size=(n, 2 * n + b))
        mats.append(B @ B.T)
    inp = torch.zeros((batch, n * n + pad), dtype=torch.float64, device=dev)
    for b in range(batch):
        inp[b, :n * n] = torch.from_numpy(np.triu(mats[b]) + np.tril(rs.normal(size=(n, n)), -1)).to(dev).reshape(-1)
    da = torch.from_numpy(dadd).to(dev).contiguous()
    full = torch.zeros((batch, n * n + pad), dtype=torch.float64, device=dev)
    inv = torch.zeros((batch, n * n + pad), dtype=torch.float64, device=dev)
    piv = torch.zeros(2 * batch, dtype=torch.float64, device=dev)
    _lib.call("pm_spd_inverse_batch_f64", _p(inp), n, n * n + pad, _p(da), n, _p(full), _p(inv), n, n * n + pad, _p(piv),
              batch, _stream())
    for b in range(batch):
        Af = mats[b] + np.diag(dadd[b])
        np.testing.assert_array_equal(full[b, :n * n].view(n, n).cpu().numpy(), Af)
        ref, got = np.linalg.inv(Af), inv[b, :n * n].view(n, n).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-14 * np.linalg.cond(Af) * np.abs(ref).max())
        d2 = np.diag(np.linalg.cholesky(Af)) ** 2
        np.testing.assert_allclose(piv[2 * b:2 * b + 2].cpu().numpy(), [d2.min(), d2.max()], rtol=1e-9)
    assert (inv[:, n * n:] == 0).all() and (full[:, n * n:] == 0).all()          # nothing written between matrices
    with pytest.raises(_lib.HipError):                                            # overlapping strides
        _lib.call("pm_spd_inverse_batch_f64", _p(inp), n, n, None, n, None, _p(inv), n, n * n, _p(piv), 2, _stream())


@pytest.mark.parametrize("n", [1, 7, 33, 100, 256])
@pytest.mark.parametrize("rel", [0.0, 1e-3, 2e-2, 0.5])
def test_spd_inverse_warm(dev, n, rel):
    """pm_spd_inverse_warm_f64: the inverse of a nearby matrix (relative distance ``rel``) refined by Newton-Schulz
    steps on the matrix cores; the device itself falls back to the exact sweep when the start residual is too large
    (rel = 0.5) -- either way the result is an inverse to ~1e-8 relative or better (the caller's solve refines once more against
    A), symmetric, and the pivots say "usable"."""
    from prosper_amd import _lib
    rs = np.random.RandomState(n)
    B = rs.normal(size=(n, 3 * n + 5))
    A = B @ B.T + np.diag(rs.uniform(0.1, 1.0, size=n))
    B0 = B + rel * rs.normal(size=B.shape)
    A0 = B0 @ B0.T + np.diag(np.diag(A) - np.diag(B @ B.T))
    dadd = rs.uniform(0.1, 1.0, size=n)
    Af = A + np.diag(dadd)
    prev = torch.from_numpy(np.linalg.inv(A0 + np.diag(dadd))).to(dev)
    prev = 0.5 * (prev + prev.t())
    u = torch.from_numpy(np.triu(A) + np.tril(rs.normal(size=(n, n)), -1)).to(dev)
    da = torch.from_numpy(dadd).to(dev)
    work = torch.zeros(int(_lib.load().pm_spd_inverse_warm_work_len(n)), dtype=torch.float64, device=dev)
    full = torch.zeros((n, n), dtype=torch.float64, device=dev)
    inv = torch.zeros((n, n), dtype=torch.float64, device=dev)
    piv = torch.zeros(3, dtype=torch.float64, device=dev)
    _lib.call("pm_spd_inverse_warm_f64", _p(u), n, _p(da), n, _p(prev), n, _p(work), _p(full), _p(inv), n, _p(piv), _stream())
    np.testing.assert_array_equal(full.cpu().numpy(), Af)
    got, pv = inv.cpu().numpy(), piv.cpu().numpy()
    np.testing.assert_array_equal(got, got.T)
    R0 = np.eye(n) - Af @ prev.cpu().numpy()
    r0 = np.linalg.norm(R0)
    # round 6: the guard reads the LAST residual the four steps form, R_3 = R_0^8 (Frobenius norm below 1e-8), not the start
    with np.errstate(all="ignore"):
        r3 = np.linalg.norm(np.linalg.matrix_power(R0, 8))
    if r3 < 0.3e-8:
        # refined, sweep skipped: conditioning from the inverse's diagonal -- 1 / X_ii is row i's pivot if eliminated
        # last (a lower bound of the sweep's smallest pivot), the largest pivot is at most the largest diagonal entry
        d2 = np.diag(np.linalg.cholesky(Af)) ** 2
        np.testing.assert_allclose(pv[:2], [1.0 / np.diag(np.linalg.inv(Af)).max(), np.diag(Af).max()], rtol=1e-7)
        assert pv[0] <= d2.min() * (1 + 1e-7) and pv[1] >= d2.max() * (1 - 1e-12)
        assert pv[2] == (1.0 if r0 < 0.999 else 2.0) or 0.999 <= r0 <= 1.001      # the device's verdict: accepted
        assert np.linalg.norm(np.eye(n) - Af @ got) < max(2.0 * r3 ** 2, 1e-11 * np.linalg.cond(Af))
    elif not (r3 < 3e-8):
        d2 = np.diag(np.linalg.cholesky(Af)) ** 2                     # the exact sweep ran
        np.testing.assert_allclose(pv[:2], [d2.min(), d2.max()], rtol=1e-9)
        assert pv[2] == 0.0                                           # ... rejected: the caller refines its solve
        ref = np.linalg.inv(Af)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-14 * np.linalg.cond(Af) * np.abs(ref).max())
    assert rel != 0.5 or not (r3 < 3e-8) or n == 1
    assert rel > 1e-3 or r3 < 0.3e-8


@pytest.mark.parametrize("n", [7, 100, 256])
@pytest.mark.parametrize("kind", ["scaled", "subset", "far"])
def test_spd_inverse_warm_long(dev, n, kind):
    """pm_spd_inverse_warm_long_f64: a start that is NOT close -- the inverse of the matrix times 2.5 / 0.4 (a jump of the
    kept count of a data-truncation step), of the second moments of a subset of the data, of an unrelated matrix -- is
    scaled by 1 / ||A X0||_inf and refined by eight + one Newton-Schulz steps; accepted (flag 1 or 2, inverse exact to
    rounding) while ||A X0||_inf / lambda_min(A X0) stays below ~14, else the sweep runs (flag 0) -- right either way."""
    from prosper_amd import _lib
    rs = np.random.RandomState(n + len(kind))
    S = (rs.random_sample((40 * n, n)) < 0.1).astype(np.float64)
    A = S.T @ S
    dadd = rs.uniform(0.1, 1.0, size=n)
    Af = A + np.diag(dadd)
    if kind == "scaled":
        X0 = np.linalg.inv(Af) * (2.5 if n != 100 else 0.4)
    elif kind == "subset":
        keep = rs.random_sample(S.shape[0]) < 0.6
        X0 = np.linalg.inv(S[keep].T @ S[keep] + np.diag(dadd))
    else:
        B = rs.normal(size=(n, 3 * n))
        X0 = np.linalg.inv(B @ B.T + np.eye(n))
    X0 = 0.5 * (X0 + X0.T)
    T = Af @ X0
    x = np.abs(T).sum(axis=1).max() / np.sort(np.linalg.eigvals(T).real)[0]
    prev = torch.from_numpy(X0).to(dev)
    u = torch.from_numpy(np.triu(A) + np.tril(rs.normal(size=(n, n)), -1)).to(dev)
    da = torch.from_numpy(dadd).to(dev)
    work = torch.zeros(int(_lib.load().pm_spd_inverse_warm_work_len(n)), dtype=torch.float64, device=dev)
    full = torch.zeros((n, n), dtype=torch.float64, device=dev)
    inv = torch.zeros((n, n), dtype=torch.float64, device=dev)
    piv = torch.zeros(3, dtype=torch.float64, device=dev)
    _lib.call("pm_spd_inverse_warm_long_f64", _p(u), n, _p(da), n, _p(prev), n, _p(work), _p(full), _p(inv), n, _p(piv), _stream())
    np.testing.assert_array_equal(full.cpu().numpy(), Af)
    got, pv = inv.cpu().numpy(), piv.cpu().numpy()
    np.testing.assert_array_equal(got, got.T)
    ref = np.linalg.inv(Af)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * np.linalg.cond(Af) * np.abs(ref).max())
    if x < 10:
        assert pv[2] in (1.0, 2.0), (x, pv)
        assert kind != "scaled" or pv[2] == 2.0            # ||I - 2.5 I||_F >= 1: a far start
    if x > 40:
        assert pv[2] == 0.0, (x, pv)
    assert kind == "far" or x < 10


def test_spd_inverse_warm_batch(dev):
    """pm_spd_inverse_warm_batch_f64: two matrices per launch (GSC's M-step), one refined from a close start, the other
    recomputed by the sweep because its start is far off."""
    from prosper_amd import _lib
    n, batch, pad = 40, 2, 6
    rs = np.random.RandomState(9)
    mats, starts = [], []
    for b, rel in enumerate((1e-3, 0.7)):
        B = rs.normal(size=(n, 2 * n + b))
        B0 = B + rel * rs.normal(size=B.shape)
        mats.append(B @ B.T)
        starts.append(np.linalg.inv(B0 @ B0.T + 0.3 * np.eye(n)))
    dadd = np.full((batch, n), 0.3)
    inp = torch.zeros((batch, n * n + pad), dtype=torch.float64, device=dev)
    for b in range(batch):
        inp[b, :n * n] = torch.from_numpy(np.triu(mats[b]) + np.tril(rs.normal(size=(n, n)), -1)).to(dev).reshape(-1)
    prev = torch.from_numpy(np.stack([0.5 * (x + x.T) for x in starts])).to(dev).contiguous()
    da = torch.from_numpy(dadd).to(dev).contiguous()
    work = torch.zeros(batch * int(_lib.load().pm_spd_inverse_warm_work_len(n)), dtype=torch.float64, device=dev)
    inv = torch.zeros((batch, n * n), dtype=torch.float64, device=dev)
    piv = torch.zeros(2 * batch, dtype=torch.float64, device=dev)
    _lib.call("pm_spd_inverse_warm_batch_f64", _p(inp), n, n * n + pad, _p(da), n, _p(prev), n * n, _p(work), _p(inv),
              n * n, _p(piv), batch, _stream())
    pv = piv.cpu().numpy()
    for b in range(batch):
        Af = mats[b] + np.diag(dadd[b])
        got = inv[b].view(n, n).cpu().numpy()
        np.testing.assert_array_equal(got, got.T)
        assert np.linalg.norm(np.eye(n) - Af @ got) < 1e-9 * np.linalg.cond(Af)
    A0f = mats[0] + np.diag(dadd[0])                                    # refined: [min_i 1 / X_ii, max_i A_ii]
    np.testing.assert_allclose(pv[:2], [1.0 / np.diag(np.linalg.inv(A0f)).max(), np.diag(A0f).max()], rtol=1e-7)
    d2 = np.diag(np.linalg.cholesky(mats[1] + np.diag(dadd[1]))) ** 2  # swept
    np.testing.assert_allclose(pv[2:], [d2.min(), d2.max()], rtol=1e-9)


@pytest.mark.parametrize("n,asym", [(24, 1e-6), (128, 1e-6), (128, 1e-4), (256, 1e-5)])
def test_inverse_warm_batch_general(dev, n, asym):
    """pm_inverse_warm_batch_f64: [an SPD matrix given by its upper triangle + diag_add ; a GENERAL matrix given in full]
    -- GSC's (sum xpt_ss + eps I, sum xpt_szsz) from the second EM step on.  Cold start as the model does it (sweep of
    the upper-mirrored matrices, then the refinement): the general matrix comes back as the inverse of its TRANSPOSE,
    exact to rounding like LAPACK's general inverse, not symmetrised; a far-off start is reported as not accepted."""
    from prosper_amd import _lib
    rs = np.random.RandomState(n)
    B = rs.normal(size=(n, 3 * n))
    S0, S1 = B @ B.T / n, None
    C = rs.normal(size=(n, 3 * n))
    S1 = C @ C.T / n
    E = rs.normal(size=(n, n))
    A1 = S1 + asym * np.abs(S1).max() * (E - E.T)                   # non-symmetric, as sum xpt_szsz is
    dadd = np.concatenate([np.full(n, 1e-5), np.zeros(n)])
    mats = torch.from_numpy(np.stack([np.triu(S0), A1])).to(dev)
    da = torch.from_numpy(dadd).to(dev)
    inv = torch.zeros((2, n, n), dtype=torch.float64, device=dev)
    piv = torch.zeros(4, dtype=torch.float64, device=dev)
    acc = torch.full((2,), -1.0, dtype=torch.float64, device=dev)
    work = torch.zeros(2 * int(_lib.load().pm_spd_inverse_warm_work_len(n)), dtype=torch.float64, device=dev)
    _lib.call("pm_spd_inverse_batch_f64", _p(mats), n, n * n, _p(da), n, None, _p(inv), n, n * n, _p(piv), 2, _stream())
    prev = inv.clone()
    _lib.call("pm_inverse_warm_batch_f64", _p(mats), n, n * n, _p(da), n, _p(prev), n * n, _p(work), _p(inv), n * n, _p(piv),
              _p(acc), 2, 2, _stream())
    assert acc.cpu().tolist() == [1.0, 1.0]
    got = inv.cpu().numpy()
    ref0, ref1 = np.linalg.inv(S0 + 1e-5 * np.eye(n)), np.linalg.inv(A1)
    c0, c1 = np.linalg.cond(S0), np.linalg.cond(A1)
    np.testing.assert_allclose(got[0], ref0, rtol=0, atol=1e-14 * c0 * np.abs(ref0).max())
    np.testing.assert_allclose(got[1], ref1.T, rtol=0, atol=1e-14 * c1 * np.abs(ref1).max())
    assert np.abs(got[1] - got[1].T).max() > 0.1 * np.abs(ref1 - ref1.T).max()            # not symmetrised
    assert np.abs(A1 @ got[1].T - np.eye(n)).max() < 1e-13 * c1
    # warm: a nearby pair refined from these inverses
    mats2 = torch.from_numpy(np.stack([np.triu(S0 * 1.001), A1 * 0.999 + 1e-4 * asym * E])).to(dev)
    prev = inv.clone()
    _lib.call("pm_inverse_warm_batch_f64", _p(mats2), n, n * n, _p(da), n, _p(prev), n * n, _p(work), _p(inv), n * n, _p(piv),
              _p(acc), 2, 2, _stream())
    assert acc.cpu().tolist() == [1.0, 1.0]
    ref1b = np.linalg.inv(A1 * 0.999 + 1e-4 * asym * E)
    np.testing.assert_allclose(inv[1].cpu().numpy(), ref1b.T, rtol=0, atol=1e-14 * c1 * np.abs(ref1b).max())
    # a start that is far off: the general matrix is reported as NOT accepted (the caller inverts on the host)
    prev = (inv * 3.0).clone()
    _lib.call("pm_inverse_warm_batch_f64", _p(mats2), n, n * n, _p(da), n, _p(prev), n * n, _p(work), _p(inv), n * n, _p(piv),
              _p(acc), 2, 2, _stream())
    assert acc.cpu().tolist() == [0.0, 0.0]
    ref0b = np.linalg.inv(S0 * 1.001 + 1e-5 * np.eye(n))
    np.testing.assert_allclose(inv[0].cpu().numpy(), ref0b, rtol=0, atol=1e-13 * c0 * np.abs(ref0b).max())   # (the sweep's)


@pytest.mark.parametrize("n", [257, 300, 512, 700])
def test_spd_inverse_blocked(dev, n):
    """H x H systems beyond the one-workgroup inverse (n > 256): recursive 2 x 2 blocking with Schur complements on the
    library's own kernels (DeviceCAModel._spd_inverse_blocked; no rocSOLVER) -- inverse and pivots against numpy."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    rs = np.random.RandomState(n)
    B = rs.normal(size=(n, 2 * n))
    A = B @ B.T + np.diag(rs.uniform(0.5, 2.0, size=n))
    m = BSC_ET(8, 4, 2, 2)
    inv, pmin, pmax = m._spd_inverse_blocked(torch.from_numpy(A).to(dev))
    got = inv.cpu().numpy()
    ref = np.linalg.inv(A)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * np.linalg.cond(A) * np.abs(ref).max())
    np.testing.assert_array_equal(got, got.T)
    d2 = np.diag(np.linalg.cholesky(A)) ** 2           # the pivots of the unblocked elimination
    np.testing.assert_allclose([float(pmin), float(pmax)], [d2.min(), d2.max()], rtol=1e-8)


def test_spd_inverse_rejects_large(dev):
    from prosper_amd import _lib
    t = torch.zeros((300, 300), dtype=torch.float64, device=dev)
    with pytest.raises(_lib.HipError):
        _lib.call("pm_spd_inverse_f64", _p(t), 300, None, 300, None, _p(t), 300, None, _stream())


def test_col_moments_and_standard_init(dev):
    """pm_col_moments_f64 and the device standard_init (camodels/__init__.py:196-235) against the
    reference's golden init (same RNG stream, data passes on the device)."""
    from prosper_amd import _lib
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    rs = np.random.RandomState(3)
    Y = rs.normal(size=(1237, 300)) * 3 + rs.normal(size=300)
    y = torch.from_numpy(Y).to(dev)
    s1 = torch.zeros(300, dtype=torch.float64, device=dev)
    _lib.call("pm_col_moments_f64", _p(y), 300, 1237, 300, None, _p(s1), _stream())
    np.testing.assert_allclose(s1.cpu().numpy(), Y.sum(0), rtol=1e-12, atol=1e-10)
    c = torch.from_numpy(Y.mean(0)).to(dev)
    s2 = torch.zeros(300, dtype=torch.float64, device=dev)
    _lib.call("pm_col_moments_f64", _p(y), 300, 1237, 300, _p(c), _p(s2), _stream())
    np.testing.assert_allclose(s2.cpu().numpy(), ((Y - Y.mean(0)) ** 2).sum(0), rtol=1e-12)
    g = golden("bsc_init_c1.npz")
    m = BSC_ET(int(g["D"]), int(g["H"]), 5, 3)
    np.random.seed(int(g["seed_init"]))
    init = m.standard_init({"y": g["y"]})
    np.testing.assert_allclose(init["W"], g["W0"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(init["sigma"], g["sigma0"], rtol=1e-12)
    assert init["pi"] == float(g["pi0"])


# ------------------------------------------------------------------------- BSC vs golden
class _An(dict):
    crit_params = []

    def __missing__(self, k):
        return 0.0

    def as_dict(self):
        return dict(self)


PATHS = ["fused", "rows16", "wave64"]   # one-kernel E-step | scores GEMM + 16-lane row kernel | generic kernels


def _set_path(m, path):
    m.use_rows16 = path != "wave64"     # False: the generic one-wavefront-per-datapoint kernels
    m.use_fused = path == "fused"       # scores GEMM + select + E-step in one kernel (bsc_fused.hip)
    return m


def _device_stats(m):
    """(Wp (H,D), Wq (H,H)) assembled from the packed statistics buffer of the model's last M-step."""
    from prosper_amd import _lib
    lib = _lib.load()
    H, D = m.H, m.D
    st = m._ws["stats"].cpu().numpy()
    o_wq, o_qd, o_mus = lib.pm_bsc_stats_offset_wq(H, D), lib.pm_bsc_stats_offset_qdiag(H, D), lib.pm_bsc_stats_offset_mus(H, D)
    U = st[o_wq:o_qd].reshape(H, H)
    return st[:o_wq].reshape(H, D).copy(), np.triu(U) + np.triu(U, 1).T + np.diag(st[o_qd:o_mus])


def _run_step(g, to_learn=None, path="fused"):
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    to_learn = to_learn or [str(s) for s in g["to_learn"]]
    m = _set_path(BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]), to_learn=to_learn), path)
    an = _An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
    params = {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}
    if bool(g["has_mu"]):
        params["mu"] = g["mu"].copy()
    h = dlog.set_handler(("L", "N", "N_use"), StoreInMemory)
    try:
        data = m.select_Hprimes(params, {"y": g["y"]})
        ss = m.E_step(an, params, data)
        new = m.M_step(an, params, ss, data)
    finally:
        dlog.remove_handler(h)
    return m, params, data, ss, new, h.tables


@pytest.mark.parametrize("case", bsc_step_cases())
@pytest.mark.parametrize("path", PATHS)
def test_bsc_step_matches_reference_golden(dev, case, path):
    g = golden(case)
    m, params, data, ss, new, log = _run_step(g, path=path)
    assert m._state_tables()["fast"] == (path != "wave64") and m._fused() == (path == "fused")
    cand = np.asarray(data["candidates"])
    assert cand.dtype == np.int64 and np.array_equal(cand, g["candidates"])
    logpj = np.asarray(ss["logpj"])
    assert logpj.shape == g["logpj"].shape and logpj.dtype == np.float64
    np.testing.assert_allclose(logpj, g["logpj"], rtol=1e-10, atol=1e-9)
    assert "mu" in params                                  # E_step inserts mu (bsc_et.py:145-149)
    assert int(log["N"][0]) == int(g["N"]) and int(log["N_use"][0]) == int(g["N_use"])
    np.testing.assert_allclose(float(log["L"][0]), float(g["L"]), rtol=1e-11)
    scale = np.abs(g["W_new"]).max()
    if "Wq" in g:      # config-2 fixtures: the all-reduced statistics the reference handed to lstsq (bsc_et.py:373-380)
        Wp, Wq = _device_stats(m)
        np.testing.assert_allclose(Wq, g["Wq"], rtol=1e-9, atol=1e-12 * np.abs(g["Wq"]).max())
        np.testing.assert_allclose(Wp, g["Wp"], rtol=1e-9, atol=1e-12 * np.abs(g["Wp"]).max())
    if not rank_deficient(g):      # (else W_new is defined only up to lstsq's SVD cutoff: the statistics pin it)
        np.testing.assert_allclose(new["W"], g["W_new"], rtol=RTOL_STEP, atol=RTOL_STEP * scale)
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=RTOL_STEP)
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=RTOL_STEP)
    np.testing.assert_allclose(new["mu"], g["mu_new"], rtol=1e-7, atol=1e-9)
    assert new["W"].shape == (int(g["D"]), int(g["H"]))


@pytest.mark.parametrize("case", ["bsc_step_c2_plain.npz", "bsc_step_c2_fullrank.npz", "bsc_step_h256.npz", "bsc_step_c1_plain.npz"])
def test_install_parameters_entry_of_the_bench(dev, case):
    """bench.py's timed region drives `install_parameters` -> `select_Hprimes` -> `E_step` (W^T already on the device, as
    an M-step leaves it; bench.py:estep_pass): the same three calls against the reference's golden -- incl. a second
    install of a DIFFERENT W^T in between, whose Gram matrix / scores must not survive."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    g = golden(case)
    m = BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    an = _An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
    params = {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}
    if bool(g["has_mu"]):
        params["mu"] = g["mu"].copy()
    data = {"y": g["y"]}
    Wt_host = np.ascontiguousarray(g["W"].T)
    Wt_dev = torch.from_numpy(Wt_host).to(dev)
    other_host = np.ascontiguousarray(Wt_host[::-1] * 1.5)
    other_dev = torch.from_numpy(other_host).to(dev)
    for _ in range(2):
        m.install_parameters(data, other_dev, other_host)
        m.E_step(an, dict(params, W=np.ascontiguousarray(other_host.T)), m.select_Hprimes(dict(params, W=np.ascontiguousarray(other_host.T)), data))
        m.install_parameters(data, Wt_dev, Wt_host)
        d = m.select_Hprimes(params, data)
        ss = m.E_step(an, params, d)
        assert np.array_equal(np.asarray(d["candidates"]), g["candidates"])
        np.testing.assert_allclose(np.asarray(ss["logpj"]), g["logpj"], rtol=1e-10, atol=1e-9)


def test_m_step_accepts_foreign_numpy_inputs(dev):
    """candidates / logpj handed in as plain NumPy arrays (not our device handles)."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    g = golden("bsc_step_c1_anneal_cut.npz")
    m = BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    an = _An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=True)
    params = {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"]), "mu": g["mu"].copy()}
    new = m.M_step(an, params, {"logpj": g["logpj"]}, {"y": g["y"], "candidates": g["candidates"]})
    np.testing.assert_allclose(new["W"], g["W_new"], rtol=RTOL_STEP, atol=RTOL_STEP * np.abs(g["W_new"]).max())
    np.testing.assert_allclose(new["sigma"], g["sigma_new"], rtol=RTOL_STEP)
    ss = m.E_step(an, params, {"y": g["y"], "candidates": g["candidates"]})
    np.testing.assert_allclose(np.asarray(ss["logpj"]), g["logpj"], rtol=1e-10, atol=1e-9)


def test_to_learn_subset_keeps_parameters(dev):
    g = golden("bsc_step_c1_plain.npz")
    _, _, _, _, new, _ = _run_step(g, to_learn=["pi"])
    assert np.array_equal(new["W"], g["W"]) and new["sigma"] == float(g["sigma"])
    np.testing.assert_allclose(new["pi"], g["pi_new"], rtol=RTOL_STEP)


def test_trajectory_c1_matches_reference(dev):
    """BASELINE config 1 through the drop-in EM driver: 20 annealed steps on bars data."""
    from prosper_amd.em import EM
    from prosper_amd.em.annealing import LinearAnnealing
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    g = golden("bsc_traj_c1.npz")
    steps = int(g["steps"])
    model = BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    anneal = LinearAnnealing(steps)
    anneal["T"] = [(0, 2.), (.7, 1.)]
    anneal["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    anneal["anneal_prior"] = False
    h = dlog.set_handler("*", StoreInMemory)
    try:
        em = EM(model=model, anneal=anneal, data={"y": g["y"]},
                lparams={"W": g["W0"].copy(), "pi": float(g["pi0"]), "sigma": float(g["sigma0"])})
        em.run()
    finally:
        dlog.remove_handler(h)
    W = np.stack(h.tables["W"])
    np.testing.assert_allclose(np.array(h.tables["L"], dtype=float), g["L"], rtol=1e-8)
    assert np.array_equal(np.array(h.tables["N_use"], dtype=int), g["N_use"])
    np.testing.assert_allclose(W, g["W"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.array(h.tables["pi"], dtype=float), g["pi"], rtol=1e-7)
    np.testing.assert_allclose(np.array(h.tables["sigma"], dtype=float), g["sigma"], rtol=1e-7)
    np.testing.assert_allclose(em.lparams["W"], g["W"][-1], rtol=BASELINE_RTOL, atol=1e-6)
    # learned bars recover the ground truth up to permutation (the reference's de-facto integration test)
    from prosper_amd.utils.barstest import find_permutation
    _, mae = find_permutation(em.lparams["W"], g["W_gt"])
    assert mae < 0.5


def _bsc_problem(seed, D=96, H=32, N=3000):
    rng = np.random.RandomState(seed)
    W_gt = rng.normal(size=(D, H)) * 2
    s = rng.random_sample((N, H)) < 3.0 / H
    y = s @ W_gt.T + rng.normal(size=(N, D))
    return W_gt, y, {"W": W_gt + 0.3 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.2}


def test_em_loop_pipelining_is_transparent(dev, monkeypatch):
    """The EM-loop shortcuts (W kept on the device between steps, prefetched / speculative scores GEMM) never
    change results (up to the run-to-run rounding of the f64 atomics in the split-K GEMMs and statistics): same
    parameters with and without them, also when the caller edits W in place or feeds other parameters in between."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    _, y, p0 = _bsc_problem(11)
    an = _An(T=1.0)

    def run(speculate, edit):
        m = BSC_ET(96, 32, 6, 3)
        m.speculate = speculate
        p = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in p0.items()}
        out = []
        for it in range(6):
            if edit and it == 3:
                p["W"][:, 0] *= 1.5            # in place, on the array M_step handed out
            if edit and it == 4:
                p = dict(p, W=np.ascontiguousarray(p["W"]) + 0.01)      # a different array, C order
            p = m.step(an, p, {"y": y})
            out.append((p["W"].copy(), p["pi"], p["sigma"]))
        return out, m

    for edit in (False, True):
        a, ma = run(True, edit)
        b, _ = run(False, edit)
        for (Wa, pa, sa), (Wb, pb, sb) in zip(a, b):
            np.testing.assert_allclose(Wa, Wb, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose([pa, sa], [pb, sb], rtol=1e-10)
    # a fresh model fed the edited parameters gives the same next step as the pipelined one
    a, _ = run(True, True)
    b, _ = run(True, False)
    m2 = BSC_ET(96, 32, 6, 3)
    p = {"W": a[2][0].copy(), "pi": a[2][1], "sigma": a[2][2]}
    p["W"][:, 0] *= 1.5
    q = m2.step(an, p, {"y": y})
    np.testing.assert_allclose(q["W"], a[3][0], rtol=1e-9, atol=1e-10)
    assert not np.allclose(q["W"], b[3][0], rtol=1e-3)          # ... and the edit did change the outcome


def test_em_run_with_partial_data_and_parameter_noise(dev):
    """EM.run with anneal['partial'] < 1 and W_noise > 0 (select_partial_data / noisify_params,
    camodels/__init__.py:124-161, em/__init__.py:97-150): device-resident rows are sub-sampled, the noisy W
    misses the seeded parameters, the loop still converges on the data."""
    from prosper_amd.em import EM
    from prosper_amd.em.annealing import LinearAnnealing
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    W_gt, y, p0 = _bsc_problem(12)
    model = BSC_ET(96, 32, 6, 3)
    anneal = LinearAnnealing(12)
    anneal["T"] = [(0, 1.5), (.6, 1.)]
    anneal["partial"] = [(0, .5), (.5, 1.)]
    anneal["W_noise"] = [(0, 0.05), (.5, 0.)]
    anneal["anneal_prior"] = False
    np.random.seed(5)
    h = dlog.set_handler(("L", "N"), StoreInMemory)
    try:
        em = EM(model=model, anneal=anneal, data={"y": y}, lparams=dict(p0))
        em.run()
    finally:
        dlog.remove_handler(h)
    N_seen = np.array(h.tables["N"], dtype=int)
    assert N_seen[0] == 1500 and N_seen[-1] == 3000
    assert np.isfinite(em.lparams["W"]).all() and np.isfinite(em.lparams["sigma"])
    L = np.array(h.tables["L"], dtype=float)
    assert np.isfinite(L).all() and (np.diff(L[-4:]) > -1e-9).all()      # plain EM once T = 1, all data, no noise


# ------------------------------------------------------------------------- BSC vs oracle
@pytest.mark.parametrize("D,H,Hp,gamma,N,T,ncut,ap", [
    (256, 128, 6, 3, 3000, 1.0, 0.0, False),
    (100, 300, 8, 4, 777, 1.3, 0.7, True),          # H > 256: the blocked device inverse (_spd_inverse_blocked)
    (48, 600, 5, 3, 4000, 1.0, 0.0, False),         # H > 512: generic row kernels, two levels of blocking
    (64, 64, 3, 2, 500, 1.0, 0.3, False),
    (32, 20, 2, 2, 129, 2.0, 0.0, False),
    (64, 256, 16, 2, 4000, 1.2, 0.6, False),        # rows16 fits, its list-writing M-step pass does not (66 KB of LDS):
    (64, 240, 16, 2, 4000, 1.2, 0.0, False),        # ... the plain pass + the dense statistics GEMM run (round-3 advisor finding)
])
def test_bsc_step_matches_oracle(dev, D, H, Hp, gamma, N, T, ncut, ap):
    from oracle import bsc_oracle as O
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    rng = np.random.RandomState(D + H + N)
    W_gt = rng.normal(size=(D, H))
    pi_gt = 2.0 / H
    y, _ = O.generate_bsc_data(W_gt, pi_gt, 1.0, N, rng)
    params = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": pi_gt * 1.2, "sigma": 1.1}
    om = O.make_model(D, H, Hp, gamma)
    an = O.Anneal(T=T, Ncut_factor=ncut, anneal_prior=ap)
    ref, rlog = O.em_step(an, om, dict(params), y, stats_fn=O.m_step_stats_vec, vec=True)

    m = BSC_ET(D, H, Hp, gamma)
    if Hp == 16:
        # (round 5: the 16-lane row kernels may take the CU's whole LDS, so the list-writing M-step pass fits here too; the
        # H = 240 case keeps the plain pass + dense statistics GEMM combination of the round-3 advisor finding covered)
        assert m._state_tables()["fast"] and m._state_tables()["fast_nz"]
        if H == 240:
            m._state_tables()["fast_nz"] = False
    h = dlog.set_handler(("L", "N_use"), StoreInMemory)
    try:
        new = m.step(_An(T=T, Ncut_factor=ncut, anneal_prior=ap), dict(params), {"y": y})
    finally:
        dlog.remove_handler(h)
    assert int(h.tables["N_use"][0]) == rlog["N_use"]
    np.testing.assert_allclose(float(h.tables["L"][0]), rlog["L"], rtol=1e-10)
    # W_new solves Wq X = Wp; the reference's SVD lstsq and the device Cholesky each carry an
    # error ~ cond(Wq) * eps, so the comparison is scaled by the conditioning of this case
    tol = max(1e-8, 20 * np.linalg.cond(rlog["stats"]["Wq"]) * np.finfo(float).eps)
    assert tol < BASELINE_RTOL
    np.testing.assert_allclose(new["W"], ref["W"], rtol=10 * tol, atol=tol * np.abs(ref["W"]).max())
    np.testing.assert_allclose(new["pi"], ref["pi"], rtol=1e-9)
    np.testing.assert_allclose(new["sigma"], ref["sigma"], rtol=1e-9)


# ------------------------------------------------------------------------- one-kernel E-step vs the two-kernel path
@pytest.mark.parametrize("D,H,Hp,gamma,N,with_mu", [
    (1024, 256, 8, 4, 4500, False),     # config-2 dims, ragged last tile (4500 = 70 * 64 + 20)
    (64, 200, 7, 3, 1000, False),       # H not a multiple of 16: clamped latent rows, masked columns
    (40, 100, 5, 3, 333, True),         # H <= 128 instantiation, D padded to 40 -> 40, data offset mu
    (25, 10, 5, 3, 129, False),         # config-1 dims: K dimension zero-padded 25 -> 32
    (512, 256, 6, 6, 64, False),        # gamma = H' (2^H' - H' - 1 multi-cause states), exactly one tile
])
def test_fused_estep_matches_two_kernel_path(dev, D, H, Hp, gamma, N, with_mu):
    """pm_bsc_estep_fused_f64 (scores GEMM + select + E-step in one launch) against pm_gemm_nt_f64 followed by
    pm_bsc_select_estep_f64: same candidates, same log-joints, same log-evidence -- in all three modes (selection only,
    E-step on given candidates, both)."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    rng = np.random.RandomState(D + H + N)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 2.5 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": 2.5 / H, "sigma": 1.1}
    if with_mu:
        params["mu"] = 0.3 * rng.normal(size=D)
    an = _An(T=1.3, anneal_prior=True)
    out = {}
    for path in ("fused", "rows16"):
        m = _set_path(BSC_ET(D, H, Hp, gamma), path)
        assert m._fused() == (path == "fused")
        p = dict(params)
        d = m.select_Hprimes(p, {"y": y})
        ss = m.E_step(an, p, d)                                    # mode 3: selection + log-joints in one pass
        cand, lp, lse = np.asarray(d["candidates"]), np.asarray(ss["logpj"]), ss["logpj"].lse.cpu().numpy()
        m2 = _set_path(BSC_ET(D, H, Hp, gamma), path)
        c1 = np.asarray(m2.select_Hprimes(dict(params), {"y": y})["candidates"])      # mode 1: selection alone
        ss2 = m2.E_step(an, dict(params), {"y": y, "candidates": cand})               # mode 2: given candidates
        assert np.array_equal(c1, cand)
        # (not bit for bit: small shards of the two-kernel path go through the split-K GEMM, whose f64 atomics sum in
        # run-to-run order)
        np.testing.assert_allclose(np.asarray(ss2["logpj"]), lp, rtol=1e-12, atol=1e-10)
        out[path] = (cand, lp, lse)
    assert np.array_equal(out["fused"][0], out["rows16"][0])
    np.testing.assert_allclose(out["fused"][1], out["rows16"][1], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(out["fused"][2], out["rows16"][2], rtol=1e-12, atol=1e-10)


def _expect_of(m):
    """E[s] rows of the model's last statistics pass: the dense buffer, with the rows the list-writing pass did not store
    (pm_bsc_estep_fused8_nz_f64 keeps the dense row of a datapoint only when its list overflowed) rebuilt from the lists."""
    E = m._ws["expect"].cpu().numpy().copy()
    if getattr(m, "_expect_rows", "all") == "overflowed":
        idx = m._ws["nz_idx"].cpu().numpy().view(np.uint16).astype(np.int64)
        val = m._ws["nz_val"].cpu().numpy()
        listed = idx[:, 0] != 0xFFFE
        E[listed] = 0.0
        rows = np.repeat(np.arange(E.shape[0]), idx.shape[1]).reshape(idx.shape)
        sel = (idx != 0xFFFF) & listed[:, None]
        E[rows[sel], idx[sel]] = val[sel]
    return E


@pytest.mark.parametrize("D,H,Hp,gamma,N", [(1024, 256, 8, 4, 40000), (64, 200, 7, 3, 3000), (25, 10, 5, 3, 333),
                                            # round 6: H' = 5 .. 7 on the 16-wavefront kernel (main launch + TAIL)
                                            (128, 256, 6, 3, 40000), (64, 192, 5, 4, 36000), (64, 200, 7, 4, 34000)])
def test_step_with_fused_mstep_statistics(dev, D, H, Hp, gamma, N):
    """Inside ``step`` with no data truncation ahead the fused E-step kernel also produces the M-step's per-datapoint
    statistics (E[s] rows, Wq block, mus, scalars; the ragged last round of a large shard still goes through
    pm_bsc_mstep_rows16_f64): same new parameters, free energy and statistics buffer as the separate M-step pass."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    rng = np.random.RandomState(N + H)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 3.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.05}
    out = {}
    for fuse in (True, False):
        m = BSC_ET(D, H, Hp, gamma)
        m.fuse_mstats = fuse
        h = dlog.set_handler(("L", "N_use"), StoreInMemory)
        try:
            new = m.step(_An(T=1.1), dict(params), {"y": y})
        finally:
            dlog.remove_handler(h)
        out[fuse] = (new, float(h.tables["L"][0]), int(h.tables["N_use"][0]), m._ws["stats"].cpu().numpy().copy(),
                     _expect_of(m))
    a, b = out[True], out[False]
    assert a[2] == b[2] == N
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12)
    np.testing.assert_allclose(a[4], b[4], rtol=1e-10, atol=1e-14)                       # E[s]
    # packed statistics; the diagonal of the second moments may sit in the Wq block, in qdiag, or in both (the whole-shard
    # passes leave all of it in qdiag): compare the assembled diagonal
    from prosper_amd import _lib
    lib = _lib.load()
    o_wq, o_qd, o_mus = lib.pm_bsc_stats_offset_wq(H, D), lib.pm_bsc_stats_offset_qdiag(H, D), lib.pm_bsc_stats_offset_mus(H, D)

    def assembled(st):
        st = st.copy()
        st[lib.pm_bsc_stats_offset_scalars(H, D) + 3] = 0.0     # (count of overflowing non-zero lists: only where lists are made)
        Wq = st[o_wq:o_qd].reshape(H, H)
        st[o_qd:o_mus] += np.diagonal(Wq)
        Wq[np.arange(H), np.arange(H)] = 0.0
        return st
    np.testing.assert_allclose(assembled(a[3]), assembled(b[3]), rtol=1e-9, atol=1e-9 * np.abs(b[3]).max())
    for k in ("W", "pi", "sigma"):
        np.testing.assert_allclose(a[0][k], b[0][k], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("N,T,overflow,D,H,gamma", [
    (40000, 1.0, False, 128, 256, 4), (3000, 1.0, False, 128, 256, 4), (40000, 40.0, True, 128, 256, 4),
    (777, 6.0, None, 128, 256, 4),
    (35000, 1.0, False, 200, 200, 3),     # H < 256 (guarded latent slots), D not a multiple of the 64-column chunks
    (5000, 1.0, None, 72, 200, 3),        # ... and a broad posterior: some lists overflow, the dense product runs
])
def test_sparse_wp_from_nonzero_lists(dev, N, T, overflow, D, H, gamma):
    """The statistics pass leaves every E[s] row as a list of its non-zeros too (pm_bsc_estep_fused8_nz_f64), and
    Wp = E[s]^T Y (bsc_et.py:339-363) is accumulated from the lists (pm_bsc_wp_sparse_f64).  The lists hold exactly the
    non-zeros of the dense rows, bit for bit; the statistics and the new parameters equal those of the dense product;
    rows with more than PM_BSC_NZ_MAX non-zeros (a hot annealing temperature) are counted and the dense product runs
    instead, decided on the device."""
    from prosper_amd import _lib
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    Hp = 8
    rng = np.random.RandomState(N + int(T))
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 3.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.05}
    out = {}
    for sparse in (True, False):
        m = BSC_ET(D, H, Hp, gamma)
        assert m._tile8_whole_shard()
        m.sparse_wp = sparse
        names = []
        orig = m._call
        m._call = lambda label, name, *a, _o=orig, _n=names: (_n.append(name), _o(label, name, *a))[1]
        if sparse:
            m._buf("expect", (N, H)).fill_(float("nan"))       # (what the list-writing pass does not store stays NaN)
        new = m.step(_An(T=T), dict(params), {"y": y})
        out[sparse] = (new, m._ws["stats"].cpu().numpy().copy(), m._ws["expect"].cpu().numpy().copy(), names)
        if sparse:
            idx = m._ws["nz_idx"].cpu().numpy().view(np.uint16).astype(np.int64)
            val = m._ws["nz_val"].cpu().numpy()
            E_lists = _expect_of(m)
    a, b = out[True], out[False]
    assert a[3].index("pm_bsc_wp_sparse_expand_f64") < a[3].index("pm_gemm_tn_acc_gated_f64")
    assert not any("wp_sparse" in n for n in b[3]) and "pm_gemm_tn_acc_f64" in b[3]
    o_sc = _lib.load().pm_bsc_stats_offset_scalars(H, D)
    E = b[2]                                                    # the dense pass (no lists) stores every row
    assert np.isfinite(E).all()
    nnz = (E != 0).sum(axis=1)
    n_over = int(a[1][o_sc + 3])
    assert n_over == int((nnz > 16).sum())
    if overflow is not None:
        assert (n_over > 0) == overflow
    # the lists: the non-zeros of the dense row, in ascending-lane order of the kernel, 0xFFFF behind them; slot 0 of an
    # overflowed list is the marker, and only THOSE datapoints' dense rows are stored by the list-writing pass -- unless
    # some list overflowed: then the dense product ran, behind pm_bsc_expand_lists_gated_f64, on a complete buffer
    ok = nnz <= 16
    assert np.array_equal(idx[:, 0] == 0xFFFE, ~ok)
    assert np.array_equal(a[2][~ok], E[~ok])
    if n_over:
        assert np.array_equal(a[2], E)
    else:
        assert np.isnan(a[2]).all()
    assert np.array_equal(E_lists, E)
    cnt = (idx != 0xFFFF).sum(axis=1)
    assert np.array_equal(cnt[ok], nnz[ok])
    rebuilt = np.zeros_like(E)
    rows = np.repeat(np.arange(N), 16).reshape(N, 16)
    sel = (idx != 0xFFFF) & ok[:, None]
    rebuilt[rows[sel], idx[sel]] = val[sel]
    assert np.array_equal(rebuilt[ok], E[ok])
    assert all(len(set(r[r != 0xFFFF])) == (r != 0xFFFF).sum() for r in idx[ok][:500])
    # statistics and parameters
    sa, sb = a[1].copy(), b[1].copy()
    sa[o_sc + 3] = sb[o_sc + 3] = 0.0
    np.testing.assert_allclose(sa, sb, rtol=1e-9, atol=1e-11 * np.abs(sb).max())
    Wp_ref = E.T @ y
    np.testing.assert_allclose(a[1][:H * D].reshape(H, D), Wp_ref, rtol=1e-10, atol=1e-11 * np.abs(Wp_ref).max())
    for k in ("W", "pi", "sigma"):
        np.testing.assert_allclose(a[0][k], b[0][k], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("D,H,Hp,gamma,N,T", [(128, 256, 8, 4, 20000, 1.0), (512, 256, 8, 4, 6000, 1.0), (64, 100, 6, 3, 3000, 1.0),
                                              (64, 100, 6, 3, 3000, 50.0)])
def test_sparse_wp_after_data_truncation(dev, D, H, Hp, gamma, N, T):
    """With data truncation ahead the statistics need the global cut first, so the M-step runs its own per-datapoint pass
    (pm_bsc_mstep_rows16_nz_f64): it leaves the non-zeros of E[s] as lists too and Wp comes from them.  Same statistics
    and parameters as the dense product; the lists hold exactly the non-zeros of the dense rows."""
    from prosper_amd import _lib
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    rng = np.random.RandomState(N + H)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 3.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.05}
    out = {}
    for sparse in (True, False):
        m = BSC_ET(D, H, Hp, gamma)
        m.sparse_wp = sparse
        m.defer_stats = False          # (this test holds the M-step's OWN pass; the deferred form: test_anneal_path_gpu.py)
        names = []
        orig = m._call
        m._call = lambda label, name, *a, _o=orig, _n=names: (_n.append(name), _o(label, name, *a))[1]
        new = m.step(_An(T=T, Ncut_factor=0.6), dict(params), {"y": y})
        out[sparse] = (new, m._ws["stats"].cpu().numpy().copy(), m._ws["expect"].cpu().numpy().copy(), names)
        if sparse:
            idx = m._ws["nz_idx"].cpu().numpy().view(np.uint16).astype(np.int64)
            val = m._ws["nz_val"].cpu().numpy()
    a, b = out[True], out[False]
    assert "pm_bsc_mstep_rows16_nz_f64" in a[3] and "pm_bsc_wp_sparse_f64" in a[3] and "pm_bsc_wp_sparse_f64" not in b[3]
    o_sc = _lib.load().pm_bsc_stats_offset_scalars(H, D)
    E = a[2]
    assert (E == 0).all(axis=1).sum() > N // 10                 # the cut datapoints: empty rows, empty lists
    nnz = (E != 0).sum(axis=1)
    n_over = int(a[1][o_sc + 3])
    assert n_over == int((nnz > 16).sum())                      # (some rows do overflow at these small D, all at T = 50)
    assert T < 10 or n_over > N // 4
    ok = nnz <= 16
    assert np.array_equal((idx != 0xFFFF).sum(axis=1)[ok], nnz[ok])
    rebuilt = np.zeros_like(E)
    rows = np.repeat(np.arange(N), 16).reshape(N, 16)
    sel = (idx != 0xFFFF) & ok[:, None]
    rebuilt[rows[sel], idx[sel]] = val[sel]
    assert np.array_equal(rebuilt[ok], E[ok])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-12, atol=1e-300)
    sa, sb = a[1].copy(), b[1].copy()
    sa[o_sc + 3] = sb[o_sc + 3] = 0.0
    np.testing.assert_allclose(sa, sb, rtol=1e-9, atol=1e-11 * np.abs(sb).max())
    for k in ("W", "pi", "sigma"):
        np.testing.assert_allclose(a[0][k], b[0][k], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("D,H,Hp,gamma,N", [(256, 64, 6, 3, 40000), (25, 10, 5, 3, 333)])
def test_em_loop_with_speculative_estep(dev, D, H, Hp, gamma, N):
    """On a flat annealing schedule the M-step launches the next step's E-step itself, as soon as pi_new / sigma_new
    are on the host and while the device still solves for W (``_speculate_estep``); ``E_step`` adopts that pass when it
    is called with exactly those parameters and drops it otherwise (here: a temperature change, a data-truncation
    step, a caller that edits pi).  Same parameters and free energies as the loop that never speculates."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    rng = np.random.RandomState(N + H)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 2.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 2.0 / H, "sigma": 1.05}
    #        T, Ncut_factor, edit pi before the step
    plan = [(1.2, 0.0, False), (1.0, 0.0, False), (1.0, 0.0, False), (1.0, 0.0, False), (1.0, 0.0, False),
            (1.0, 0.0, True), (1.0, 0.0, False), (1.0, 0.0, False), (1.1, 0.0, False), (1.1, 0.0, False),
            (1.1, 0.0, False), (1.1, 0.9, False), (1.1, 0.9, False), (1.1, 0.9, False)]
    out = {}
    for spec in (True, False):
        m = BSC_ET(D, H, Hp, gamma)
        m.speculate_estep = spec
        m.speculate = True                            # (whatever PM_SPECULATE* say in the environment)
        h = dlog.set_handler(("L", "N_use"), StoreInMemory)
        p, trace = dict(params), []
        try:
            for T, ncut, edit in plan:
                if edit:
                    p = dict(p, pi=p["pi"] * 1.01)
                p = m.step(_An(T=T, Ncut_factor=ncut), p, {"y": y})
                trace.append((np.array(p["W"]), float(p["pi"]), float(p["sigma"])))
        finally:
            dlog.remove_handler(h)
        out[spec] = (trace, np.array(h.tables["L"]), np.array(h.tables["N_use"]), m.spec_hits)
    a, b = out[True], out[False]
    assert b[3] == 0
    if BSC_ET(D, H, Hp, gamma)._fused():
        # adopted: steps 4, 5 (flat since step 2/3), 8 (flat again after the edit), 11 (flat T=1.1), 14 (flat Ncut)
        assert a[3] >= 4, a[3]
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_allclose(a[1], b[1], rtol=1e-11)
    for (Wa, pa, sa), (Wb, pb, sb) in zip(a[0], b[0]):
        np.testing.assert_allclose(Wa, Wb, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose([pa, sa], [pb, sb], rtol=1e-10)


def test_host_lstsq_fallback_under_speculation(dev, monkeypatch):
    """When the device solve is rejected (numerically singular Wq) the M-step falls back to LAPACK's lstsq on the host
    -- AFTER the speculative next E-step has been enqueued, which zeroes and refills the statistics workspace the
    right-hand side Wp lives in.  The fallback must read its own copy (round-2 advisor finding: it read zeros and
    returned W ~ 0 silently).  The rejection is forced at steps 3 and 4 of a flat-schedule loop (the steps at which
    the M-step speculates); same W as the loop that never speculates, and as the device solve of the same system."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 64, 24, 5, 3, 4000
    rng = np.random.RandomState(11)
    W_gt = rng.normal(size=(D, H))
    y = (rng.random_sample((N, H)) < 2.0 / H) @ W_gt.T + rng.normal(size=(N, D))
    params = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 2.0 / H, "sigma": 1.05}
    an = _An(T=1.0)

    def loop(spec, reject):
        m = BSC_ET(D, H, Hp, gamma)
        m.speculate, m.speculate_estep = True, spec
        calls = {"n": 0}
        real = BSC_ET._solve_ok

        def solve_ok(*a):
            calls["n"] += 1
            return False if calls["n"] in reject else real(*a)
        monkeypatch.setattr(m, "_solve_ok", solve_ok)
        p, trace = dict(params), []
        for _ in range(6):
            p = m.step(an, p, {"y": y})
            trace.append(np.array(p["W"]))
        return trace, m.spec_hits

    a, hits = loop(True, (3, 4))
    b, _ = loop(False, (3, 4))
    c, _ = loop(False, ())
    if BSC_ET(D, H, Hp, gamma)._fused():
        assert hits >= 1
    for Wa, Wb, Wc in zip(a, b, c):
        assert np.abs(Wa).max() > 0.1
        np.testing.assert_allclose(Wa, Wb, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(Wa, Wc, rtol=1e-6, atol=1e-8)


def test_warm_inverse_is_transparent_across_unrelated_problems(dev):
    """The W solve warm-starts from the model's previous inverse (pm_spd_inverse_warm_f64).  A model that is handed an
    unrelated problem next (other data, other parameters: the start residual is large, the device falls back to the
    exact sweep) and one that sees a slightly perturbed problem (refined, sweep skipped) both return what a fresh model
    returns."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 64, 40, 5, 3, 3000
    rng = np.random.RandomState(3)

    def problem(scale, seed):
        r = np.random.RandomState(seed)
        W_gt = scale * r.normal(size=(D, H))
        y = (r.random_sample((N, H)) < 2.5 / H) @ W_gt.T + r.normal(size=(N, D))
        return {"W": W_gt + 0.1 * r.normal(size=(D, H)), "pi": 2.5 / H, "sigma": 1.1}, y

    pa, ya = problem(1.0, 1)
    pb, yb = problem(3.0, 2)                     # unrelated
    pc, yc = dict(pa, W=pa["W"] * (1 + 1e-3 * rng.normal(size=(D, H)))), ya     # close to the first
    an = _An(T=1.0)
    m = BSC_ET(D, H, Hp, gamma)
    m.step(an, dict(pa), {"y": ya})
    for p, y in ((pb, yb), (pc, yc), (pa, ya)):
        got = m.step(an, dict(p), {"y": y})
        ref = BSC_ET(D, H, Hp, gamma).step(an, dict(p), {"y": y})
        np.testing.assert_allclose(got["W"], ref["W"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose([got["pi"], got["sigma"]], [ref["pi"], ref["sigma"]], rtol=1e-12)


@pytest.mark.parametrize("cond", [1e3, 1e9])
def test_solve_after_a_rejected_warm_start_is_refined(dev, cond):
    """Round-3 advisor finding: the solve's refinement pass was skipped on the host's guess that the device would accept
    the inverse's warm start; when the device rejected it (start residual >= 0.1) the sweep's inverse was applied
    UNREFINED, i.e. the accuracy of W depended on the call history.  Now the device's verdict travels with the pivots and a
    rejected start repeats the solve with the refinement: the result equals, bit for bit, what a model without history
    (cold sweep + refinement) returns, and solves Wq X = Wp like LAPACK's lstsq -- also for cond(Wq) = 1e9."""
    from prosper_amd.em.camodels._device import DeviceCAModel
    H, D = 96, 80
    rs = np.random.RandomState(int(np.log10(cond)))

    def spd(c):
        Q, _ = np.linalg.qr(rs.normal(size=(H, H)))
        return (Q * np.logspace(0, -np.log10(c), H)) @ Q.T

    def run(model, A, B):
        Wq_u = torch.from_numpy(np.triu(A) - np.diag(np.diag(A)) * 0.5).to(dev)       # upper triangle, half the diagonal ...
        qd = torch.from_numpy(np.diag(A) * 0.5).to(dev)                              # ... the other half as diag_add
        X, status, _ = model._solve_normal_eq(Wq_u, qd, torch.from_numpy(B).to(dev))
        host = torch.cat([status, X.reshape(-1)]).cpu().numpy()
        assert DeviceCAModel._solve_ok(host[0], host[1])
        redo = model._solve_accurate(float(host[2]))
        return (redo if redo is not None else host[3:].reshape(H, D)), float(host[2]), redo is not None

    A1, A2 = spd(10.0), spd(cond)
    A1, A2 = 0.5 * (A1 + A1.T), 0.5 * (A2 + A2.T)
    B = rs.normal(size=(H, D))
    warm = DeviceCAModel(D, H, 4, 2)
    _, flag, redone = run(warm, A1, B)
    assert flag == 1.0 and not redone                           # first call: cold sweep, refined
    X_w, flag, redone = run(warm, A2, B)                        # unrelated matrix: the device rejects the warm start
    assert flag == 0.0 and redone
    X_c, flag, redone = run(DeviceCAModel(D, H, 4, 2), A2, B)   # no history
    assert flag == 1.0 and not redone
    assert np.array_equal(X_w, X_c)
    ref = np.linalg.lstsq(A2, B, rcond=None)[0]
    resid = lambda X: np.abs(A2 @ X - B).max() / (np.abs(A2).max() * np.abs(X).max())
    assert resid(X_w) < 50 * np.finfo(float).eps * H            # backward error at rounding level, whatever cond is
    np.testing.assert_allclose(X_w, ref, rtol=0, atol=1e-13 * cond * np.abs(ref).max())
    X_n, flag, redone = run(warm, A2 * (1 + 1e-4), B)           # a nearby matrix next: accepted, nothing repeated
    assert flag == 1.0 and not redone
    np.testing.assert_allclose(X_n * (1 + 1e-4), X_c, rtol=0, atol=1e-13 * cond * np.abs(ref).max())


def test_config2_fullrank_w_through_cold_and_warm_inverse(dev):
    """bsc_step_c2_fullrank.npz: config-2 dimensions with N = 512 > H datapoints and a full-rank Wq (smin / smax =
    4e-6), so the reference's W_new = lstsq(Wq, Wp) is well posed and is compared itself -- through the exact sweep (a
    fresh model), through the warm-started inverse (the same model called again: Newton-Schulz from its previous inverse,
    sweep skipped) and on the two-kernel path."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    g = golden("bsc_step_c2_fullrank.npz")
    assert not rank_deficient(g)
    an = _An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
    params = {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}
    for path in ("fused", "rows16"):
        m = _set_path(BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"])), path)
        for call in ("cold", "warm", "warm again"):
            p = dict(params)
            data = m.select_Hprimes(p, {"y": g["y"]})
            ss = m.E_step(an, p, data)                        # (inserts p['mu'], as upstream)
            new = m.M_step(an, p, ss, data)
            assert (getattr(m, "_winv_prev", None) is not None), call
            Wp, Wq = _device_stats(m)
            np.testing.assert_allclose(Wq, g["Wq"], rtol=1e-9, atol=1e-12 * np.abs(g["Wq"]).max(), err_msg=call)
            np.testing.assert_allclose(Wp, g["Wp"], rtol=1e-9, atol=1e-11 * np.abs(g["Wp"]).max(), err_msg=call)
            np.testing.assert_allclose(new["W"], g["W_new"], rtol=1e-8, atol=1e-8 * np.abs(g["W_new"]).max(),
                                       err_msg="%s / %s" % (path, call))
            np.testing.assert_allclose([new["pi"], new["sigma"]], [float(g["pi_new"]), float(g["sigma_new"])], rtol=1e-10)


def test_constant_partial_never_reuses_a_stale_shard(dev):
    """Regression (round 1): with a constant ``anneal['partial'] = 0.9`` every step draws a NEW random subset of the rows
    (camodels/__init__.py:125-152) of the same shape as the previous one; the resident-shard cache must key on the
    rows' content, not on their shape.  Each step is compared with the oracle on exactly the subset the step drew."""
    from oracle import bsc_oracle as O
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 48, 20, 5, 3, 1500
    rng = np.random.RandomState(23)
    W_gt = rng.normal(size=(D, H))
    y, _ = O.generate_bsc_data(W_gt, 2.0 / H, 1.0, N, rng)
    params = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": 2.0 / H, "sigma": 1.1}
    model = O.make_model(D, H, Hp, gamma)
    m = BSC_ET(D, H, Hp, gamma)
    an = _An(T=1.0, partial=0.9)
    p = dict(params)
    subsets = []
    for step in range(4):
        np.random.seed(100 + step)
        sel = np.sort(np.random.permutation(N)[:int(np.ceil(N * 0.9))])
        subsets.append(sel)
        np.random.seed(100 + step)                         # CAModel.step draws the same permutation
        ref, _ = O.em_step(O.Anneal(T=1.0), model, dict(p), y[sel], stats_fn=O.m_step_stats_vec, vec=True)
        p = m.step(an, dict(p), {"y": y})
        np.testing.assert_allclose(p["W"], ref["W"], rtol=1e-7, atol=1e-9, err_msg="step %d" % step)
        np.testing.assert_allclose([p["pi"], p["sigma"]], [ref["pi"], ref["sigma"]], rtol=1e-9, err_msg="step %d" % step)
    assert not np.array_equal(subsets[0], subsets[1])


def test_config2_full_shard_against_oracle(dev):
    """BASELINE config 2 at its real size -- D=1024 H=256 H'=8 gamma=4, N = 200 000 drawn by bench.py's default recipe
    (np.random.RandomState(0) for W_gt and the start W, RandomState(rank) for the rows, latents then noise)
    -- through the shipped launches (one fused E-step launch of 3072 workgroups = 196 608 rows of whole rounds + the
    3 392-row remainder path; with PM_FUSED=0: whole GEMM rounds + fused K-slices).  Datapoints are independent given the parameters, so the vectorised oracle is run
    on ~1000 sampled rows (spread over the shard, dense around the whole-rounds / remainder boundary and at the end):
    candidates exact, log-joints to 1e-10; the M-step statistics of the sample as its own shard against the oracle's,
    and additivity of the full shard's statistics over two halves."""
    from oracle import bsc_oracle as O
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 1024, 256, 8, 4, 200_000
    rs0 = np.random.RandomState(0)
    W_gt_h = rs0.randn(D, H)
    W0 = np.ascontiguousarray((W_gt_h + 0.1 * rs0.randn(D, H)).T).T
    W_gt = torch.from_numpy(W_gt_h).to(dev)
    rs = np.random.RandomState(0)                        # rank 0's rows
    Y = torch.empty(N, D, dtype=torch.float64, device=dev)
    for lo in range(0, N, 25_000):
        S = torch.from_numpy((rs.random_sample((25_000, H)) < 4.0 / H).astype(np.float64)).to(dev)
        noise = torch.from_numpy(rs.normal(size=(25_000, D))).to(dev)
        Y[lo:lo + 25_000] = torch.addmm(noise, S, W_gt.t())
    del S, noise
    params = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
    an = _An(T=1.0)
    rng = np.random.RandomState(5)
    rows = np.unique(np.concatenate([rng.randint(0, N, size=600), np.arange(196_608 - 100, 196_608 + 100),
                                     np.arange(N - 100, N), np.arange(0, 64)]))
    y_s = Y[torch.from_numpy(rows).to(dev)].cpu().numpy()
    om = O.make_model(D, H, Hp, gamma)
    cand_ref = O.select_hprimes_vec(W0, y_s, Hp)
    lp_ref = O.e_step_vec(O.Anneal(T=1.0), W0, params["pi"], params["sigma"], params["mu"], y_s, cand_ref,
                          om["SM"], om["state_abs"])
    st_ref = O.m_step_stats_vec(W0, params["mu"], y_s, cand_ref, lp_ref, om["SM"])

    for path in ("fused", "rows16"):
        m = _set_path(BSC_ET(D, H, Hp, gamma), path)
        d = m.select_Hprimes(dict(params), {"y": Y})
        ss = m.E_step(an, dict(params), d)
        idx = torch.from_numpy(rows).to(dev)
        cand = d["candidates"].tensor[idx].cpu().numpy()
        assert np.array_equal(cand, cand_ref), path
        np.testing.assert_allclose(ss["logpj"].tensor[idx].cpu().numpy(), lp_ref, rtol=1e-10, atol=1e-9, err_msg=path)
        torch.testing.assert_close(ss["logpj"].lse, torch.logsumexp(ss["logpj"].tensor, dim=1), rtol=1e-12, atol=1e-10)
        # M-step statistics: the full shard, then additivity over two halves
        m.M_step(an, dict(params), ss, d)
        full = m._ws["stats"].clone()
        if path == "fused":
            acc = torch.zeros_like(full)
            for sl in (slice(0, 98_765), slice(98_765, N)):
                mh = _set_path(BSC_ET(D, H, Hp, gamma), path)
                dh = mh.select_Hprimes(dict(params), {"y": Y[sl]})
                mh.M_step(an, dict(params), mh.E_step(an, dict(params), dh), dh)
                acc += mh._ws["stats"]
                del mh, dh
            torch.testing.assert_close(acc, full, rtol=1e-9, atol=1e-8)
        del m, d, ss
    # the sampled rows as a shard of their own: statistics against the oracle's
    ms = _set_path(BSC_ET(D, H, Hp, gamma), "fused")
    ds = ms.select_Hprimes(dict(params), {"y": y_s})
    ms.M_step(an, dict(params), ms.E_step(an, dict(params), ds), ds)
    Wp, Wq = _device_stats(ms)
    np.testing.assert_allclose(Wp, st_ref["Wp"], rtol=1e-9, atol=1e-11 * np.abs(st_ref["Wp"]).max())
    np.testing.assert_allclose(Wq, st_ref["Wq"], rtol=1e-9, atol=1e-12 * np.abs(st_ref["Wq"]).max())
    from prosper_amd import _lib
    sc = ms._ws["stats"].cpu().numpy()[_lib.load().pm_bsc_stats_offset_scalars(H, D):][:3]
    np.testing.assert_allclose(sc[0], st_ref["sigma"], rtol=1e-10)
    assert int(round(sc[2])) == len(rows)


def test_config2_full_shard_every_row_against_oracle(dev):
    """The same shard, EVERY row (round 6; the sampled test above was the answer to a slow host -- the GPU boxes' hosts run the
    vectorised oracle at ~20 k datapoints/s): candidates of all 200 000 datapoints equal to the oracle's, all 82 M log-joints
    to 1e-10, and the full shard's M-step statistics (Wp, Wq, the sigma sum, the kept count) against the oracle's summed over
    chunks -- through the shipped launches (1536 sixteen-wavefront workgroups + the TAIL launch), statistics from the pass."""
    from oracle import bsc_oracle as O
    from prosper_amd import _lib
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 1024, 256, 8, 4, 200_000
    rs0 = np.random.RandomState(0)
    W_gt_h = rs0.randn(D, H)
    W0 = np.ascontiguousarray((W_gt_h + 0.1 * rs0.randn(D, H)).T).T
    W_gt = torch.from_numpy(W_gt_h).to(dev)
    rs = np.random.RandomState(0)
    Y = torch.empty(N, D, dtype=torch.float64, device=dev)
    for lo in range(0, N, 25_000):
        S = torch.from_numpy((rs.random_sample((25_000, H)) < 4.0 / H).astype(np.float64)).to(dev)
        noise = torch.from_numpy(rs.normal(size=(25_000, D))).to(dev)
        Y[lo:lo + 25_000] = torch.addmm(noise, S, W_gt.t())
    del S, noise
    params = {"W": W0, "pi": 4.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
    an = _An(T=1.0)
    m = BSC_ET(D, H, Hp, gamma)
    p = m.step(an, dict(params), {"y": Y})                  # (the statistics-carrying pass + M-step, as an EM loop runs them)
    Wp, Wq = _device_stats(m)
    sc = m._ws["stats"].cpu().numpy()[_lib.load().pm_bsc_stats_offset_scalars(H, D):][:3]
    d = m.select_Hprimes(dict(params), {"y": Y})
    ss = m.E_step(an, dict(params), d)
    cand, lp = d["candidates"].tensor, ss["logpj"].tensor
    om = O.make_model(D, H, Hp, gamma)
    Wp_ref, Wq_ref, sig_ref, worst, bad = np.zeros((H, D)), np.zeros((H, H)), 0.0, 0.0, 0
    CH = 8192
    for lo in range(0, N, CH):
        y_c = Y[lo:lo + CH].cpu().numpy()
        c_ref = O.select_hprimes_vec(W0, y_c, Hp)
        bad += int((cand[lo:lo + CH].cpu().numpy() != c_ref).any(axis=1).sum())
        lp_ref = O.e_step_vec(O.Anneal(T=1.0), W0, params["pi"], params["sigma"], params["mu"], y_c, c_ref, om["SM"],
                              om["state_abs"])
        got = lp[lo:lo + CH].cpu().numpy()
        worst = max(worst, float(np.max(np.abs(got - lp_ref) / (1e-9 + 1e-10 * np.abs(lp_ref)))))
        st = O.m_step_stats_vec(W0, params["mu"], y_c, c_ref, lp_ref, om["SM"])
        Wp_ref += st["Wp"]
        Wq_ref += st["Wq"]
        sig_ref += st["sigma"]
    assert bad == 0, "%d of %d datapoints with other candidates than the oracle's" % (bad, N)
    assert worst <= 1.0, "log-joints: %.2f times the tolerance (rtol 1e-10, atol 1e-9)" % worst
    np.testing.assert_allclose(Wp, Wp_ref, rtol=1e-9, atol=1e-11 * np.abs(Wp_ref).max())
    np.testing.assert_allclose(Wq, Wq_ref, rtol=1e-9, atol=1e-12 * np.abs(Wq_ref).max())
    np.testing.assert_allclose(sc[0], sig_ref, rtol=1e-10)
    assert int(round(sc[2])) == N and np.isfinite(p["W"]).all()


def test_config2_full_shard_truncation_step_against_oracle(dev):
    """A whole EM step of config 2 at its real size on an ANNEALED, DATA-TRUNCATING point of the reference's schedule (T = 1.6,
    Ncut_factor = 0.5: what 49 of its 50 steps look like) against the oracle on all 200 000 rows: ``CAModel.step`` -- the pass
    that leaves per-datapoint records, the cut selected on the device, the apply kernel, the sparse product, the solve --
    returns the oracle's W, pi, sigma; the same N_use and free energy are logged (round 6)."""
    from oracle import bsc_oracle as O
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    from prosper_amd.utils.datalog import dlog, StoreInMemory
    D, H, Hp, gamma, N = 1024, 256, 8, 4, 200_000
    # (bench.py's own data -- causes of norm 32 in unit noise, four per datapoint -- sits in the regime where the reference's
    # un-stabilised exp(logpj) sums underflow for every datapoint with a cause outside its state set: its cut is 0 and it
    # keeps ALL datapoints, which the device mirrors (N_use = N in both: measured with this test's first version).  Causes of
    # norm ~10, two per datapoint: a genuine cut.)
    rs0 = np.random.RandomState(0)
    W_gt_h = 0.3 * rs0.randn(D, H)
    W0 = np.ascontiguousarray((W_gt_h + 0.03 * rs0.randn(D, H)).T).T
    rs = np.random.RandomState(0)
    y = np.empty((N, D))
    for lo in range(0, N, 25_000):
        S = (rs.random_sample((25_000, H)) < 2.0 / H).astype(np.float64)
        y[lo:lo + 25_000] = rs.normal(size=(25_000, D)) + S @ W_gt_h.T
    Y = torch.from_numpy(y).to(dev)
    params = {"W": W0, "pi": 2.0 / H, "sigma": 1.0, "mu": np.zeros(D)}
    an = _An(T=1.6, Ncut_factor=0.5)
    m = BSC_ET(D, H, Hp, gamma)
    h = dlog.set_handler(("N_use", "L"), StoreInMemory)
    try:
        new = m.step(an, dict(params), {"y": Y})
    finally:
        dlog.remove_handler(h)
    assert m.defer_stats and m._fused()
    om = O.make_model(D, H, Hp, gamma)
    oan = O.Anneal(T=1.6, Ncut_factor=0.5, anneal_prior=False)
    K = 1 + H + om["SM"].shape[0]
    cand = np.empty((N, Hp), dtype=np.int64)
    logpj = np.empty((N, K))
    CH = 8192
    for lo in range(0, N, CH):
        cand[lo:lo + CH] = O.select_hprimes_vec(W0, y[lo:lo + CH], Hp)
        logpj[lo:lo + CH] = O.e_step_vec(oan, W0, params["pi"], params["sigma"], params["mu"], y[lo:lo + CH], cand[lo:lo + CH],
                                         om["SM"], om["state_abs"])
    ref, log = O.m_step(oan, om, W0, params["pi"], params["sigma"], params["mu"], y, cand, logpj,
                        stats_fn=O.m_step_stats_vec, shards=[np.arange(lo, min(lo + CH, N)) for lo in range(0, N, CH)])
    assert int(h.tables["N_use"][0]) == log["N_use"] and 0.5 * N < log["N_use"] < N
    np.testing.assert_allclose(h.tables["L"][0], log["L"], rtol=1e-10)
    np.testing.assert_allclose([new["pi"], new["sigma"]], [ref["pi"], ref["sigma"]], rtol=1e-9)
    np.testing.assert_allclose(new["W"], ref["W"], rtol=0, atol=1e-8 * np.abs(ref["W"]).max())


def test_empty_and_tiny_shards(dev):
    """N = 0 rows on a rank must not launch anything; N = 1 works."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    m = BSC_ET(25, 10, 5, 3)
    rng = np.random.RandomState(0)
    params = {"W": rng.normal(size=(25, 10)), "pi": 0.2, "sigma": 1.0}
    data = m.select_Hprimes(params, {"y": np.zeros((0, 25))})
    assert np.asarray(data["candidates"]).shape == (0, 5)
    ss = m.E_step(_An(T=1.0), params, data)
    assert np.asarray(ss["logpj"]).shape == (0, 31)
    y1 = rng.normal(size=(1, 25))
    d1 = m.select_Hprimes(params, {"y": y1})
    s1 = m.E_step(_An(T=1.0), params, d1)
    assert np.isfinite(np.asarray(s1["logpj"])).all()


# ------------------------------------------------------------------------- config-2 properties
def test_config2_properties(dev):
    """BASELINE config 2 dims (D=1024 H=256 H'=8 gamma=4) at N = 20000: properties that need
    no oracle -- top-H' correctness, log-sum-exp consistency, sharding linearity of the
    statistics, posterior normalisation."""
    from prosper_amd import _lib
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 1024, 256, 8, 4, 20000
    gen = torch.Generator(device=dev).manual_seed(0)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
    S = (torch.rand(N, H, generator=gen, device=dev) < 4.0 / H).to(torch.float64)
    Y = S @ W_gt.t() + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    W0 = (W_gt + 0.1 * torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)).cpu().numpy()
    params = {"W": W0, "pi": 4.0 / H, "sigma": 1.0}
    m = BSC_ET(D, H, Hp, gamma)
    an = _An(T=1.0)
    data = m.select_Hprimes(params, {"y": Y})
    ss = m.E_step(an, params, data)
    cand = data["candidates"].tensor.long()
    logpj = ss["logpj"].tensor
    assert logpj.shape == (N, 411)

    # (1) candidates are the H' best normalised scores, ascending
    Wt = torch.from_numpy(np.ascontiguousarray(W0.T)).to(dev)
    sim = (Y @ Wt.t()) / Wt.norm(dim=1)[None, :]
    top = torch.topk(sim, Hp, dim=1).indices.flip(1)
    assert (top == cand).float().mean().item() > 0.9999
    # (2) lse is the log-sum-exp of the row
    torch.testing.assert_close(ss["logpj"].lse, torch.logsumexp(logpj, dim=1), rtol=1e-12, atol=1e-10)
    # (3) singleton columns equal the direct energies for a sample of rows
    rows = torch.arange(0, N, 997, device=dev)
    e_direct = ((Wt[None, :, :] - Y[rows][:, None, :]) ** 2).sum(-1)
    pil = np.log(params["pi"] / (1 - params["pi"]))
    torch.testing.assert_close(logpj[rows, 1:H + 1], pil - 0.5 * e_direct, rtol=1e-10, atol=1e-8)
    # (4) statistics are additive over shards (what the all-reduce relies on)
    new_all = m.M_step(an, params, ss, data)
    stats_all = m._ws["stats"].clone()
    half = N // 2 + 3
    acc = torch.zeros_like(stats_all)
    for sl in (slice(0, half), slice(half, N)):
        m2 = BSC_ET(D, H, Hp, gamma)
        d2 = m2.select_Hprimes(params, {"y": Y[sl].contiguous()})
        s2 = m2.E_step(an, params, d2)
        m2.M_step(an, params, s2, d2)
        acc += m2._ws["stats"]
    torch.testing.assert_close(acc, stats_all, rtol=1e-9, atol=1e-9)
    # (5) E[s] rows: sum_h E[s_h] = E|s| and the posterior is normalised
    q = torch.exp(logpj - ss["logpj"].lse[:, None])
    torch.testing.assert_close(q.sum(1), torch.ones(N, dtype=torch.float64, device=dev), rtol=1e-12, atol=1e-12)
    assert np.isfinite(new_all["W"]).all() and 0 < new_all["pi"] < 1 and new_all["sigma"] > 0
    # a well-initialised step moves W towards the ground truth
    assert np.abs(new_all["W"] - W_gt.cpu().numpy()).mean() < np.abs(W0 - W_gt.cpu().numpy()).mean()


# ------------------------------------------------------------------------- inference
@pytest.mark.parametrize("tag,kw", [("plain", dict(topK=5, adaptive=False)), ("adaptive", dict(topK=4, adaptive=True)),
                                    ("capped", dict(topK=3, adaptive=True, Hprime_max=5, gamma_max=3, logprob=True))])
def test_inference_matches_reference(dev, tag, kw, capsys):
    """CAModel.inference (camodels/__init__.py:256-375): top-K states, marginals, adaptive H'/gamma growth."""
    from prosper_amd.em.annealing import LinearAnnealing
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    g = golden("bsc_inference.npz")
    m = BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    anneal = LinearAnnealing(1)
    anneal["T"] = [(0, 1.)]
    anneal["anneal_prior"] = False
    res = m.inference(anneal, {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}, {"y": g["y"]}, **kw)
    assert (m.Hprime, m.gamma, m.no_states) == (int(g["Hprime"]), int(g["gamma"]), 6)
    assert np.array_equal(res["gamma"], g[tag + "_gamma"]) and np.array_equal(res["Hprime"], g[tag + "_Hprime"])
    assert res["s"].dtype == np.int8 and np.array_equal(res["s"], g[tag + "_s"])
    np.testing.assert_allclose(res["p"], g[tag + "_p"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(res["m"], g[tag + "_m"], rtol=1e-8, atol=1e-12)


# ------------------------------------------------------------------------- chunked pipeline
def test_inference_refuses_nan_log_joints(dev):
    """pm_infer_topk_f64 skips NaN log-joints and reports index -1 when fewer than topK comparable entries exist (round-3
    advisor finding: the caller indexed with it unchecked).  `inference` raises instead of returning a wrapped state; rows
    with -inf entries but enough finite ones are fine."""
    from prosper_amd import _lib
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 16, 8, 3, 2, 12
    rng = np.random.RandomState(4)
    W = rng.normal(size=(D, H))
    y = rng.normal(size=(N, D))
    m = BSC_ET(D, H, Hp, gamma)
    params = {"W": W, "pi": 0.2, "sigma": 1.0}
    K = 1 + H + m.no_states
    res = m.inference(_An(T=1.0), params, {"y": y}, topK=K, adaptive=False)          # k_eff == K: every column is used
    assert res["s"].shape == (N, K, H) and np.isfinite(res["p"]).all()
    y_bad = y.copy()
    y_bad[5, 3] = np.nan
    with pytest.raises(_lib.HipError):
        BSC_ET(D, H, Hp, gamma).inference(_An(T=1.0), params, {"y": y_bad}, topK=4, adaptive=False)


@pytest.mark.parametrize("overlap", [False, True])
def test_chunked_pipeline_matches_whole_shard(dev, overlap):
    """Shards beyond ``max_chunk_rows`` (or ``PM_CHUNK_ROUNDS``) go through the E-step in chunks of whole GEMM
    rounds with two alternating score buffers, optionally with the next chunk's GEMM on a side stream: same
    candidates, same log-joints as the one-buffer whole-shard mode."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    D, H, Hp, gamma, N = 32, 256, 3, 2, 70000          # one round of resident tiles = 32768 rows at H = 256
    gen = torch.Generator(device=dev).manual_seed(7)
    W_gt = torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)
    S = (torch.rand(N, H, generator=gen, device=dev) < 2.0 / H).to(torch.float64)
    Y = S @ W_gt.t() + torch.randn(N, D, generator=gen, device=dev, dtype=torch.float64)
    params = {"W": (W_gt + 0.1 * torch.randn(D, H, generator=gen, device=dev, dtype=torch.float64)).cpu().numpy(),
              "pi": 2.0 / H, "sigma": 1.0}
    an = _An(T=1.0)
    ref = BSC_ET(D, H, Hp, gamma)
    assert ref._whole_shard(N)
    d0 = ref.select_Hprimes(params, {"y": Y})
    s0 = ref.E_step(an, params, d0)
    m = BSC_ET(D, H, Hp, gamma)
    m.chunk_rounds, m.overlap_streams = 1, overlap
    assert not m._whole_shard(N) and m._chunk_rows(N) == 32768
    d1 = m.select_Hprimes(params, {"y": Y})
    s1 = m.E_step(an, params, d1)
    assert torch.equal(d0["candidates"].tensor, d1["candidates"].tensor)
    torch.testing.assert_close(s1["logpj"].tensor, s0["logpj"].tensor, rtol=1e-12, atol=1e-10)
    torch.testing.assert_close(s1["logpj"].lse, s0["logpj"].lse, rtol=1e-12, atol=1e-10)
    new0, new1 = ref.M_step(an, params, s0, d0), m.M_step(an, params, s1, d1)
    for k in ("W", "pi", "sigma"):
        np.testing.assert_allclose(new1[k], new0[k], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("gate", [0.0, 3.0])
def test_expand_lists_rebuilds_only_listed_rows_and_only_behind_the_gate(dev, gate):
    """pm_bsc_expand_lists_gated_f64: rows whose list is complete are rebuilt from the list, rows marked PM_BSC_NZ_OVERFLOW keep
    the dense row the E-step pass stored, and nothing is touched while the gate is 0."""
    from prosper_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(5)
    N, H = 1000, 200
    idx = np.full((N, 16), 0xFFFF, dtype=np.uint16)
    val = np.zeros((N, 16))
    ref = rng.normal(size=(N, H))                                    # what the buffer holds before (stale / stored rows)
    want = ref.copy()
    over = rng.random_sample(N) < 0.2
    for n in range(N):
        k = rng.randint(0, 17)
        hs = rng.choice(H, size=k, replace=False)
        idx[n, :k], val[n, :k] = hs, rng.normal(size=k)
        if over[n]:
            idx[n, 0] = 0xFFFE
        elif gate:
            want[n] = 0.0
            want[n, hs] = val[n, :k]
    d_idx = torch.from_numpy(idx.view(np.int16)).to(dev)
    d_val, d_E = torch.from_numpy(val).to(dev), torch.from_numpy(ref.copy()).to(dev)
    d_gate = torch.tensor([gate], dtype=torch.float64, device=dev)
    rc = lib.pm_bsc_expand_lists_gated_f64(_p(d_idx), _p(d_val), _p(d_E), H, N, H, _p(d_gate), _stream())
    assert rc == 0
    assert np.array_equal(d_E.cpu().numpy(), want)
