"""ORACLE -- test infrastructure, not product code.

CPU restatement (NumPy, float64) of the reference's Binary-Sparse-Coding truncated-EM
hot path, prosper/em/camodels/bsc_et.py + camodels/__init__.py (reference v0.1.0).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker; nothing under prosper_amd/ imports it.

Two flavours of every stage:
  *_loop  -- per-datapoint loops that follow the reference line by line, including its
             redundant work (np.inner(W, W) per datapoint in the selection, H x D
             temporaries per datapoint in the M-step).  This IS the reference algorithm
             and is what bench.py times as cpu_baseline (kind "port").
  *_vec   -- the GEMM + Gram-matrix algebra the HIP kernels implement; used to check
             larger cases in seconds.

Parity pinning: the reference has no tests or golden vectors for this path.
The oracle is pinned against outputs of the reference itself, generated in the build
container by tests/golden/make_golden.py (reference imported from /root/reference via
single-rank mpi4py / tables shims) and committed as tests/golden/*.npz; see
tests/test_oracle_golden.py.
"""
from itertools import combinations
from math import pi as _PI

import numpy as np
from scipy.special import comb


# --------------------------------------------------------------------------- a1
def generate_state_matrix(Hprime, gamma):
    """camodels/__init__.py:21-47 -- rows = combinations(range(Hprime), g), g = 2..gamma."""
    sl = []
    for g in range(2, gamma + 1):
        for s in combinations(list(range(Hprime)), g):
            sl.append(np.array(s, dtype=np.int8))
    sm = np.zeros((len(sl), Hprime), dtype=np.uint8)
    for i, s in enumerate(sl):
        sm[i, s] = 1
    return sl, len(sl), sm, sm.sum(axis=1)


class Anneal(dict):
    """Stand-in for LinearAnnealing at one fixed step: unknown keys read 0.0
    (annealing.py:93-94)."""

    def __missing__(self, key):
        return 0.0


# --------------------------------------------------------------------------- a3
def select_hprimes_loop(W_DH, Y, Hprime):
    """bsc_et.py:98-115.  W_DH is model_params['W'] (D, H); returns (N, Hprime) int64,
    ascending by similarity (last = best)."""
    W = W_DH.T
    my_N = Y.shape[0]
    candidates = np.zeros((my_N, Hprime), dtype=np.int64)
    for n in range(my_N):
        sim = np.inner(W, Y[n]) / np.sqrt(np.diag(np.inner(W, W))) / np.sqrt(np.inner(Y[n], Y[n]))
        candidates[n] = np.argsort(sim)[-Hprime:]
    return candidates


def select_hprimes_vec(W_DH, Y, Hprime):
    """Same ranking from one N x D . D x H GEMM."""
    W = W_DH.T
    A = Y @ W.T
    sim = A / np.sqrt((W * W).sum(axis=1))[None, :] / np.sqrt((Y * Y).sum(axis=1))[:, None]
    return np.argsort(sim, axis=1)[:, -Hprime:].astype(np.int64)


# --------------------------------------------------------------------------- a4
def _prior_row(pies, H, state_abs):
    pil_bar = np.log(pies / (1. - pies))
    return np.concatenate(([0.], np.full(H, pil_bar), pil_bar * state_abs))


def e_step_loop(anneal, W_DH, pies, sigma, mu, Y, candidates, SM, state_abs):
    """bsc_et.py:119-192 -> logpj (N, 1+H+S)."""
    W = W_DH.T
    H, D = W.shape
    my_N = Y.shape[0]
    beta = 1. / anneal['T']
    pre1 = -1. / 2. / sigma / sigma
    F = np.empty([my_N, 1 + H + SM.shape[0]])
    pre_F = np.tile(_prior_row(pies, H, state_abs), (my_N, 1))
    for n in range(my_N):
        y = Y[n, :] - mu
        cand = candidates[n, :]
        F[n, 0] = pre1 * (y ** 2).sum()
        F[n, 1:H + 1] = pre1 * ((W - y) ** 2).sum(axis=1)
        Wbar = np.dot(SM, W[cand])
        F[n, 1 + H:] = pre1 * ((Wbar - y) ** 2).sum(axis=1)
    if anneal['anneal_prior']:
        return beta * (pre_F + F)
    return pre_F + beta * F


def energies_vec(W_DH, mu, Y, candidates, SM):
    """Squared reconstruction errors of every truncated state via scores + Gram:
    e_0 = |y|^2, e_h = |W_h|^2 - 2 a_h + |y|^2,
    e_s = |y|^2 - 2 sum_{j in s} a_{c_j} + sum_{j,j' in s} G_{c_j c_j'}   (SURVEY 8a/a4)."""
    W = W_DH.T
    Yc = Y - mu
    A = Yc @ W.T                                   # (N, H)
    G = W @ W.T                                    # (H, H)
    yn = (Yc * Yc).sum(axis=1)
    e0 = yn[:, None]
    e1 = np.diag(G)[None, :] - 2. * A + yn[:, None]
    SMf = SM.astype(np.float64)
    Ac = np.take_along_axis(A, candidates, axis=1)                 # (N, H')
    Gc = G[candidates[:, :, None], candidates[:, None, :]]         # (N, H', H')
    lin = Ac @ SMf.T                                               # (N, S)
    quad = np.einsum('si,nij,sj->ns', SMf, Gc, SMf)
    es = yn[:, None] - 2. * lin + quad
    return np.concatenate([e0, e1, es], axis=1)


def e_step_vec(anneal, W_DH, pies, sigma, mu, Y, candidates, SM, state_abs):
    H = W_DH.shape[1]
    beta = 1. / anneal['T']
    pre1 = -1. / 2. / sigma / sigma
    F = pre1 * energies_vec(W_DH, mu, Y, candidates, SM)
    pre_F = _prior_row(pies, H, state_abs)[None, :]
    if anneal['anneal_prior']:
        return beta * (pre_F + F)
    return pre_F + beta * F


# --------------------------------------------------------------------------- a5
def pi_gamma_factors(pies, H, gamma):
    """bsc_et.py:237-244 -> (A_pi_gamma, B_pi_gamma, E_pi_gamma)."""
    A = 0
    B = 0
    for g in range(gamma + 1):
        t = comb(H, g) * (pies ** g) * ((1 - pies) ** (H - g))
        A += t
        B += g * t
    return A, B, pies * H * A / B


def truncate(anneal, all_denoms, A_pi_gamma, N, all_denoms_global=None):
    """bsc_et.py:246-260 -> (keep mask for the local rows, cut value or None).
    ``all_denoms_global``: concatenation over shards (what parallel.allsort sees)."""
    if anneal['Ncut_factor'] > 0.0:
        N_use = int(N * (1 - (1 - A_pi_gamma) * anneal['Ncut_factor']))
        pool = all_denoms if all_denoms_global is None else all_denoms_global
        cut_denom = np.sort(pool, kind='mergesort')[-N_use]
        return np.array(all_denoms >= cut_denom), cut_denom
    return np.ones(all_denoms.shape[0], dtype=bool), None


def m_step_stats_loop(W_DH, mu, Y, candidates, logpj_all, SM, learn_sigma=True):
    """The per-datapoint accumulation loops of bsc_et.py:334-366 (W, pi, mu statistics)
    and :395-415 (sigma statistic) over the rows handed in (already truncated)."""
    W = W_DH.T
    H, D = W.shape
    my_N = Y.shape[0]
    corr_all = logpj_all.max(axis=1) if my_N else np.zeros(0)
    pjb_all = np.exp(logpj_all - corr_all[:, None])
    my_Wp = np.zeros_like(W)
    my_Wq = np.zeros((H, H))
    my_pi = 0.0
    my_sigma = 0.0
    my_mus = np.zeros(H)
    for n in range(my_N):
        y = Y[n, :] - mu
        cand = candidates[n, :]
        pjb = pjb_all[n, :]
        this_Wp = np.outer(pjb[1:(H + 1)], y)
        this_Wq = pjb[1:(H + 1)] * np.identity(H)
        this_pi = pjb[1:(H + 1)].sum()
        this_mus = pjb[1:(H + 1)].copy()
        this_Wp[cand] += np.dot(np.outer(y, pjb[(1 + H):]), SM).T
        this_Wq_tmp = np.zeros_like(my_Wq[cand])
        this_Wq_tmp[:, cand] = np.dot(pjb[(1 + H):] * SM.T, SM)
        this_Wq[cand] += this_Wq_tmp
        this_pi += np.inner(pjb[(1 + H):], SM.sum(axis=1))
        this_mus[cand] += np.inner(SM.T, pjb[(1 + H):])
        denom = pjb.sum()
        my_Wp += this_Wp / denom
        my_Wq += this_Wq / denom
        my_pi += this_pi / denom
        my_mus += this_mus / denom
    if learn_sigma:
        for n in range(my_N):
            y = Y[n, :] - mu
            cand = candidates[n, :]
            logpj = logpj_all[n, :]
            pjb = np.exp(logpj - logpj.max())
            this_sigma = pjb[0] * (y ** 2).sum()
            this_sigma += (pjb[1:(H + 1)] * ((W - y) ** 2).sum(axis=1)).sum()
            Wbar = np.dot(SM, W[cand])
            this_sigma += (pjb[(H + 1):] * ((Wbar - y) ** 2).sum(axis=1)).sum()
            my_sigma += this_sigma / pjb.sum()
    return {'Wp': my_Wp, 'Wq': my_Wq, 'pi': my_pi, 'sigma': my_sigma, 'mus': my_mus,
            'data_sum': Y.sum(axis=0)}


def m_step_stats_vec(W_DH, mu, Y, candidates, logpj_all, SM):
    """Same statistics as dense algebra: Wp = E[s]^T Y, Wq = sum_n E[s s^T]_n,
    sigma statistic = sum_nk q_nk e_nk."""
    H = W_DH.shape[1]
    N = Y.shape[0]
    SMf = SM.astype(np.float64)
    q = np.exp(logpj_all - logpj_all.max(axis=1, keepdims=True)) if N else np.zeros_like(logpj_all)
    q = q / q.sum(axis=1, keepdims=True)
    q1, qs = q[:, 1:H + 1], q[:, H + 1:]
    rows = np.arange(N)[:, None]
    Es = q1.copy()
    np.add.at(Es, (rows, candidates), qs @ SMf)
    Yc = Y - mu
    Wp = Es.T @ Yc
    Wq = np.diag(q1.sum(axis=0))
    M = np.einsum('ns,si,sj->nij', qs, SMf, SMf)                   # (N, H', H')
    np.add.at(Wq, (candidates[:, :, None], candidates[:, None, :]), M)
    my_pi = q1.sum() + (qs @ SMf.sum(axis=1)).sum()
    e = energies_vec(W_DH, mu, Y, candidates, SM)
    return {'Wp': Wp, 'Wq': Wq, 'pi': my_pi, 'sigma': (q * e).sum(), 'mus': Es.sum(axis=0),
            'data_sum': Y.sum(axis=0)}


def reference_rcond():
    """The singular-value cutoff the reference hands to lstsq (bsc_et.py:377-380): ``None`` when
    ``float(np.__version__[2:]) >= 14.0`` (NumPy 1.14 .. 1.x), else -1 = machine precision -- which is what
    that expression yields under NumPy 2.x ('2.2.6'[2:] = '2.6').  Only matters for a rank-deficient Wq."""
    try:
        return None if float(np.__version__[2:]) >= 14.0 else -1
    except ValueError:
        return None


def m_step(anneal, model, W_DH, pies, sigma, mu, Y, candidates, logpj_all,
           to_learn=('W', 'pi', 'sigma'), stats_fn=m_step_stats_loop, shards=None):
    """bsc_et.py:195-438.  ``model`` = dict(H, gamma, SM).  ``shards`` (optional list of
    row-index arrays) emulates P MPI ranks: statistics are computed per shard and summed
    (the Allreduce calls of :225-417), the cut uses the global sort (:252).
    Returns (new_params dict, log dict with N, L, N_use, and the summed statistics)."""
    H, gamma, SM = model['H'], model['gamma'], model['SM']
    W = W_DH.T
    D = W.shape[1]
    all_denoms = np.exp(logpj_all).sum(axis=1)
    N = Y.shape[0]
    A_pg, B_pg, E_pg = pi_gamma_factors(pies, H, gamma)
    keep, cut = truncate(anneal, all_denoms, A_pg, N)
    N_use = int(keep.sum())
    Yk, ck, lk = Y[keep], candidates[keep], logpj_all[keep]

    L = H * np.log(1 - pies) - 0.5 * D * np.log(2 * _PI * sigma ** 2) - np.log(A_pg)
    Fs = np.log(np.exp(lk).sum(axis=1)).sum()
    L += Fs / N_use

    if shards is None:
        shards = [np.arange(Yk.shape[0])]
    else:  # shards index the ORIGINAL rows; map to kept rows
        pos = -np.ones(N, dtype=np.int64)
        pos[np.where(keep)[0]] = np.arange(N_use)
        shards = [pos[s][pos[s] >= 0] for s in shards]
    tot = None
    for idx in shards:
        st = stats_fn(W_DH, mu, Yk[idx], ck[idx], lk[idx], SM)
        tot = st if tot is None else {k: tot[k] + st[k] for k in tot}

    if 'W' in to_learn:
        W_new = np.linalg.lstsq(tot['Wq'], tot['Wp'], rcond=reference_rcond())[0]
    else:
        W_new = W
    pi_new = E_pg * tot['pi'] / H / N_use if 'pi' in to_learn else pies
    sigma_new = np.sqrt(tot['sigma'] / D / N_use) if 'sigma' in to_learn else sigma
    if 'mu' in to_learn:
        my_N = Yk.shape[0]
        mu_new = tot['data_sum'] / my_N - np.inner(W_new.T / my_N, tot['mus'])
    else:
        mu_new = mu
    params = {'W': W_new.T, 'pi': pi_new, 'sigma': sigma_new, 'mu': mu_new}
    log = {'N': N_use, 'L': L, 'N_use': N_use, 'Fs': Fs, 'cut': cut, 'keep': keep, 'stats': tot}
    return params, log


# --------------------------------------------------------------------------- a6
def em_step(anneal, model, params, Y, stats_fn=m_step_stats_loop, vec=False):
    """One CAModel.step without noise / partial data (camodels/__init__.py:163-193)."""
    Hp, SM = model['Hprime'], model['SM']
    state_abs = SM.sum(axis=1)
    mu = params.get('mu', np.zeros(Y.shape[1]))
    if vec:
        cand = select_hprimes_vec(params['W'], Y, Hp)
        logpj = e_step_vec(anneal, params['W'], params['pi'], params['sigma'], mu, Y, cand, SM, state_abs)
    else:
        cand = select_hprimes_loop(params['W'], Y, Hp)
        logpj = e_step_loop(anneal, params['W'], params['pi'], params['sigma'], mu, Y, cand, SM, state_abs)
    new, log = m_step(anneal, model, params['W'], params['pi'], params['sigma'], mu, Y, cand, logpj,
                      stats_fn=stats_fn)
    log['candidates'] = cand
    log['logpj'] = logpj
    return new, log


def make_model(D, H, Hprime, gamma):
    sl, S, SM, state_abs = generate_state_matrix(Hprime, gamma)
    return {'D': D, 'H': H, 'Hprime': Hprime, 'gamma': gamma, 'SM': SM, 'state_abs': state_abs, 'S': S}


# --------------------------------------------------------------------------- a9
def standard_init(Y, H, rng):
    """camodels/__init__.py:196-235 with an explicit RandomState."""
    D = Y.shape[1]
    W_mean = Y.sum(axis=0) / Y.shape[0]
    sigma_sq = ((Y - W_mean) ** 2).sum(axis=0) / Y.shape[0]
    sigma_init = np.sqrt(sigma_sq).sum() / D
    W_init = W_mean[:, None] + rng.normal(scale=sigma_init / 4., size=[D, H])
    return {'W': W_init, 'pi': 1. / H, 'sigma': sigma_init}


def generate_bsc_data(W_DH, pies, sigma, N, rng):
    """camodels/__init__.py:104-122 + bsc_et.py:67-95 with an explicit RandomState:
    latents from one random((N, H)) draw, then y = s W^T + N(0, sigma^2)."""
    D, H = W_DH.shape
    s = rng.random_sample(size=(N, H)) < pies
    y = s.astype(np.float64) @ W_DH.T
    y += rng.normal(scale=sigma, size=(N, D))
    return y, s
