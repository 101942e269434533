"""ORACLE -- test infrastructure, not product code.

CPU restatement (NumPy, float64) of the reference's Maximal Causes Analysis truncated-EM hot
path, prosper/em/camodels/mca_et.py (reference v0.1.0).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

  *_loop : per-datapoint loops following the reference line by line (the timed CPU baseline)
  *_vec  : vectorised algebra the HIP kernels implement

Pinned against outputs of the reference itself (tests/golden/mca_step_*.npz, minted by
tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
from math import pi as _PI

import numpy as np
from scipy.special import comb

from .bsc_oracle import generate_state_matrix, Anneal, make_model  # noqa: F401  (shared with BSC)

RHO_TEMP_BOUND = 1.05   # mca_et.py:31
W_TOL = 1e-4            # mca_et.py:32


def check_params(params):
    """mca_et.py:44-55: W is clamped to >= W_tol."""
    params = dict(params)
    params['W'] = np.maximum(params['W'], W_TOL)
    return params


def rho_of(T):
    """mca_et.py:142-144: rho = 1 / (1 - 1/max(T, 1.05))."""
    return 1. / (1. - 1. / np.maximum(T, RHO_TEMP_BOUND))


# ------------------------------------------------------------------------------------- select
def select_hprimes_loop(W_DH, Y, Hprime):
    """mca_et.py:88-111: the Hprime latents with the SMALLEST sum_d |max(W_hd, y_d) - y_d|,
    ascending (first = best)."""
    W = W_DH.T
    N = Y.shape[0]
    cand = np.zeros((N, Hprime), dtype=np.int64)
    for n in range(N):
        W_interm = np.maximum(W, Y[n])
        sim = np.abs(W_interm - Y[n]).sum(axis=1)
        cand[n] = np.argsort(sim)[0:Hprime]
    return cand


def select_scores_vec(W_DH, Y):
    W = W_DH.T
    return np.maximum(W[None, :, :] - Y[:, None, :], 0.0).sum(axis=2)      # (N, H)


def select_hprimes_vec(W_DH, Y, Hprime):
    return np.argsort(select_scores_vec(W_DH, Y), axis=1)[:, :Hprime].astype(np.int64)


# ------------------------------------------------------------------------------------- E-step
def e_step_loop(anneal, W_DH, pies, sigma, Y, cand, SM, state_abs):
    """mca_et.py:114-179 -> logpj (N, 1+H+S).  No beta here (applied in the M-step)."""
    W = W_DH.T
    H = W.shape[0]
    N = Y.shape[0]
    rho = rho_of(anneal['T'])
    pre1 = -1. / 2. / sigma / sigma
    pil_bar = np.log(pies / (1. - pies))
    Wrho = np.exp(rho * np.log(W))
    F = np.empty([N, 1 + H + SM.shape[0]])
    for n in range(N):
        y = Y[n, :]
        F[n, 0] = pre1 * (y ** 2).sum()
        F[n, 1:H + 1] = pil_bar + pre1 * ((W - y) ** 2).sum(axis=1)
        Wbar = np.exp(np.log(np.dot(SM, Wrho[cand[n]])) / rho)
        F[n, 1 + H:] = pil_bar * state_abs + pre1 * ((Wbar - y) ** 2).sum(axis=1)
    return F


def e_step_vec(anneal, W_DH, pies, sigma, Y, cand, SM, state_abs):
    W = W_DH.T
    rho = rho_of(anneal['T'])
    pre1 = -1. / 2. / sigma / sigma
    pil_bar = np.log(pies / (1. - pies))
    Wrho = np.exp(rho * np.log(W))
    yn = (Y * Y).sum(axis=1)
    e1 = (W * W).sum(axis=1)[None, :] - 2. * (Y @ W.T) + yn[:, None]
    T = np.einsum('sj,njd->nsd', SM.astype(np.float64), Wrho[cand])          # (N, S, D)
    Wbar = np.exp(np.log(T) / rho)
    es = ((Wbar - Y[:, None, :]) ** 2).sum(axis=2)
    return np.concatenate([pre1 * yn[:, None], pil_bar + pre1 * e1, pil_bar * state_abs[None, :] + pre1 * es], axis=1)


# ------------------------------------------------------------------------------------- M-step
def pi_gamma_factors(pies, H, gamma):
    """mca_et.py:229-234 (comb(..., exact=1))."""
    A = 0.
    B = 0.
    for gp in range(0, gamma + 1):
        a = comb(H, gp, exact=1) * pies ** gp * (1. - pies) ** (H - gp)
        A += a
        B += gp * a
    return A, B


def m_step(anneal, model, W_DH, pies, sigma, Y, cand, logpj, to_learn=('W', 'pi', 'sigma'), vec=False):
    """mca_et.py:182-377 -> (params incl. 'Q', log dict)."""
    H, gamma, SM = model['H'], model['gamma'], model['SM']
    state_abs = SM.sum(axis=1)
    W = W_DH.T
    D = W.shape[1]
    N = Y.shape[0]
    T = anneal['T']
    rho = rho_of(T)
    beta = 1. / T
    pil_bar = np.log(pies / (1. - pies))
    Wl = np.log(W)
    Wrho = np.exp(rho * Wl)
    Wsquared = W * W

    my_corr = beta * logpj.max(axis=1)
    my_pjb = np.exp(beta * logpj - my_corr[:, None])
    A_pg, B_pg = pi_gamma_factors(pies, H, gamma)

    if anneal['Ncut_factor'] > 0.0:
        my_denoms = np.log(my_pjb.sum(axis=1)) + my_corr
        N_use = int(N * (1 - (1 - A_pg) * anneal['Ncut_factor']))
        cut_denom = np.sort(my_denoms, kind='mergesort')[-N_use]
        which = np.array(my_denoms >= cut_denom)
        Y, cand, logpj, my_pjb, my_corr = Y[which], cand[which], logpj[which], my_pjb[which], my_corr[which]
        N_use = Y.shape[0]
    else:
        N_use = N
    my_N = Y.shape[0]

    my_Wp = np.zeros_like(W)
    my_Wq = np.zeros_like(W)
    my_pi = 0.0
    my_sigma = 0.0
    ldenom_sum = 0.0
    if not vec:
        for n in range(my_N):
            y, c, lp, pjb, corr = Y[n], cand[n], logpj[n], my_pjb[n], my_corr[n]
            this_Wp = np.zeros_like(W)
            this_Wq = np.zeros_like(W)
            this_sigma = pjb[0] * (y ** 2).sum()
            this_Wp += (pjb[1:(H + 1), None] * Wsquared) * y[None, :]
            this_Wq += (pjb[1:(H + 1), None] * Wsquared)
            this_pi = pjb[1:(H + 1)].sum()
            this_sigma += (pjb[1:(H + 1)] * ((W - y) ** 2).sum(axis=1)).sum()
            Wl_, Wrho_ = Wl[c], Wrho[c]
            Wlrhom1 = (rho - 1) * Wl_
            Wlbar = np.log(np.dot(SM, Wrho_)) / rho
            Wbar = np.exp(Wlbar)
            blpj = beta * lp[1 + H:] - corr
            Aid = (SM[:, :, None] * np.exp(blpj[:, None, None] + (1 - rho) * Wlbar[:, None, :] + Wlrhom1[None, :, :])).sum(axis=0)
            this_Wp[c] += Aid * y[None, :]
            this_Wq[c] += Aid
            this_pi += (pjb[1 + H:] * state_abs).sum()
            this_sigma += (pjb[1 + H:] * ((Wbar - y) ** 2).sum(axis=1)).sum()
            denom = pjb.sum()
            my_Wp += this_Wp / denom
            my_Wq += this_Wq / denom
            my_pi += this_pi / denom
            my_sigma += this_sigma / denom
            ldenom_sum += np.log(np.sum(np.exp(lp)))
    else:
        q = my_pjb / my_pjb.sum(axis=1, keepdims=True)
        q1, qs = q[:, 1:H + 1], q[:, H + 1:]
        SMf = SM.astype(np.float64)
        Tsd = np.einsum('sj,njd->nsd', SMf, Wrho[cand])                        # (n, S, D)
        Wbar = np.exp(np.log(Tsd) / rho)
        # (W_j / Wbar_s)^(rho-1) = W_j^(rho-1) * Wbar_s / T_s
        Wrm1 = np.exp((rho - 1) * Wl)
        V = np.einsum('ns,sj,nsd->njd', qs, SMf, Wbar / Tsd)                   # (n, H', D)
        Aid = V * Wrm1[cand]
        my_Wp = (q1.T @ Y) * Wsquared
        my_Wq = q1.sum(axis=0)[:, None] * Wsquared
        np.add.at(my_Wp, cand, Aid * Y[:, None, :])
        np.add.at(my_Wq, cand, Aid)
        my_pi = q1.sum() + (qs @ state_abs).sum()
        pre1 = -1. / 2. / sigma / sigma
        prior = np.concatenate(([0.], np.full(H, pil_bar), pil_bar * state_abs))
        e = (logpj - prior[None, :]) / pre1
        my_sigma = (q * e).sum()
        ldenom_sum = np.log(np.exp(logpj).sum(axis=1)).sum()

    if 'W' in to_learn:
        Wp, Wq = my_Wp.copy(), my_Wq.copy()
        tiny = np.finfo(Wq.dtype).tiny
        Wp[Wq < tiny] = 0.
        Wq[Wq < tiny] = tiny
        W_new = (Wp / Wq).T
    else:
        W_new = W.T
    pi_new = A_pg / B_pg * pies * my_pi / N_use if 'pi' in to_learn else pies
    sigma_new = np.sqrt(my_sigma / D / N_use) if 'sigma' in to_learn else sigma
    lAi = (H * np.log(1. - pi_new)) - ((D / 2) * np.log(2 * _PI)) - (D * np.log(sigma_new))
    Q = (lAi * N_use) + ldenom_sum
    params = {'W': W_new, 'pi': pi_new, 'sigma': sigma_new, 'Q': Q}
    return params, {'N_use': N_use, 'stats': {'Wp': my_Wp, 'Wq': my_Wq, 'pi': my_pi, 'sigma': my_sigma}}


def em_step(anneal, model, params, Y, vec=True):
    """check_params -> select -> E -> M (camodels/__init__.py:163-193 for MCA_ET)."""
    params = check_params(params)
    SM = model['SM']
    sel = select_hprimes_vec if vec else select_hprimes_loop
    est = e_step_vec if vec else e_step_loop
    cand = sel(params['W'], Y, model['Hprime'])
    logpj = est(anneal, params['W'], params['pi'], params['sigma'], Y, cand, SM, SM.sum(axis=1))
    new, log = m_step(anneal, model, params['W'], params['pi'], params['sigma'], Y, cand, logpj, vec=vec)
    log['candidates'], log['logpj'] = cand, logpj
    return new, log


def generate_mca_data(W_DH, pies, sigma, N, rng):
    """mca_et.py:58-85 with an explicit RandomState: per datapoint one random(H) draw, max-rule
    superposition, then one normal((N, D)) noise draw."""
    D, H = W_DH.shape
    W = W_DH.T
    y = np.zeros((N, D))
    s = np.zeros((N, H), dtype=bool)
    for n in range(N):
        s[n] = rng.random_sample(H) < pies
        if s[n].any():
            y[n] = W[s[n]].max(axis=0).clip(min=0.0)
    y += rng.normal(scale=sigma, size=(N, D))
    return y, s
