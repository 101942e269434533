"""ORACLE -- test infrastructure, not product code.

CPU restatement (NumPy, float64) of the reference's signed Maximum-Magnitude Causes Analysis
truncated-EM hot path, prosper/em/camodels/mmca_et.py (reference v0.1.0).  Imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

  *_loop : per-datapoint loops following the reference line by line
  *_vec  : vectorised algebra the HIP kernels implement

Pinned against outputs of the reference itself (tests/golden/mmca_step_*.npz, minted by
tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
from math import pi as _PI

import numpy as np

from .bsc_oracle import generate_state_matrix, Anneal, make_model  # noqa: F401  (shared with BSC)
from .mca_oracle import pi_gamma_factors as _mca_factors

RHO_T_BOUND = 1.20      # mmca_et.py:37
RHO_LBOUND = 1          # mmca_et.py:38
RHO_UBOUND = 35         # mmca_et.py:39
TOL = 1e-4              # mmca_et.py:40
INERTIA_ALPHA = 2.5     # mmca_et.py:385


def check_params(params):
    """mmca_et.py:50-63: |W| >= tol, sign kept (zeros become +tol, then the second rule sees +tol)."""
    params = dict(params)
    W = np.array(params['W'], dtype=np.float64, copy=True)
    W[np.logical_and(W >= 0., W < +TOL)] = +TOL
    W[np.logical_and(W <= 0., W > -TOL)] = -TOL
    params['W'] = W
    return params


def rho_of(T):
    """mmca_et.py:158-161."""
    T_rho = np.maximum(T, RHO_T_BOUND)
    rho = 1. / (1. - 1. / T_rho)
    return np.maximum(np.minimum(rho, RHO_UBOUND), RHO_LBOUND)


def pi_gamma_factors(pies, H, gamma):
    """mmca_et.py:266-271 (scipy comb without exact=1: float; same values for these sizes)."""
    return _mca_factors(pies, H, gamma)


def generate_from_hidden(W_DH, s):
    """mmca_et.py:66-93 without the noise draw: per dimension the active cause of largest magnitude."""
    W = W_DH.T
    N, D = s.shape[0], W.shape[1]
    y = np.zeros((N, D))
    for n in range(N):
        t0 = s[n, :, None] * W
        idx = np.argmax(np.abs(t0), axis=0)
        y[n] = t0[idx].diagonal()
    return y


# ------------------------------------------------------------------------------------- select
def select_hprimes_loop(W_DH, Y, Hprime):
    """mmca_et.py:96-124: the Hprime latents with the SMALLEST |W_h - y|^2, ascending."""
    W = W_DH.T
    N = Y.shape[0]
    cand = np.zeros((N, Hprime), dtype=np.int64)
    for n in range(N):
        sim = ((W - Y[n]) ** 2).sum(axis=1)
        cand[n] = np.argsort(sim)[0:Hprime]
    return cand


def select_scores_vec(W_DH, Y):
    """|W_h|^2 - 2 <W_h, y> (+ |y|^2, which does not change the ranking): the Gram form the device ranks."""
    W = W_DH.T
    return (W * W).sum(axis=1)[None, :] - 2. * (Y @ W.T)


def select_hprimes_vec(W_DH, Y, Hprime):
    return np.argsort(select_scores_vec(W_DH, Y), axis=1, kind='stable')[:, :Hprime].astype(np.int64)


# ------------------------------------------------------------------------------------- E-step
def e_step_loop(anneal, W_DH, pies, sigma, Y, cand, SM, state_abs):
    """mmca_et.py:126-199 -> logpj (N, 1+H+S).  No beta here (applied in the M-step)."""
    W = W_DH.T
    H = W.shape[0]
    N = Y.shape[0]
    rho = rho_of(anneal['T'])
    pre1 = -1. / 2. / sigma / sigma
    pil_bar = np.log(pies / (1. - pies))
    Wrhos = np.sign(W) * np.exp(rho * np.log(np.abs(W)))
    F = np.empty([N, 1 + H + SM.shape[0]])
    with np.errstate(divide='ignore', under='ignore'):
        for n in range(N):
            y = Y[n, :]
            F[n, 0] = pre1 * (y ** 2).sum()
            F[n, 1:H + 1] = pil_bar + pre1 * ((W - y) ** 2).sum(axis=1)
            t0 = np.dot(SM, Wrhos[cand[n]])
            Wbar = np.sign(t0) * np.exp(np.log(np.abs(t0)) / rho)
            F[n, 1 + H:] = pil_bar * state_abs + pre1 * ((Wbar - y) ** 2).sum(axis=1)
    return F


def e_step_vec(anneal, W_DH, pies, sigma, Y, cand, SM, state_abs):
    W = W_DH.T
    rho = rho_of(anneal['T'])
    pre1 = -1. / 2. / sigma / sigma
    pil_bar = np.log(pies / (1. - pies))
    Wrhos = np.sign(W) * np.exp(rho * np.log(np.abs(W)))
    yn = (Y * Y).sum(axis=1)
    e1 = (W * W).sum(axis=1)[None, :] - 2. * (Y @ W.T) + yn[:, None]
    t0 = np.einsum('sj,njd->nsd', SM.astype(np.float64), Wrhos[cand])
    with np.errstate(divide='ignore', under='ignore'):
        Wbar = np.sign(t0) * np.exp(np.log(np.abs(t0)) / rho)
    es = ((Wbar - Y[:, None, :]) ** 2).sum(axis=2)
    return np.concatenate([pre1 * yn[:, None], pil_bar + pre1 * e1, pil_bar * state_abs[None, :] + pre1 * es], axis=1)


# ------------------------------------------------------------------------------------- M-step
def m_step(anneal, model, W_DH, pies, sigma, Y, cand, logpj, to_learn=('W', 'pi', 'sigma'), vec=False):
    """mmca_et.py:201-427 -> (params incl. 'Q', log dict)."""
    H, gamma, SM = model['H'], model['gamma'], model['SM']
    state_abs = SM.sum(axis=1)
    W = W_DH.T
    D = W.shape[1]
    N = Y.shape[0]
    T = anneal['T']
    rho = rho_of(T)
    beta = 1. / T
    pil_bar = np.log(pies / (1. - pies))
    Wl = np.log(np.abs(W))
    Wrho = np.exp(rho * Wl)
    Wrhos = np.sign(W) * Wrho

    my_corr = beta * logpj.max(axis=1)
    my_pjb = np.exp(beta * logpj - my_corr[:, None])
    A_pg, B_pg = pi_gamma_factors(pies, H, gamma)

    if anneal['Ncut_factor'] > 0.0:
        my_logdenoms = np.log(my_pjb.sum(axis=1)) + my_corr
        N_use = int(N * (1 - (1 - A_pg) * anneal['Ncut_factor']))
        cut_denom = np.sort(my_logdenoms, kind='mergesort')[-N_use]
        which = np.array(my_logdenoms >= cut_denom)
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
    with np.errstate(divide='ignore', under='ignore', invalid='ignore'):
        if not vec:
            for n in range(my_N):
                y, c, lp, pjb, corr = Y[n], cand[n], logpj[n], my_pjb[n], my_corr[n]
                logpjb = beta * lp - corr
                this_Wp = np.zeros_like(W)
                this_Wq = np.zeros_like(W)
                this_sigma = pjb[0] * (y ** 2).sum()
                this_Wp += pjb[1:(H + 1), None] * y[None, :]
                this_Wq += pjb[1:(H + 1), None]
                this_pi = pjb[1:(H + 1)].sum()
                this_sigma += (pjb[1:(H + 1)] * ((W - y) ** 2).sum(axis=1)).sum()
                Wl_ = Wl[c]
                t0 = np.dot(SM, Wrhos[c])
                Wlbar = np.log(np.abs(t0)) / rho
                Wbar = np.sign(t0) * np.exp(Wlbar)
                t = np.maximum(Wlbar[:, None, :] - Wl_[None, :, :], 0.)
                Aid = (SM[:, :, None] * np.exp(logpjb[H + 1:, None, None] - (rho - 1) * t)).sum(axis=0)
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
            t0 = np.einsum('sj,njd->nsd', SMf, Wrhos[cand])                      # (n, S, D)
            aT = np.abs(t0)
            Wbar = np.exp(np.log(aT) / rho)                                      # |Wbar|
            # min(1, (|W_j| / |Wbar_s|)^(rho-1)) with (.)^(rho-1) = |W_j|^(rho-1) |Wbar_s| / |t0_s|
            Wrm1 = np.exp((rho - 1) * Wl)
            r = np.where(aT > 0, Wbar / np.where(aT > 0, aT, 1.0), np.inf)       # (n, S, D)
            fac = np.minimum(1.0, r[:, :, None, :] * Wrm1[cand][:, None, :, :])  # (n, S, H', D)
            Aid = np.einsum('ns,sj,nsjd->njd', qs, SMf, fac)
            my_Wp = q1.T @ Y
            my_Wq = np.repeat(q1.sum(axis=0)[:, None], D, axis=1)
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
        Wq[Wq < TOL] = TOL                                         # mmca_et.py:378-379
        W_new = Wp / Wq
        inertia = np.maximum(1. - np.exp(-Wq / INERTIA_ALPHA), 0.2)
        W_new = (inertia * W_new + (1 - inertia) * W).T
    else:
        W_new = W.T
    pi_new = A_pg / B_pg * pies * my_pi / N_use if 'pi' in to_learn else pies
    sigma_new = np.sqrt(my_sigma / D / N_use) if 'sigma' in to_learn else sigma
    lAi = (H * np.log(1. - pi_new)) - ((D / 2) * np.log(2 * _PI)) - (D * np.log(sigma_new))
    Q = (lAi * N_use) + ldenom_sum
    params = {'W': W_new, 'pi': pi_new, 'sigma': sigma_new, 'Q': Q}
    return params, {'N_use': N_use, 'stats': {'Wp': my_Wp, 'Wq': my_Wq, 'pi': my_pi, 'sigma': my_sigma}}


def em_step(anneal, model, params, Y, vec=True):
    """check_params -> select -> E -> M (camodels/__init__.py:163-193 for MMCA_ET)."""
    params = check_params(params)
    SM = model['SM']
    sel = select_hprimes_vec if vec else select_hprimes_loop
    est = e_step_vec if vec else e_step_loop
    cand = sel(params['W'], Y, model['Hprime'])
    logpj = est(anneal, params['W'], params['pi'], params['sigma'], Y, cand, SM, SM.sum(axis=1))
    new, log = m_step(anneal, model, params['W'], params['pi'], params['sigma'], Y, cand, logpj, vec=vec)
    log['candidates'], log['logpj'] = cand, logpj
    return new, log
