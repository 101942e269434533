"""ORACLE -- test infrastructure, not product code.

CPU restatement (NumPy, float64) of the reference's Ternary Sparse Coding truncated-EM hot path,
prosper/em/camodels/tsc_et.py (reference v0.1.0): latents in {-1, 0, +1} with prior pi/2, 1-pi, pi/2,
linear superposition, Gaussian noise; every state lives in the H' candidate positions (null and one-cause
states included in the table, no global singleton block).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

The reference class cannot be constructed (``states`` undefined in tsc_et.py:131); its methods
do run on an object built without ``__init__`` -- that is how tests/golden/tsc_step_*.npz were minted
(tests/golden/make_golden.py:_make_tsc).  Two upstream behaviours matter for parity:
  * select_Hprimes takes the latents of the H' best one-cause STATES, so a latent can appear twice
    (both signs) among a datapoint's candidates (tsc_et.py:208-211);
  * the M-step scatters with NumPy fancy-index ``+=`` / ``=`` (tsc_et.py:471-475): for a repeated
    candidate only its LAST position contributes to Wp and Wq (pi and sigma see every position).

  *_loop : per-datapoint loops following the reference line by line
  *_vec  : the scores + Gram algebra the HIP kernels implement, with the last-position mask
"""
import itertools as itls

import numpy as np
from scipy.special import comb

from .bsc_oracle import Anneal  # noqa: F401

STATES = np.array([-1., 0., 1.])


def make_model(D, H, Hprime, gamma):
    """tsc_et.py:23-80: single-state matrix ((K-1)H, H), state matrix (all H'-vectors with at most gamma
    non-zeros, itertools.product order, null state first), ``no_states`` = 3**H' as upstream."""
    ss = np.concatenate([np.eye(H, dtype=np.int8) * v for v in STATES if v != 0])
    SSM = ss[np.sum(np.abs(ss), 1) == 1]
    s = np.array(list(itls.product(STATES, repeat=Hprime)), dtype=np.int8)
    state_abs = np.empty((3, 3 ** Hprime))
    for i in range(3):
        state_abs[i, :] = (s == STATES[i]).sum(axis=1)
    SM = s[np.sum(np.abs(s), axis=1) <= gamma]
    return {'D': D, 'H': H, 'Hprime': Hprime, 'gamma': gamma, 'SSM': SSM, 'SM': SM, 'no_states': s.shape[0],
            'state_abs': state_abs}


def log_prior(SM, pi):
    """tsc_et.py:327-337: sum over the H' positions of log(pi/2) for +-1 and log(1-pi) for 0."""
    pm = np.where(SM != 0, pi / 2, 1 - pi)
    return np.log(pm).sum(axis=1)


# ------------------------------------------------------------------------------------- select
def select_hprimes_loop(model, W_DH, pi, sigma, Y):
    """tsc_et.py:142-213: latent indices of the H' best one-cause states, ascending (best last)."""
    H, Hp, SSM = model['H'], model['Hprime'], model['SSM']
    W = W_DH.T
    pre1 = -1. / 2. / sigma / sigma
    pil_bar = log_prior(SSM, pi)
    Wbar = np.dot(SSM, W)
    cand = np.zeros((Y.shape[0], Hp), dtype=np.int64)
    for n in range(Y.shape[0]):
        F__ = pil_bar + pre1 * (((Wbar - Y[n]) ** 2).sum(axis=1))
        tmp = np.argsort(F__)[-Hp:]
        cand[n] = np.nonzero(SSM[tmp])[1]
    return cand


def select_scores_vec(model, W_DH, Y):
    """(N, 2H): minus the squared distance of every one-cause state up to |y|^2, -1 block then +1 block."""
    W = W_DH.T
    A = Y @ W.T
    w2 = (W * W).sum(axis=1)
    return np.concatenate([-(w2[None, :] + 2. * A), -(w2[None, :] - 2. * A)], axis=1)


def select_hprimes_vec(model, W_DH, pi, sigma, Y):
    R = select_scores_vec(model, W_DH, Y)
    idx = np.argsort(R, axis=1, kind='stable')[:, -model['Hprime']:]
    return (idx % model['H']).astype(np.int64)


# ------------------------------------------------------------------------------------- E-step
def e_step_loop(anneal, model, W_DH, pi, sigma, Y, cand):
    """tsc_et.py:277-356 -> logpj (N, S)."""
    SM = model['SM'].astype(np.float64)
    W = W_DH.T
    beta = 1. / anneal['T']
    pre1 = -1. / 2. / sigma / sigma
    pil_bar = log_prior(model['SM'], pi)
    F = np.empty((Y.shape[0], SM.shape[0]))
    for n in range(Y.shape[0]):
        Wbar = np.dot(SM, W[cand[n]])
        F[n] = pre1 * (((Wbar - Y[n]) ** 2).sum(axis=1))
    if anneal['anneal_prior']:
        F += pil_bar[None, :]
        F *= beta
    else:
        F *= beta
        F += pil_bar[None, :]
    return F


def energies_vec(model, W_DH, Y, cand):
    SM = model['SM'].astype(np.float64)
    W = W_DH.T
    A = Y @ W.T
    G = W @ W.T
    yn = (Y * Y).sum(axis=1)
    Ac = np.take_along_axis(A, cand, axis=1)
    Gc = G[cand[:, :, None], cand[:, None, :]]
    return yn[:, None] - 2. * Ac @ SM.T + np.einsum('sj,njk,sk->ns', SM, Gc, SM)


def e_step_vec(anneal, model, W_DH, pi, sigma, Y, cand):
    beta = 1. / anneal['T']
    pre1 = -1. / 2. / sigma / sigma
    pil_bar = log_prior(model['SM'], pi)
    F = pre1 * energies_vec(model, W_DH, Y, cand)
    if anneal['anneal_prior']:
        return (F + pil_bar[None, :]) * beta
    return F * beta + pil_bar[None, :]


# ------------------------------------------------------------------------------------- M-step
def pi_gamma_factors(pi, H, gamma):
    """tsc_et.py:425-432."""
    A = 0.0
    B = 0.0
    for gam1 in range(gamma + 1):
        for gam2 in range(gamma - gam1 + 1):
            cmb = comb(gam1, gam1) * comb(gam1 + gam2, gam2) * comb(H, H - gam1 - gam2)
            t = cmb * ((pi / 2) ** (gam1 + gam2)) * ((1 - pi) ** (H - gam1 - gam2))
            A += t
            B += (gam1 + gam2) * t
    return A, B, pi * H * A / B


def last_position_mask(cand):
    """(N, H') True where position j is the LAST occurrence of its latent among the datapoint's candidates."""
    N, Hp = cand.shape
    later_same = np.zeros((N, Hp), dtype=bool)
    for j in range(Hp):
        for k in range(j + 1, Hp):
            later_same[:, j] |= cand[:, j] == cand[:, k]
    return ~later_same


def m_step(anneal, model, W_DH, pi, sigma, Y, cand, logpj, to_learn=('W', 'pi', 'sigma'), vec=False):
    """tsc_et.py:359-542 -> (params, log dict with L, N_use)."""
    H, gamma = model['H'], model['gamma']
    SM = model['SM'].astype(np.float64)
    state_abs = np.abs(SM).sum(axis=1)
    W = W_DH.T
    N, D = Y.shape
    with np.errstate(divide='ignore', under='ignore'):
        all_denoms = np.exp(logpj).sum(axis=1)
    A_pg, B_pg, E_pg = pi_gamma_factors(pi, H, gamma)
    if anneal['Ncut_factor'] > 0.0:
        N_use = int(N * (1 - (1 - A_pg) * anneal['Ncut_factor']))
        cut_denom = np.sort(all_denoms, kind='mergesort')[-N_use]
        which = np.array(all_denoms >= cut_denom)
        Y, cand, logpj = Y[which], cand[which], logpj[which]
        N_use = Y.shape[0]
    else:
        N_use = N
    my_N = Y.shape[0]
    with np.errstate(divide='ignore', under='ignore'):
        Fs = np.log(np.exp(logpj).sum(axis=1)).sum()
    L = -0.5 * D * np.log(2 * np.pi * sigma ** 2) - np.log(A_pg) + Fs / N_use

    corr_all = logpj.max(axis=1)
    pjb_all = np.exp(logpj - corr_all[:, None])
    my_Wp = np.zeros_like(W)
    my_Wq = np.zeros((H, H))
    my_pi = 0.0
    my_sigma = 0.0
    if not vec:
        for n in range(my_N):
            y, c, pjb = Y[n], cand[n], pjb_all[n]
            this_Wp = np.zeros_like(my_Wp)
            this_Wq = np.zeros_like(my_Wq)
            this_Wp[c] += np.dot(np.outer(y, pjb), SM).T
            this_Wq_tmp = np.zeros_like(my_Wq[c])
            this_Wq_tmp[:, c] = np.dot(pjb * SM.T, SM)
            this_Wq[c] += this_Wq_tmp
            denom = pjb.sum()
            my_Wp += this_Wp / denom
            my_Wq += this_Wq / denom
            my_pi += np.inner(pjb, state_abs) / denom
            Wbar = np.dot(SM, W[c])
            my_sigma += (pjb * ((Wbar - y) ** 2).sum(axis=1)).sum() / denom
    else:
        q = pjb_all / pjb_all.sum(axis=1, keepdims=True)
        last = last_position_mask(cand).astype(np.float64)                      # (n, H')
        m = (q @ SM) * last                                                     # E[s] per position, masked
        expect = np.zeros((my_N, H))
        np.add.at(expect, (np.arange(my_N)[:, None], cand), m)
        my_Wp = expect.T @ Y
        B = np.einsum('ns,sj,sk->njk', q, SM, SM) * last[:, :, None] * last[:, None, :]
        np.add.at(my_Wq, (cand[:, :, None], cand[:, None, :]), B)
        my_pi = (q @ state_abs).sum()
        my_sigma = (q * energies_vec(model, W_DH, Y, cand)).sum()

    W_new = np.dot(np.linalg.pinv(my_Wq), my_Wp) if 'W' in to_learn else W
    pi_new = E_pg * my_pi / H / N_use if 'pi' in to_learn else pi
    sigma_new = np.sqrt(my_sigma / D / N_use) if 'sigma' in to_learn else sigma
    params = {'W': W_new.transpose(), 'pi': pi_new, 'sigma': sigma_new, 'Q': 0.}
    return params, {'N_use': N_use, 'L': L, 'stats': {'Wp': my_Wp, 'Wq': my_Wq, 'pi': my_pi, 'sigma': my_sigma}}


def em_step(anneal, model, params, Y, vec=True):
    sel = select_hprimes_vec if vec else select_hprimes_loop
    est = e_step_vec if vec else e_step_loop
    cand = sel(model, params['W'], params['pi'], params['sigma'], Y)
    logpj = est(anneal, model, params['W'], params['pi'], params['sigma'], Y, cand)
    new, log = m_step(anneal, model, params['W'], params['pi'], params['sigma'], Y, cand, logpj, vec=vec)
    log['candidates'], log['logpj'] = cand, logpj
    return new, log
