"""ORACLE -- test infrastructure, not product code.

CPU restatement (NumPy, float64) of the reference's Discrete Sparse Coding truncated-EM hot path,
prosper/em/camodels/dsc_et.py (reference v0.1.0): K-ary latents with values ``states`` (one of them 0),
linear superposition, Gaussian noise.  Imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.

  *_loop : per-datapoint loops following the reference line by line
  *_vec  : vectorised algebra the HIP kernels implement (scores GEMM + Gram identities)

Pinned against outputs of the reference itself (tests/golden/dsc_step_*.npz, minted by
tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
import itertools as itls

import numpy as np
from scipy.special import gammaln, logsumexp

from .bsc_oracle import Anneal  # noqa: F401


def make_model(D, H, Hprime, gamma, states):
    """dsc_et.py:135-191: state tables.  Multi-cause rows in itertools.product order (NOT by size)."""
    states = np.asarray(states, dtype=np.float64)
    assert Hprime <= H and gamma <= Hprime
    K = states.shape[0]
    K_0 = int(np.argwhere(states == 0.)[0, 0])
    ss = np.empty((0, H), dtype=np.int8)
    for i in range(K):
        if i == K_0:
            continue
        ss = np.concatenate((ss, np.eye(H, dtype=np.int8) * states[i]))
    SSM = ss[np.sum(np.abs(np.sign(ss)), 1) == 1]
    sl = [np.array(c) for c in itls.product(states, repeat=Hprime)
          if (np.sum(np.array(c) != 0) <= gamma and np.sum(np.array(c) != 0) > 1)]
    SM = np.array(sl) if sl else np.zeros((0, Hprime))
    S = SM.shape[0]
    state_abs = np.empty((K, S))
    for i in range(K):
        state_abs[i, :] = (SM == states[i]).sum(axis=1)
    state_abs[K_0, :] = H - state_abs.sum(0) + state_abs[K_0, :]
    return {'D': D, 'H': H, 'Hprime': Hprime, 'gamma': gamma, 'states': states, 'K': K, 'K_0': K_0,
            'SSM': SSM, 'SM': SM, 'state_abs': state_abs, 'no_states': S}


def prior_terms(model, pi):
    """dsc_et.py:539-558: pre_F (1 + (K-1)H + S,)."""
    H, K, K_0, S = model['H'], model['K'], model['K_0'], model['no_states']
    l_pis = np.zeros(S)
    for i in range(K):
        l_pis += model['state_abs'][i] * np.log(pi[i])
    pre_F = np.empty(1 + (K - 1) * H + S)
    pre_F[0] = H * np.log(pi[K_0])
    c = 0
    for state in range(K):
        if state == K_0:
            continue
        pre_F[c * H + 1:(c + 1) * H + 1] = np.log(pi[state]) + ((H - 1) * np.log(pi[K_0]))
        c += 1
    pre_F[(K - 1) * H + 1:] = l_pis
    return pre_F


# ------------------------------------------------------------------------------------- select
def select_hprimes_loop(model, W_DH, pi, sigma, Y):
    """dsc_et.py:347-410: latents ranked by their best singleton log-joint, best first."""
    H, K, K_0, Hp, SSM = model['H'], model['K'], model['K_0'], model['Hprime'], model['SSM']
    W = W_DH.T
    N = Y.shape[0]
    pre1 = -1. / 2. / sigma / sigma
    l_pis = np.zeros(H * (K - 1))
    c = 0
    for i in range(K):
        if i == K_0:
            continue
        l_pis[c * H:(c + 1) * H] += np.log(pi[i]) + (H - 1) * np.log(pi[K_0])
        c += 1
    cand = np.zeros((N, Hp), dtype=np.int64)
    Wbar = np.dot(SSM, W)
    for n in range(N):
        F__ = pre1 * (((Wbar - Y[n]) ** 2).sum(axis=1)) + l_pis
        sort_prob_ind = np.mod(np.argsort(F__), H)[::-1]
        Fu, Si = np.unique(sort_prob_ind, return_index=True)
        cand[n] = Fu[np.argsort(Si)][:Hp]
    return cand


def select_scores_vec(model, W_DH, pi, sigma, Y):
    """Per latent the best singleton log-joint up to datapoint constants:
    max_k pre1 (v_k^2 |W_h|^2 - 2 v_k <W_h,y>) + log pi_k."""
    K, K_0, states = model['K'], model['K_0'], model['states']
    W = W_DH.T
    pre1 = -1. / 2. / sigma / sigma
    A = Y @ W.T
    w2 = (W * W).sum(axis=1)
    best = np.full(A.shape, -np.inf)
    for k in range(K):
        if k == K_0:
            continue
        v = states[k]
        best = np.maximum(best, pre1 * (v * v * w2[None, :] - 2. * v * A) + np.log(pi[k]))
    return best


def select_hprimes_vec(model, W_DH, pi, sigma, Y):
    best = select_scores_vec(model, W_DH, pi, sigma, Y)
    return np.argsort(-best, axis=1, kind='stable')[:, :model['Hprime']].astype(np.int64)


# ------------------------------------------------------------------------------------- E-step
def e_step_loop(anneal, model, W_DH, pi, sigma, Y, cand):
    """dsc_et.py:492-585 -> logpj (N, 1 + (K-1)H + S)."""
    H, K, SSM, SM, S = model['H'], model['K'], model['SSM'], model['SM'], model['no_states']
    W = W_DH.T
    N = Y.shape[0]
    beta = 1. / anneal['T']
    pre1 = -1. / 2. / sigma / sigma
    pre_F = prior_terms(model, pi)
    F = np.empty([N, 1 + (K - 1) * H + S])
    SSMW = np.dot(SSM, W)
    for n in range(N):
        y = Y[n, :]
        F[n, 0] = pre1 * (y ** 2).sum()
        F[n, 1:(K - 1) * H + 1] = pre1 * ((SSMW - y) ** 2).sum(axis=1)
        if model['gamma'] > 1:
            Wbar = np.dot(SM, W[cand[n]])
            F[n, (K - 1) * H + 1:] = pre1 * (((Wbar - y) ** 2).sum(axis=1))
    if anneal['anneal_prior']:
        F[:, :] += pre_F[None, :]
        F[:, :] *= beta
    else:
        F[:, :] *= beta
        F[:, :] += pre_F[None, :]
    return F


def energies_vec(model, W_DH, Y, cand):
    """All squared reconstruction errors through a = Y W^T and G = W W^T."""
    H, K, K_0, states, SM = model['H'], model['K'], model['K_0'], model['states'], model['SM']
    W = W_DH.T
    A = Y @ W.T
    G = W @ W.T
    yn = (Y * Y).sum(axis=1)
    cols = [yn[:, None]]
    for k in range(K):
        if k == K_0:
            continue
        v = states[k]
        cols.append(v * v * np.diag(G)[None, :] - 2. * v * A + yn[:, None])
    if SM.shape[0]:
        Ac = np.take_along_axis(A, cand, axis=1)                                 # (N, H')
        Gc = G[cand[:, :, None], cand[:, None, :]]                               # (N, H', H')
        cols.append(yn[:, None] - 2. * Ac @ SM.T + np.einsum('sj,njk,sk->ns', SM, Gc, SM))
    return np.concatenate(cols, axis=1)


def e_step_vec(anneal, model, W_DH, pi, sigma, Y, cand):
    beta = 1. / anneal['T']
    pre1 = -1. / 2. / sigma / sigma
    pre_F = prior_terms(model, pi)
    F = pre1 * energies_vec(model, W_DH, Y, cand)
    if anneal['anneal_prior']:
        return (F + pre_F[None, :]) * beta
    return F * beta + pre_F[None, :]


# ------------------------------------------------------------------------------------- M-step
def multinom2(n, k):
    return np.exp(gammaln(n + 1) - gammaln(k + 1).sum())


def scaling_factor(model, pi):
    """dsc_et.py:798-823: prior mass of the states with at most gamma non-zeros."""
    H, gamma, K_0 = model['H'], model['gamma'], model['K_0']
    A = 0.0
    for gp in itls.product(np.arange(gamma + 1), repeat=model['K'] - 1):
        ngp = np.array(gp)
        if ngp.sum() > gamma:
            continue
        abs_array = np.insert(ngp, K_0, H - ngp.sum())
        A += multinom2(abs_array.sum(), abs_array) * np.prod(pi ** abs_array)
    return A


def m_step(anneal, model, W_DH, pi, sigma, Y, cand, logpj, to_learn=('W', 'pi', 'sigma'), vec=False):
    """dsc_et.py:587-774 -> (params, log dict with L, N_use, prior_mass)."""
    H, K, K_0, states = model['H'], model['K'], model['K_0'], model['states']
    SSM, SM, state_abs, gamma = model['SSM'], model['SM'], model['state_abs'], model['gamma']
    W = W_DH.T
    N, D = Y.shape
    nss = (K - 1) * H
    all_denoms = np.exp(logpj).sum(axis=1)
    A_pi_gamma = scaling_factor(model, pi)

    if anneal['Ncut_factor'] > 0.0:                                             # dsc_et.py:825-843 (strict >)
        N_use = int(N * (1 - (1 - A_pi_gamma) * anneal['Ncut_factor']))
        cut_denom = np.sort(all_denoms, kind='mergesort')[-N_use]
        which = np.array(all_denoms > cut_denom)
        Y, cand, logpj = Y[which], cand[which], logpj[which]
        N_use = Y.shape[0]
    else:
        N_use = N
    my_N = Y.shape[0]

    corr_all = logpj.max(axis=1)
    pjb_all = np.exp(logpj - corr_all[:, None])
    L = -0.5 * D * np.log(2 * np.pi * sigma ** 2) + logsumexp(logpj, 1).sum() / N_use

    my_Wp = np.zeros_like(W)
    my_Wq = np.zeros((H, H))
    my_pi = np.zeros_like(pi)
    my_sigma = 0.0
    if not vec:
        for n in range(my_N):
            y, c, pjb = Y[n], cand[n], pjb_all[n]
            this_Wp = np.zeros_like(my_Wp)
            this_Wq = np.zeros_like(my_Wq)
            this_pi = np.zeros_like(pi)
            this_pi[K_0] = H * pjb[0]
            this_sigma = pjb[0] * (y ** 2).sum()
            cc = 0
            for state in range(K):
                if state == K_0:
                    continue
                sspjb = pjb[cc * H + 1:(cc + 1) * H + 1]
                this_pi[state] += sspjb.sum()
                sqe = ((states[state] * W - y) ** 2).sum(1)
                this_sigma += (sspjb * sqe).sum()
                cc += 1
            this_pi[K_0] += ((H - 1) * pjb[1:nss + 1]).sum()
            this_Wp += np.dot(np.outer(y, pjb[1:nss + 1]), SSM).T
            this_Wq += np.dot(pjb[1:nss + 1] * SSM.T, SSM)
            if gamma > 1:
                this_Wp[c] += np.dot(np.outer(y, pjb[nss + 1:]), SM).T
                this_Wq_tmp = np.zeros_like(my_Wq[c])
                this_Wq_tmp[:, c] = np.dot(pjb[nss + 1:] * SM.T, SM)
                this_Wq[c] += this_Wq_tmp
                this_pi += np.inner(pjb[nss + 1:], state_abs)
                Wbar = np.dot(SM, W[c])
                this_sigma += (pjb[nss + 1:] * ((Wbar - y) ** 2).sum(axis=1)).sum()
            denom = pjb.sum()
            my_Wp += this_Wp / denom
            my_Wq += this_Wq / denom
            my_pi += this_pi / denom
            my_sigma += this_sigma / denom / D
    else:
        q = pjb_all / pjb_all.sum(axis=1, keepdims=True)
        q1 = q[:, 1:nss + 1].reshape(my_N, K - 1, H)
        qs = q[:, nss + 1:]
        vals = np.array([states[k] for k in range(K) if k != K_0])
        expect = np.einsum('nkh,k->nh', q1, vals)                                # E[s]
        diag = np.einsum('nkh,k->h', q1, vals * vals)
        my_Wq = np.diag(diag)
        if SM.shape[0]:
            m = qs @ SM                                                          # (n, H')
            np.add.at(expect, (np.arange(my_N)[:, None], cand), m)
            B = np.einsum('ns,sj,sk->njk', qs, SM, SM)
            np.add.at(my_Wq, (cand[:, :, None], cand[:, None, :]), B)
        my_Wp = expect.T @ Y
        cnt = np.zeros(K)
        c = 0
        for k in range(K):
            if k == K_0:
                continue
            cnt[k] = q1[:, c, :].sum() + ((qs @ (SM == states[k]).sum(axis=1)) .sum() if SM.shape[0] else 0.0)
            c += 1
        cnt[K_0] = H * my_N - cnt.sum()
        my_pi = cnt
        beta = 1. / anneal['T']
        pre1 = -1. / 2. / sigma / sigma
        pre_F = prior_terms(model, pi)
        e = ((logpj / beta - pre_F[None, :]) if anneal['anneal_prior'] else (logpj - pre_F[None, :]) / beta) / pre1
        my_sigma = (q * e).sum() / D

    W_new = np.linalg.lstsq(my_Wq, my_Wp, rcond=None)[0]
    pi_new = my_pi / my_pi.sum()
    eps = 1e-6
    if np.any(pi_new < eps):
        which_lo = pi_new < eps
        which_hi = pi_new >= eps
        pi_new[which_lo] += eps - pi_new[which_lo]
        pi_new[which_hi] -= (eps * np.sum(which_lo)) / np.sum(which_hi)
    sigma_new = np.sqrt(my_sigma / N_use)
    if 'W' not in to_learn:
        W_new = W
    if 'pi' not in to_learn:
        pi_new = pi
    if 'sigma' not in to_learn:
        sigma_new = sigma
    params = {'W': W_new.transpose(), 'pi': pi_new, 'sigma': sigma_new, 'Q': 0.}
    return params, {'N_use': N_use, 'L': L, 'prior_mass': A_pi_gamma,
                    'stats': {'Wp': my_Wp, 'Wq': my_Wq, 'pi': my_pi, 'sigma': my_sigma}}


def em_step(anneal, model, params, Y, vec=True):
    """select -> E -> M (camodels/__init__.py:163-193 for DSC_ET; check_params only asserts)."""
    sel = select_hprimes_vec if vec else select_hprimes_loop
    est = e_step_vec if vec else e_step_loop
    cand = sel(model, params['W'], params['pi'], params['sigma'], Y)
    logpj = est(anneal, model, params['W'], params['pi'], params['sigma'], Y, cand)
    new, log = m_step(anneal, model, params['W'], params['pi'], params['sigma'], Y, cand, logpj, vec=vec)
    log['candidates'], log['logpj'] = cand, logpj
    return new, log
