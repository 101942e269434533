"""ORACLE -- test infrastructure, not product code.

CPU restatement (NumPy, float64) of the reference's Gaussian (spike-and-slab) Sparse Coding
truncated-EM hot path, prosper/em/camodels/gsc_et.py (class ``GSC``, reference v0.1.0), for the three
observation-noise types: scalar (BASELINE config 4), diagonal (D,) and full (D,D) ``sigma_sq``.  Imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

The reference buckets datapoints by candidate set and returns its statistics in bucket order
(gsc_et.py:721-749, 572-573); everything the M-step consumes is a sum over datapoints, so this
restatement keeps the original order.  It uses the scores + Gram algebra the kernels implement
(GSC delta):  a = W^T y,  G = W^T W,
    r = y - W_a mu_a,  |r|^2 = |y|^2 - 2 mu_a.a_a + mu_a^T G_aa mu_a,  b = W_a^T r = a_a - G_aa mu_a
    Lambda = G_aa / s2 + Psi_a^-1
    lp = -(logdet Psi_a + logdet Lambda) - |r|^2 / s2 + b^T Lambda^-1 b / s2^2 + sum_a logit(pi)
    kappa = Lambda^-1 b / s2 + mu_a,   E[z z^T] = kappa kappa^T + Lambda^-1
Weights are the reference's UN-stabilised exp(beta * lp), NaN / underflow clamped to ``tiny``
(gsc_et.py:354-356, 518-522); the null state's weight exp(-beta |y|^2 / s2) is not clamped (:476-478).

For a diagonal / full noise covariance Sigma every inner product above becomes Sigma^-1-weighted:
a = W^T Sigma^-1 y, G = W^T Sigma^-1 W, |y|^2 -> y^T Sigma^-1 y, and s2 = 1 (gsc_et.py:321-346, 414-425).

Pinned against outputs of the reference itself (tests/golden/gsc_step_*.npz).
"""
import numpy as np

from .bsc_oracle import generate_state_matrix, Anneal  # noqa: F401

TINY = np.finfo(np.float64).tiny
LOG_TINY = np.finfo(np.float64).min


def make_model(D, H, Hprime, gamma):
    """GSC.__init__ (gsc_et.py:32-55): out-of-range gamma / Hprime are silently reset."""
    if gamma <= 0 or gamma > H:
        gamma = H
    if Hprime <= 0 or Hprime > H:
        Hprime = H
    elif Hprime < gamma:
        gamma = Hprime
    sl, S, SM, state_abs = generate_state_matrix(Hprime, gamma)
    return {'D': D, 'H': H, 'Hprime': Hprime, 'gamma': gamma, 'SM': SM, 'state_abs': state_abs, 'S': S}


def noise_terms(params, Y=None):
    """(Ws, s2, yn): Ws = Sigma^-1 W for a diagonal / full sigma_sq (W itself for a scalar one, with its
    1/s2 kept explicit as upstream), s2 the remaining scalar, yn = y^T Sigma^-1 y (|y|^2 for scalar)."""
    W, sig = params['W'], np.asarray(params['sigma_sq'], dtype=np.float64)
    if sig.ndim == 0:
        return W, float(sig), None if Y is None else (Y * Y).sum(axis=1)
    if sig.ndim == 1:
        sinv = 1. / sig
        return W * sinv[:, None], 1.0, None if Y is None else (Y * Y * sinv[None, :]).sum(axis=1)
    Sinv = np.linalg.inv(sig)
    return Sinv @ W, 1.0, None if Y is None else ((Y @ Sinv) * Y).sum(axis=1)


def _singleton_terms(params):
    """Per-latent constants of the one-cause states (gsc_et.py:481-511, 777-799)."""
    W, mu = params['W'], params['mu']
    Ws, s2, _ = noise_terms(params)
    psi = np.diag(params['psi_sq'])
    Gd = (Ws * W).sum(axis=0)
    lam = Gd / s2 + 1. / psi
    norm_const = -(np.log(psi) + np.log(lam))
    return Gd, lam, norm_const


def component_scores(params, Y):
    """gsc_et.py:752-809: singleton log-posterior without the prior, NaN/-inf -> float min, inf -> 0."""
    W, mu = params['W'], params['mu']
    Ws, s2, yn = noise_terms(params, Y)
    Gd, lam, norm_const = _singleton_terms(params)
    A = Y @ Ws
    r2 = yn[:, None] - 2. * mu[None, :] * A + (mu * mu * Gd)[None, :]
    b = A - (Gd * mu)[None, :]
    post = norm_const[None, :] - r2 / s2 + b * b / lam[None, :] / s2 ** 2
    post[np.isnan(post)] = LOG_TINY
    post[post < LOG_TINY] = LOG_TINY
    post[np.isinf(post)] = 0
    return post


def select_hprimes(params, Y, Hprime):
    """gsc_et.py:721-728: the Hprime best-scoring latents, sorted by index."""
    cand = np.argsort(component_scores(params, Y), axis=1)[:, -Hprime:]
    return np.sort(cand, axis=1).astype(np.int64)


def _state_quantities(params, Y, cand, act):
    """For the multi-cause state using candidate positions ``act`` (tuple): lp without prior
    (N,), kappa (N,g), Lambda^-1 (N,g,g), and the latent indices (N,g)."""
    W, mu, psi = params['W'], params['mu'], params['psi_sq']
    Ws, s2, yn = noise_terms(params, Y)
    idx = cand[:, list(act)]                                        # (N, g)
    N, g = idx.shape
    Wa = W[:, idx].transpose(1, 0, 2)                               # (N, D, g)
    Wsa = Ws[:, idx].transpose(1, 0, 2)
    mua = mu[idx]                                                   # (N, g)
    Psia = psi[idx[:, :, None], idx[:, None, :]]                    # (N, g, g)
    Gaa = np.einsum('ndi,ndj->nij', Wsa, Wa)
    Gaa = 0.5 * (Gaa + Gaa.transpose(0, 2, 1)) if Ws is not W else Gaa
    aa = np.einsum('ndi,nd->ni', Wsa, Y)
    Lam = Gaa / s2 + np.linalg.inv(Psia)
    Lam_inv = np.linalg.inv(Lam)
    r2 = yn - 2. * (mua * aa).sum(axis=1) + np.einsum('ni,nij,nj->n', mua, Gaa, mua)
    b = aa - np.einsum('nij,nj->ni', Gaa, mua)
    quad = np.einsum('ni,nij,nj->n', b, Lam_inv, b)
    C_det = np.linalg.slogdet(Psia)[1] + np.linalg.slogdet(Lam)[1]
    lp = -C_det - r2 / s2 + quad / s2 ** 2
    kappa = np.einsum('nij,nj->ni', Lam_inv, b) / s2 + mua
    return lp, kappa, Lam_inv, idx


def compute_lpj(model, params, Y, cand):
    """gsc_et.py:811-944 -> logpj (N, 1+H+S) (no beta, prior included)."""
    H, SM = model['H'], model['SM']
    Ws, s2, yn = noise_terms(params, Y)
    lpi = np.log(params['pi']) - np.log(1 - np.array(params['pi']))
    N = Y.shape[0]
    out = np.zeros((N, 1 + H + SM.shape[0]))
    out[:, 0] = -yn / s2
    W, mu = params['W'], params['mu']
    Gd, lam, norm_const = _singleton_terms(params)
    A = Y @ Ws
    r2 = yn[:, None] - 2. * mu[None, :] * A + (mu * mu * Gd)[None, :]
    b = A - (Gd * mu)[None, :]
    out[:, 1:H + 1] = norm_const[None, :] - r2 / s2 + b * b / lam[None, :] / s2 ** 2 + lpi[None, :]
    for s in range(SM.shape[0]):
        act = tuple(np.nonzero(SM[s])[0])
        lp, _, _, idx = _state_quantities(params, Y, cand, act)
        out[:, 1 + H + s] = lp + lpi[idx].sum(axis=1)
    return out


def e_step(anneal, model, params, Y, cand):
    """gsc_et.py:401-580 (original datapoint order) -> xpt_s (N,H), xpt_ss (N,H,H), xpt_sz (N,H),
    xpt_szsz (N,H,H)."""
    H, SM = model['H'], model['SM']
    beta = 1. / anneal['T']
    Ws, s2, yn = noise_terms(params, Y)
    N = Y.shape[0]
    lpi = np.log(params['pi']) - np.log(1 - np.array(params['pi']))
    W, mu = params['W'], params['mu']

    def weight(lp_plus_prior):
        p = np.exp(lp_plus_prior * beta)
        p[np.isnan(p)] = TINY
        p[p < TINY] = TINY
        return p

    pstr_s = np.zeros((N, H))
    pstr_ss = np.zeros((N, H, H))
    pstr_sz = np.zeros((N, H))
    pstr_szsz = np.zeros((N, H, H))
    nfac = np.exp(-yn / s2 * beta)                                   # null state, not clamped

    Gd, lam, norm_const = _singleton_terms(params)
    A = Y @ Ws
    r2 = yn[:, None] - 2. * mu[None, :] * A + (mu * mu * Gd)[None, :]
    b = A - (Gd * mu)[None, :]
    lp1 = norm_const[None, :] - r2 / s2 + b * b / lam[None, :] / s2 ** 2 + lpi[None, :]
    p1 = weight(lp1)
    kap1 = b / lam[None, :] / s2 + mu[None, :]
    nfac += p1.sum(axis=1)
    hh = np.arange(H)
    pstr_s += p1
    pstr_ss[:, hh, hh] += p1
    pstr_sz += p1 * kap1
    pstr_szsz[:, hh, hh] += p1 * (kap1 ** 2 + 1. / lam[None, :])

    rows = np.arange(N)
    for s in range(SM.shape[0]):
        act = tuple(np.nonzero(SM[s])[0])
        lp, kappa, Lam_inv, idx = _state_quantities(params, Y, cand, act)
        p = weight(lp + lpi[idx].sum(axis=1))
        nfac += p
        np.add.at(pstr_s, (rows[:, None], idx), p[:, None])
        np.add.at(pstr_sz, (rows[:, None], idx), p[:, None] * kappa)
        ii, jj = idx[:, :, None], idx[:, None, :]
        np.add.at(pstr_ss, (rows[:, None, None], ii, jj), p[:, None, None] * np.ones_like(Lam_inv))
        np.add.at(pstr_szsz, (rows[:, None, None], ii, jj),
                  p[:, None, None] * (kappa[:, :, None] * kappa[:, None, :] + Lam_inv))

    nf = 1.0 / (nfac + TINY)
    return {'xpt_s': pstr_s * nf[:, None], 'xpt_ss': pstr_ss * nf[:, None, None],
            'xpt_sz': pstr_sz * nf[:, None], 'xpt_szsz': pstr_szsz * nf[:, None, None]}


def m_step(model, params, suff, Y, to_learn=('W', 'pi', 'mu', 'sigma_sq', 'psi_sq'), N_total=None):
    """gsc_et.py:584-718.  Returns the updated parameter dict."""
    H, D = model['H'], model['D']
    eps = 1e-5
    xs, xsz, xss, xszsz = suff['xpt_s'], suff['xpt_sz'], suff['xpt_ss'], suff['xpt_szsz']
    N = Y.shape[0] if N_total is None else N_total
    sum_s, sum_sz = xs.sum(axis=0), xsz.sum(axis=0)
    sum_szsz, sum_ss = xszsz.sum(axis=0), xss.sum(axis=0)
    Wp = Y.T @ xsz
    W_n = np.dot(Wp, np.linalg.inv(sum_szsz))
    new = dict(params)
    if 'pi' in to_learn:
        pi_eps = 5e-5
        pi_new = sum_s / N
        pi_new[pi_new <= pi_eps] = pi_eps
        pi_new[pi_new >= (1 - pi_eps)] = 1 - pi_eps
        new['pi'] = pi_new
    if 'W' in to_learn:
        new['W'] = W_n
    if 'mu' in to_learn:
        new['mu'] = sum_sz * 1. / (sum_s + np.finfo(np.float64).eps)
    if 'psi_sq' in to_learn:
        mu = new['mu']
        psi = np.outer(mu, mu) * sum_ss + sum_szsz - 2 * (mu[:, None] * (xs.T @ xsz))
        new['psi_sq'] = (psi * np.linalg.inv(sum_ss + eps * np.eye(H))) + (eps * np.eye(H))
    if 'sigma_sq' in to_learn:
        U = xsz.T @ xsz                                               # sum_n xpt_sz xpt_sz^T
        kind = np.ndim(params['sigma_sq'])
        if kind == 2:                                                 # full (gsc_et.py:677-688)
            new['sigma_sq'] = (Y.T @ Y - W_n @ U @ W_n.T) / N + eps * np.eye(D)
        elif kind == 1:                                               # diagonal (gsc_et.py:690-701)
            new['sigma_sq'] = ((Y * Y).sum(axis=0) - ((W_n @ U) * W_n).sum(axis=1)) / N + eps
        else:                                                         # scalar (gsc_et.py:703-713)
            WT_outer = np.dot(W_n.T, W_n)
            my = (Y * Y).sum() - np.trace(U @ WT_outer)
            new['sigma_sq'] = my / N / D + eps
    return new


def em_step(anneal, model, params, Y):
    cand = select_hprimes(params, Y, model['Hprime'])
    suff = e_step(anneal, model, params, Y, cand)
    new = m_step(model, params, suff, Y)
    return new, {'candidates': cand, 'suff': suff}


def generate_gsc_data(params, N, rng):
    """Spike-and-slab data with scalar noise: s ~ Bernoulli(pi), z_a ~ N(mu_a, Psi_aa),
    y = W_a z_a + N(0, sigma_sq) (gsc_et.py:186-257; the test generator draws in vectorised order)."""
    D, H = params['W'].shape
    s = rng.random_sample((N, H)) <= params['pi']
    z = np.zeros((N, H))
    L = np.linalg.cholesky(params['psi_sq'])
    z_all = params['mu'][None, :] + rng.normal(size=(N, H)) @ L.T
    z[s] = z_all[s]
    y = z @ params['W'].T + np.sqrt(params['sigma_sq']) * rng.normal(size=(N, D))
    return y, s, z
