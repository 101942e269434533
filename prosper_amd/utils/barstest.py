"""Bars-test ground truth (config 1 of BASELINE.json).

``generate_bars_dict`` follows prosper/utils/barstest.py:8-32: H basis vectors holding
the R horizontal and R vertical bars of an R x R grid (R = H // 2), returned (D, H).
"""
import numpy as np


def generate_bars_dict(H, neg_bars=False):
    R = H // 2
    D = R ** 2
    W_gt = np.zeros((R, R, H))
    for i in range(R):
        W_gt[i, :, i] = 1.
        W_gt[:, i, R + i] = 1.
    if neg_bars:
        sign = 1 - 2 * np.random.randint(2, size=(H, ))
        W_gt = sign[None, None, :] * W_gt
    return W_gt.reshape((D, H))


def find_permutation(W, Wgt):
    """Greedy column matching of W against Wgt by mean absolute error; returns
    (permutation, mean abs error) -- the check the bars test eyeballs upstream
    (barstest.py:56-98)."""
    D, H = Wgt.shape
    err = np.abs(W[:, :, None] - Wgt[:, None, :]).mean(axis=0)   # (H_learned, H_gt)
    perm = -np.ones(H, dtype=int)
    used = np.zeros(W.shape[1], dtype=bool)
    total = 0.0
    for g in np.argsort(err.min(axis=0)):
        cand = np.where(~used)[0]
        best = cand[np.argmin(err[cand, g])]
        perm[g] = best
        used[best] = True
        total += err[best, g]
    return perm, total / H
