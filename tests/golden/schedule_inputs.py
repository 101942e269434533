"""Seeded inputs of the 50-step schedule trajectories (tests/golden/make_golden.py mints the reference's run on them, the GPU
tests re-create them from the same seeds: legacy ``RandomState`` streams do not change between NumPy versions).  Data only --
nothing of the reference is imported here."""
import numpy as np

DSC_STATES = np.array([-1.0, 0.0, 1.0, 2.0])
DSC_PI = np.array([0.03, 0.91, 0.04, 0.02])


def schedule_inputs(model, D, H, N, seed):
    """(y (N, D), initial parameters) for ``model`` in {"bsc", "mca", "gsc", "mmca", "dsc", "tsc"}."""
    rng = np.random.RandomState(seed)
    if model == "bsc":
        # (sigma_gt = 2: the reference's un-stabilised exp(logpj) sums stay above the underflow threshold at D = 1024)
        W_gt = rng.normal(size=(D, H))
        s = rng.random_sample((N, H)) < 2.0 / H
        y = s.astype(np.float64) @ W_gt.T + 2.0 * rng.normal(size=(N, D))
        p0 = {"W": W_gt + 0.3 * rng.normal(size=(D, H)), "pi": 2.5 / H, "sigma": 2.3}
    elif model == "mca":
        W_gt = np.abs(rng.normal(size=(D, H))) * 3.0 + 0.1
        s = rng.random_sample((N, H)) < 2.0 / H
        y = np.where(s[:, None, :], W_gt[None, :, :], 0.0).max(axis=2) + rng.normal(size=(N, D))
        p0 = {"W": W_gt * rng.uniform(0.85, 1.15, size=(D, H)), "pi": 2.5 / H, "sigma": 1.2}
    elif model == "gsc":
        W_gt = rng.normal(size=(D, H))
        s = rng.random_sample((N, H)) < 2.0 / H
        z = s * (1.5 + rng.normal(size=(N, H)))
        y = z @ W_gt.T + rng.normal(size=(N, D))
        p0 = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": np.full(H, 2.5 / H), "mu": 1.4 + 0.1 * rng.normal(size=H),
              "psi_sq": np.diag(rng.uniform(0.8, 1.3, size=H)), "sigma_sq": 1.3}
    elif model == "mmca":
        W_gt = rng.normal(size=(D, H)) * 3.0
        W_gt = np.where(np.abs(W_gt) < 0.05, 0.05, W_gt)
        s = rng.random_sample((N, H)) < 2.0 / H
        cand = np.where(s[:, None, :], W_gt[None, :, :], 0.0)
        pick = np.abs(cand).argmax(axis=2)
        y = np.take_along_axis(cand, pick[:, :, None], axis=2)[:, :, 0] + rng.normal(size=(N, D))
        p0 = {"W": W_gt * rng.uniform(0.85, 1.15, size=(D, H)), "pi": 2.5 / H, "sigma": 1.2}
    elif model in ("dsc", "tsc"):
        states = DSC_STATES if model == "dsc" else np.array([-1.0, 0.0, 1.0])
        pi_gt = DSC_PI if model == "dsc" else np.array([1.25 / H, 1.0 - 2.5 / H, 1.25 / H])
        W_gt = rng.normal(size=(D, H)) * 2.0
        s = rng.choice(states, size=(N, H), replace=True, p=pi_gt)
        y = s @ W_gt.T + rng.normal(size=(N, D))
        if model == "dsc":
            pi0 = pi_gt * rng.uniform(0.8, 1.25, size=pi_gt.shape)
            p0 = {"W": W_gt + 0.3 * rng.normal(size=(D, H)), "pi": pi0 / pi0.sum(), "sigma": 1.2}
        else:
            p0 = {"W": W_gt + 0.3 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.15}
    else:
        raise ValueError(model)
    return np.ascontiguousarray(y), p0
