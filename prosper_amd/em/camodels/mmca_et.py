"""Maximum-Magnitude Causes Analysis (signed MCA) on the MI355X: drop-in for
prosper/em/camodels/mmca_et.py.

Same constructor, ``check_params`` (|W| >= 1e-4), ``select_Hprimes / E_step / M_step`` signatures,
return keys (``W, pi, sigma, Q``) and ``dlog`` side effect (``N_use``) as the reference's ``MMCA_ET``
(mmca_et.py:25-427).  The kernels are MCA's (prosper_amd/csrc/mca_kernels.hip) with
``pm_mca_params.signed_w = 1``:

  select_Hprimes  the H' smallest |W_h - y|^2 = |W_h|^2 - 2 <W_h,y> + |y|^2: f64 MFMA scores GEMM +
                  the 16-lane selection kernel in distance mode
  E_step          Wbar_sd = sign(t) |t|^(1/rho), t = sum_{j in s} sign(W)|W|^rho [c_j, d]
  M_step          Aid[j,d] = sum_s q_s min(1, (|W_jd| / |Wbar_sd|)^(rho-1)); W update with inertia
"""
import numpy as np

from ._device import DeviceArray, _ptr
from .mca_et import MCA_ET
from ...utils import parallel
from ...utils import tracing
from ...utils.datalog import dlog

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

from math import pi as _PI


class MMCA_ET(MCA_ET):
    """Signed max-magnitude superposition with Expectation Truncation."""

    def __init__(self, D, H, Hprime, gamma, to_learn=['W', 'pi', 'sigma'], comm=parallel.COMM_WORLD,
                 device=None):
        MCA_ET.__init__(self, D, H, Hprime, gamma, to_learn, comm, device)
        self.rho_T_bound = 1.20       # for rho: never use a T smaller than this   (mmca_et.py:37)
        self.rho_lbound = 1           # for rho: never use a rho smaller than this (mmca_et.py:38)
        self.rho_ubound = 35          # for rho: never use a rho larger than this  (mmca_et.py:39)
        self.tol = 1e-4               # for W: ensure |W| >= tol                   (mmca_et.py:40)
        tol = self.tol
        self.noise_policy = {
            'W':     (-np.inf, +np.inf, False),
            'pi':    (tol, 1 - tol, False),
            'sigma': (tol, +np.inf, False),
        }
        self.signed_w = 1.0

    @tracing.traced
    def check_params(self, model_params):
        """|W| >= tol, in place like upstream (mmca_et.py:50-63)."""
        tol = self.tol
        W = model_params['W'].T
        W[np.logical_and(W >= 0., W < +tol)] = +tol
        W[np.logical_and(W <= 0., W > -tol)] = -tol
        return model_params

    def _superpose(self, model_params, s, g):
        """Per dimension the active cause of largest magnitude, first one on ties (mmca_et.py:66-93), in chunks."""
        Wt = torch.from_numpy(np.ascontiguousarray(np.asarray(model_params['W'], dtype=np.float64).T)).to(self.device)  # (H,D)
        y = torch.zeros((s.shape[0], self.D), dtype=torch.float64, device=self.device)
        step = max(1, (1 << 25) // (self.D * self.H))
        for lo in range(0, s.shape[0], step):
            t0 = s[lo:lo + step].to(torch.float64)[:, :, None] * Wt[None, :, :]          # (n,H,D)
            idx = t0.abs().argmax(dim=1, keepdim=True)
            y[lo:lo + step] = torch.gather(t0, 1, idx)[:, 0, :]
        return y

    def _generate_data_host(self, model_params, my_N):
        """CAModel.generate_data (camodels/__init__.py:104-122): one ``random((my_N, H))`` draw for the
        latents, then ``generate_from_hidden``."""
        p = np.random.random(size=(my_N, self.H))
        return self.generate_from_hidden(model_params, {'s': p < model_params['pi']})

    @tracing.traced
    def generate_from_hidden(self, model_params, my_hdata):
        """Per dimension the active cause of largest magnitude, plus one ``normal((my_N, D))`` noise
        draw (mmca_et.py:66-93)."""
        W = np.asarray(model_params['W']).T                 # (H, D)
        s = np.asarray(my_hdata['s'])
        my_N = s.shape[0]
        t0 = s[:, :, None] * W[None, :, :]                  # (N, H, D) "stacked" datapoints
        idx = np.argmax(np.abs(t0), axis=1)                 # first maximum, like the reference's argmax
        y = np.take_along_axis(t0, idx[:, None, :], axis=1)[:, 0, :].astype(np.float64)
        y += np.random.normal(scale=model_params['sigma'], size=(my_N, W.shape[1]))
        return {'y': y, 's': my_hdata['s']}

    # ------------------------------------------------------------------ plumbing
    def _rho(self, T):
        T_rho = np.maximum(T, self.rho_T_bound)
        rho = 1. / (1. - 1. / T_rho)
        return float(np.maximum(np.minimum(rho, self.rho_ubound), self.rho_lbound))

    # ------------------------------------------------------------------ hot path
    @tracing.traced
    def select_Hprimes(self, model_params, data):
        """``data['candidates']`` (N, Hprime): the latents with the smallest |W_h - y|^2, ascending
        (mmca_et.py:96-124)."""
        res = self._resident(data['y'])
        Y = res["Y"]
        N = Y.shape[0]
        H, Hp = self.H, self.Hprime
        self._masks()
        Wt = self._upload("mca_W", np.ascontiguousarray(np.asarray(model_params['W'], dtype=np.float64).T))
        cand = torch.empty((N, Hp), dtype=torch.int32, device=self.device)
        if N:
            A = self._gemm_nt(Y, Wt, self._buf("mmca_sel_scores", (N, H)), "select_gemm")
            G = self._gemm_nt(Wt, Wt, self._buf("mmca_gram", (H, H)), "gram_gemm")
            self._call("select", "pm_bsc_select_estep_f64", _ptr(A), H, _ptr(G), _ptr(res["ynorm2"]), None, None,
                       None, None, None, 0, self.gamma, None, N, H, Hp, 1 | 4 | 16, _ptr(cand), None, 0, None,
                       self._stream())
        data['candidates'] = DeviceArray(cand, np.int64)
        return data

    def _finalize(self, stats, model_params, par, A_pi_gamma, B_pi_gamma):
        """W update with inertia and the scalars (mmca_et.py:365-427), one device->host copy."""
        H, D = self.H, self.D
        pies, sigma = model_params['pi'], model_params['sigma']
        HD = H * D
        G1 = stats[:HD].view(H, D)
        Wp_m = stats[HD:2 * HD].view(H, D)
        Wq_m = stats[2 * HD:3 * HD].view(H, D)
        q1sum = stats[3 * HD:3 * HD + H]
        scal = stats[3 * HD + H:3 * HD + H + 4]
        parts = [scal]
        learn_W = 'W' in self.to_learn
        if learn_W:
            tracing.tracepoint("M_step:update W")
            Wp = G1 + Wp_m                               # singletons weigh y by q alone (mmca_et.py:300-301)
            Wq = q1sum[:, None] + Wq_m
            Wq = torch.clamp(Wq, min=self.tol)           # make sure we do not divide by zero (mmca_et.py:378-379)
            W_new = Wp / Wq
            inertia = torch.clamp(1. - torch.exp(-Wq / 2.5), min=0.2)      # mmca_et.py:385-387
            W_new = inertia * W_new + (1 - inertia) * par["Wt"]
            parts.append(W_new.reshape(-1))
        flat = torch.cat(parts)
        host = self._download(flat) if flat.is_cuda else flat.numpy()
        my_pi, my_sigma, ldenom_sum, N_use = float(host[0]), float(host[1]), float(host[2]), int(round(host[3]))
        dlog.append('N_use', N_use)

        W_new = host[4:4 + HD].reshape(H, D).T.copy() if learn_W else np.asarray(model_params['W'])
        if 'pi' in self.to_learn:
            tracing.tracepoint("M_step:update pi")
            pi_new = A_pi_gamma / B_pi_gamma * pies * my_pi / N_use
        else:
            pi_new = pies
        if 'sigma' in self.to_learn:
            tracing.tracepoint("M_step:update sigma")
            sigma_new = np.sqrt(my_sigma / D / N_use)
        else:
            sigma_new = sigma
        lAi = (H * np.log(1. - pi_new)) - ((D / 2) * np.log(2 * _PI)) - (D * np.log(sigma_new))
        loglike_et = (lAi * N_use) + ldenom_sum
        return {'W': W_new, 'pi': pi_new, 'sigma': sigma_new, 'Q': loglike_et}
