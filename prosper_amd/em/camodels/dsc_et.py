"""Discrete Sparse Coding on the MI355X: drop-in for prosper/em/camodels/dsc_et.py.

Same constructor (``states`` array incl. 0), ``select_Hprimes / E_step / M_step`` signatures, return
keys (``W, pi, sigma, Q``), state tables (``state_matrix`` in itertools.product order,
``single_state_matrix``, ``state_abs``) and ``dlog`` side effects (``prior_mass``, ``L``, ``N_use``) as the
reference's ``DSC_ET`` (dsc_et.py:96-925).  Kernels (prosper_amd/csrc/dsc_kernels.hip):

  select_Hprimes  scores GEMM a = Y.W^T (f64 MFMA) -> per-latent best singleton log-joint ->
                  16-lane selection kernel, best first
  E_step          all energies from a and the Gram matrix (no D-length work per state)
  M_step          E[s] rows + Wp = E[s]^T.Y (f64 MFMA), E[s s^T] scatter, value counts; ONE all-reduce of
                  the packed statistics; the H x H solve on the device (pm_spd_inverse_f64)
"""
import ctypes
import itertools as itls

import numpy as np
from scipy.special import gammaln

from ._device import DeviceCAModel, DeviceArray, _ptr, small_blas
from ... import _lib
from ...utils import parallel
from ...utils import tracing
from ...utils.datalog import dlog

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


_LOG_UNDERFLOW = -745.1332191019412      # log(2^-1075): exp() of anything below rounds to 0.0


def multinom2(n, k):
    """Multinomial coefficient n! / prod k_i! (dsc_et.py:23-39)."""
    return np.exp(gammaln(n + 1) - gammaln(k + 1).sum())


def get_states(states, Hprime, gamma):
    """All Hprime-vectors over ``states`` with 2..gamma non-zeros, itertools.product order (dsc_et.py:56-63)."""
    assert len(states.shape) == 1
    sl = [np.array(c) for c in itls.product(states, repeat=Hprime)
          if (np.sum(np.array(c) != 0) <= gamma and np.sum(np.array(c) != 0) > 1)]
    return np.array(sl) if sl else np.zeros((0, Hprime))


def generate_state_matrix(Hprime, gamma, states=np.array([0, 1])):
    """(no_states, state_matrix, state_abs) as dsc_et.py:66-93."""
    state_matrix = get_states(np.asarray(states), Hprime, gamma)
    return state_matrix.shape[0], state_matrix, (state_matrix != 0).sum(axis=1)


class DSC_ET(DeviceCAModel):
    """Discrete Sparse Coding (K-ary latents, linear superposition) with Expectation Truncation."""

    def __init__(self, D, H, Hprime, gamma, states=np.array([-1., 0., 1.]), to_learn=['W', 'pi', 'sigma'],
                 comm=parallel.COMM_WORLD, device=None):
        DeviceCAModel.__init__(self, D, H, Hprime, gamma, to_learn, comm, device)
        if not type(states) == np.ndarray:
            raise TypeError("DSC: states must be of type numpy.ndarray")
        if Hprime > H:
            raise Exception("Hprime must be less or equal to H")
        if gamma > Hprime:
            raise Exception("gamma must be less or equal to Hprime")
        self.states = states
        self.K = self.states.shape[0]
        self._K_0 = int(np.argwhere(states == 0.)[0, 0])
        if self.K > 8:
            raise _lib.HipError("DSC_ET: at most 8 latent values (PM_DSC_MAX_K)")
        tol = 1e-5
        self._noise_policy = {
            'W': (-np.inf, +np.inf, False),
            'pi': (tol, 1. - tol, False),
            'sigma': (0., +np.inf, False),
        }
        # state tables (dsc_et.py:167-191)
        ss = np.empty((0, self.H), dtype=np.int8)
        for i in range(self.K):
            if i == self._K_0:
                continue
            ss = np.concatenate((ss, np.eye(self.H, dtype=np.int8) * states[i]))
        self.single_state_matrix = ss[np.sum(np.abs(np.sign(ss)), 1) == 1]
        self._build_state_tables()
        self._tab = None

    def _build_state_tables(self):
        """state_matrix / no_states / state_abs (K, S) for the current Hprime, gamma (dsc_et.py:177-191)."""
        self.state_matrix = get_states(self.states, self.Hprime, self.gamma)
        self.no_states = self.state_matrix.shape[0]
        self.state_abs = np.empty((self.K, self.no_states))
        for i in range(self.K):
            self.state_abs[i, :] = (self.state_matrix == self.states[i]).sum(axis=1)
        self.state_abs[self._K_0, :] = self.H - self.state_abs.sum(0) + self.state_abs[self._K_0, :]

    # ------------------------------------------------------------------ reference-shaped helpers
    def check_params(self, model_params):
        """Finite parameters, sigma >= 0 (dsc_et.py:194-236)."""
        assert np.isfinite(model_params['W']).all()
        assert np.isfinite(model_params['pi']).all()
        assert np.isfinite(model_params['sigma']).all()
        assert model_params['sigma'] >= 0.
        return model_params

    def _draw_latents(self, model_params, my_N, g):
        """Every latent takes value states[k] with probability pi[k] (dsc_et.py:277-292)."""
        pi = torch.as_tensor(np.asarray(model_params['pi'], dtype=np.float64)).to(self.device)
        vals = torch.as_tensor(np.asarray(self.states, dtype=np.float64)).to(self.device)
        idx = torch.multinomial(pi / pi.sum(), my_N * self.H, replacement=True, generator=g)
        return vals[idx].view(my_N, self.H)

    def generate_data(self, model_params, my_N, noise_on=True, gs=None, gp=None, device=False, seed=None):
        """Latents drawn per datapoint with ``np.random.choice(states, H, p=pi)``, y = s.W^T (+ noise);
        RNG stream as upstream (dsc_et.py:238-299).  ``device=True``: drawn on the GPU (DeviceCAModel.generate_data)."""
        if device:
            if gs is not None or gp is not None:
                raise ValueError("generate_data(device=True) draws its own latents: gs / gp need the host generator")
            return self.generate_data_device(model_params, my_N, seed, noise_on=noise_on)
        D, H, states = self.D, self.H, self.states
        pi = model_params['pi']
        W = model_params['W'].T
        y = np.zeros((my_N, D))
        s = np.zeros((my_N, H), dtype=np.int8)
        for n in range(my_N):
            if gs is None:
                s[n] = np.random.choice(states, size=H, replace=True, p=pi)
            else:
                assert gs.shape[0] == my_N
                if gp is None:
                    assert len(gs.shape) == 2
                    s[n] = gs[n]
                else:
                    assert gp.shape[0] == my_N
                    assert gp.shape[1] == gs.shape[1]
                    s[n] = (gs[n] * gp[n]).sum(0)
        y = s.astype(np.float64) @ W
        if noise_on:
            y += np.random.normal(scale=model_params['sigma'], size=(my_N, D))
        return {'y': y, 's': s}

    def noisify_params(self, model_params, anneal):
        """Parameter noise under ``_noise_policy``; pi gets uniform noise and is renormalised
        (dsc_et.py:412-490)."""
        comm = self.comm
        for param, policy in self._noise_policy.items():
            pvalue = model_params[param]
            if (not param + '_noise' == 'pi_noise') and anneal[param + "_noise"] != 0.0:
                if np.isscalar(pvalue):
                    new_pvalue = 0
                    if comm.rank == 0:
                        new_pvalue = pvalue + np.random.normal(scale=anneal[param + "_noise"])
                        if new_pvalue < policy[0]:
                            new_pvalue = policy[0]
                        if new_pvalue >= policy[1]:
                            new_pvalue = policy[1]
                        if policy[2]:
                            new_pvalue = np.abs(new_pvalue)
                    pvalue = comm.bcast(new_pvalue)
                else:
                    if comm.rank == 0:
                        new_pvalue = pvalue + np.random.normal(scale=anneal[param + "_noise"], size=pvalue.shape)
                        low_bound, up_bound, absify = policy
                        new_pvalue = np.minimum(up_bound, np.maximum(low_bound, new_pvalue))
                        if absify:
                            new_pvalue = np.abs(new_pvalue)
                        pvalue = new_pvalue
                    comm.Bcast(pvalue)
            elif param + '_noise' == 'pi_noise' and anneal["pi_noise"] != 0.0:
                if comm.rank == 0:
                    new_pvalue = pvalue + np.random.rand(*pvalue.shape) * anneal["pi_noise"]
                    pvalue = new_pvalue / new_pvalue.sum()
                comm.Bcast(pvalue)
            model_params[param] = pvalue
        return model_params

    def get_scaling_factors(self, pi):
        """Prior mass of the states with at most gamma non-zero latents (dsc_et.py:798-823).  The count vectors and their
        multinomial coefficients do not depend on pi: enumerated once (the loop and its summation order are upstream's)."""
        key = (self.gamma, self.H, self._K_0, len(self.states))
        cached = getattr(self, "_scaling_terms", None)
        terms = cached[1] if cached is not None and cached[0] == key else None
        if terms is None:
            terms = []
            for gp in itls.product(np.arange(self.gamma + 1), repeat=len(self.states) - 1):
                ngp = np.array(gp)
                if ngp.sum() > self.gamma:
                    continue
                abs_array = np.insert(ngp, self._K_0, self.H - ngp.sum())
                if not abs_array.sum() == self.H:
                    raise Exception("wrong number of elements counted")
                terms.append((multinom2(abs_array.sum(), abs_array), abs_array))
            self._scaling_terms = (key, terms)
        A_pi_gamma = 0.0
        for coef, abs_array in terms:
            A_pi_gamma += coef * np.prod(pi ** abs_array)
        return A_pi_gamma

    def standard_init(self, data):
        """W = data mean + N(0, (sigma_init/4)^2), pi = (1 - 1/H) on the zero value and a random split of
        1/H over the others (dsc_et.py:872-925; RNG order: normal((D,H)) then rand(K-1))."""
        W_mean, sigma_sq = self._data_moments(data)
        D = W_mean.shape[0]
        assert D == self.D
        sigma_init = np.sqrt(sigma_sq).sum() / D
        W_init = W_mean[:, None] + np.random.normal(scale=sigma_init / 4., size=[D, self.H])
        sparsity = 1. - (1. / self.H)
        pi_init = np.random.rand(self.K - 1)
        pi_init = (1 - sparsity) * pi_init / pi_init.sum()
        pi_init = np.insert(pi_init, self._K_0, sparsity)
        return {'W': W_init, 'pi': pi_init, 'sigma': sigma_init}

    def free_energy(self, model_params, my_data):
        return 0.0

    def gain(self, old_parameters, new_parameters):
        return 0.0

    # ------------------------------------------------------------------ plumbing
    def _tables(self):
        """Device state table: (S, Hprime) uint8 indices into ``states``."""
        key = (self.Hprime, self.gamma, self.no_states)
        if self._tab is None or self._tab[0] != key:
            lib = _lib.load()
            if not lib.pm_bsc_rows16_supported(self.H, self.Hprime, 0):
                raise _lib.HipError("DSC_ET: H = %d is outside the selection kernel's range (H <= 512)" % self.H)
            idx = np.zeros((max(self.no_states, 1), self.Hprime), dtype=np.uint8)
            for k in range(self.K):
                idx[:self.no_states][self.state_matrix == self.states[k]] = k
            self._tab = (key, torch.from_numpy(idx).to(self.device))
        return self._tab[1]

    def _prior(self, pi):
        """pre_F (dsc_et.py:539-558)."""
        H, K, K0 = self.H, self.K, self._K_0
        pre_F = np.empty(1 + (K - 1) * H + self.no_states)
        l_pis = np.zeros(self.no_states)
        for i in range(K):
            l_pis += self.state_abs[i] * np.log(pi[i])
        pre_F[0] = H * np.log(pi[K0])
        c = 0
        for state in range(K):
            if state == K0:
                continue
            pre_F[c * H + 1:(c + 1) * H + 1] = np.log(pi[state]) + ((H - 1) * np.log(pi[K0]))
            c += 1
        pre_F[(K - 1) * H + 1:] = l_pis
        return pre_F

    def _params(self, anneal, pi, sigma):
        beta = 1. / anneal['T']
        pre1 = -1. / 2. / sigma / sigma
        P = _lib.DscParams(K=self.K, K0=self._K_0, pre1=float(pre1), ecoef=float(beta * pre1),
                           pscale=float(beta if anneal['anneal_prior'] else 1.0))
        with np.errstate(divide='ignore'):
            lp = np.log(np.asarray(pi, dtype=np.float64))
        for k in range(self.K):
            P.values[k] = float(self.states[k])
            P.logpi[k] = float(lp[k])
        return P

    def _params_dev(self, W, res):
        """Device copy of W^T (H,D), the Gram matrix and the scores for the current W and data."""
        return self._scores_params(W, res)

    # ------------------------------------------------------------------ hot path
    @tracing.traced
    def select_Hprimes(self, model_params, data):
        """``data['candidates']`` (N, Hprime): latents ranked by their best singleton log-joint, best
        first (dsc_et.py:347-410)."""
        res = self._resident(data['y'])
        N = res["Y"].shape[0]
        H, Hp = self.H, self.Hprime
        self._tables()
        par = self._params_dev(model_params['W'], res)
        P = self._params(FixedT(), model_params['pi'], model_params['sigma'])
        cand = torch.empty((N, Hp), dtype=torch.int32, device=self.device)
        if N and _lib.load().pm_xsc_select_supported(H, Hp, 0):
            # ranking values formed and ranked in one pass over the scores (no (N, H) buffer, one launch)
            self._call("select", "pm_xsc_select_f64", _ptr(par["A"]), H, _ptr(par["G"]), ctypes.byref(P), N, H, Hp,
                       _ptr(cand), self._stream())
        elif N:
            R = self._buf("dsc_sel", (N, H))
            self._call("select_scores", "pm_dsc_select_scores_f64", _ptr(par["A"]), H, _ptr(par["G"]),
                       ctypes.byref(P), N, H, _ptr(R), H, self._stream())
            self._call("select", "pm_bsc_select_estep_f64", _ptr(R), H, _ptr(par["G"]), _ptr(res["ynorm2"]), None,
                       None, None, None, None, 0, self.gamma, None, N, H, Hp, 1 | 4 | 8, _ptr(cand), None, 0, None,
                       self._stream())
        data['candidates'] = DeviceArray(cand, np.int64)
        return data

    @tracing.traced
    def E_step(self, anneal, model_params, my_data):
        """Log-pseudo-joints ``{'logpj': (N, 1 + (K-1)H + S)}`` (dsc_et.py:492-585)."""
        res = self._resident(my_data['y'])
        N = res["Y"].shape[0]
        H, Hp, S = self.H, self.Hprime, self.no_states
        tab = self._tables()
        par = self._params_dev(model_params['W'], res)
        cand = self._device_candidates(my_data['candidates'], N)
        P = self._params(anneal, model_params['pi'], model_params['sigma'])
        prior = self._upload("dsc_prior", self._prior(np.asarray(model_params['pi'], dtype=np.float64)))
        Kt = 1 + (self.K - 1) * H + S
        tracing.tracepoint("E_step:iterating")
        return {'logpj': self._dsc_estep(anneal, "dsc_stats", par, res, cand, tab, S, prior, P, Kt, model_params['pi'])}

    @tracing.traced
    def M_step(self, anneal, model_params, my_suff_stat, my_data):
        """New W, pi, sigma (dsc_et.py:587-774).  Logs ``prior_mass``, ``L`` and ``N_use``."""
        comm = self.comm
        H, Hp, D, S, K = self.H, self.Hprime, self.D, self.no_states, self.K
        pi = np.asarray(model_params['pi'], dtype=np.float64)
        sigma = model_params['sigma']
        res = self._resident(my_data['y'])
        Y = res["Y"]
        my_N = Y.shape[0]
        tab = self._tables()
        cand = self._device_candidates(my_data['candidates'], my_N)
        Kt = 1 + (K - 1) * H + S

        logpj = my_suff_stat['logpj']
        if isinstance(logpj, DeviceArray) and getattr(logpj, "lse", None) is not None:
            lp, lse = logpj.tensor, logpj.lse
        else:
            lp = torch.from_numpy(np.ascontiguousarray(np.asarray(logpj), dtype=np.float64)).to(self.device)
            lse = torch.logsumexp(lp, dim=1)
        lp, lse = lp.contiguous(), lse.contiguous()
        assert tuple(lp.shape) == (my_N, Kt)
        N = self._global_count(res, my_N)

        A_pi_gamma = self.get_scaling_factors(pi)
        dlog.append("prior_mass", A_pi_gamma)

        # data truncation (dsc_et.py:825-843): keep the datapoints STRICTLY above the N_use-th largest evidence
        lse_cut, cut_dev = float("-inf"), None
        if anneal['Ncut_factor'] > 0.0:
            tracing.tracepoint("M_step:truncating")
            N_use = int(N * (1 - (1 - A_pi_gamma) * anneal['Ncut_factor'])) or N    # (0: upstream's allsort(...)[-0] keeps everything)
            # the reference cuts on un-stabilised sums of exp(logpj), which are exactly 0 below the
            # underflow boundary: with the cut among those only strictly positive sums survive
            if lse.is_cuda and my_N:      # (the cut stays on the device: the row pass reads it there)
                cut_dev = torch.clamp_min(self._kth_select_dev(lse, N_use), _LOG_UNDERFLOW)
                lse_cut = float("nan")    # (not -inf: statistics a fused E-step pass may have left do not apply)
            else:
                lse_cut = max(self._kth_largest_global(lse, N_use), _LOG_UNDERFLOW)

        tracing.tracepoint("M_step:iterating")
        lib = _lib.load()
        n_stats = lib.pm_dsc_stats_len(H, D)
        P = self._params(anneal, pi, sigma)
        fused = self._dsc_fused_stats(logpj, res, cand, P, pi, lse_cut) if my_N else None
        stats = fused["stats"] if fused else self._buf("dsc_stats", (n_stats,))
        if not fused:
            stats.zero_()
        expect = self._buf("expect", (my_N, H))
        # (the fused pass has used the prior already; only the M-step's own row pass needs it again)
        prior = None if fused else self._upload("dsc_prior", self._prior(pi))
        if my_N:
            self._rows_and_wp((_ptr(lp), Kt, _ptr(lse), ctypes.c_double(lse_cut), _ptr(cand), _ptr(tab), S,
                               _ptr(prior) if prior is not None else None,
                               ctypes.byref(P), my_N, H, D, Hp, _ptr(expect), H, _ptr(stats)),
                              Kt, expect, Y, stats, my_N, self.K, int(P.flags), Hp, S, fused=fused, cut_dev=cut_dev)
        comm.allreduce_device(stats)      # replaces dsc_et.py:648,738,739,747,769 and the allreduce in get_likelihood
        self._mstep_res = res
        return self._finalize(stats, model_params)

    def _finalize(self, stats, model_params):
        """Parameter updates from the all-reduced statistics (dsc_et.py:736-774), one device->host copy."""
        H, D, K, K0 = self.H, self.D, self.K, self._K_0
        pi = np.asarray(model_params['pi'], dtype=np.float64)
        sigma = model_params['sigma']
        o_wq, o_qd = H * D, H * D + H * H
        o_cnt = o_qd + H
        Wp = stats[:o_wq].view(H, D)
        Wq_u = stats[o_wq:o_qd].view(H, H)
        qdiag = stats[o_qd:o_cnt]
        parts = [stats[o_cnt:o_cnt + 8 + 4]]
        learn_W = 'W' in self.to_learn
        Wq = None
        if learn_W:
            tracing.tracepoint("M_step:update W")
            X, status, Wq = self._solve_normal_eq(Wq_u, qdiag, Wp.contiguous())
            parts += [status, X.reshape(-1)]
        flat = torch.cat(parts)
        self._seed_rec = None
        res = getattr(self, "_mstep_res", None)
        if flat.is_cuda and learn_W and res is not None and self.speculate:
            host = self._download(flat, then=lambda: self._seed_next(res, X))
        else:
            host = self._download(flat) if flat.is_cuda else flat.numpy()
        cnt = host[:8]
        my_sigma, Fs, N_use = float(host[8]) / D, float(host[9]), int(round(host[10]))

        # (a cut that keeps NO datapoint -- upstream's strict '>' on two or three of them: dsc_et.py:832 -- divides by zero the
        # NumPy way upstream: nan / inf and a warning, no exception)
        Nf = np.float64(N_use)
        with np.errstate(divide='ignore', invalid='ignore'):
            L = -0.5 * D * np.log(2 * np.pi * sigma ** 2) + np.float64(Fs) / Nf          # dsc_et.py:845-870
        dlog.append('L', L)

        W = np.asarray(model_params['W'])
        if learn_W:
            ok = self._solve_ok(float(host[12]), float(host[13]))
            redo = self._solve_accurate(float(host[14])) if ok else None
            if redo is not None:    # the device rejected the inverse's warm start: W from the refined solve, seed void
                self._seed_rec = None
                W_new = redo
            elif ok:
                W_new = host[15:15 + H * D].reshape(H, D).copy()
                if self._seed_rec is not None:
                    self._seed_rec["W"] = W_new.copy().transpose()   # private snapshot of the W handed back (same memory order: a
                                                                     # contiguous copy and a contiguous comparison)
            else:   # numerically singular Wq: the reference's own LAPACK lstsq on the host
                self._seed_rec = None
                self._winv_prev = None        # never warm-start the next inverse from a rejected one
                with small_blas():
                    W_new = np.linalg.lstsq(Wq.cpu().numpy(), Wp.cpu().numpy(), rcond=None)[0]
            W_out = W_new.transpose()
        else:
            W_out = W

        if 'pi' in self.to_learn:
            tracing.tracepoint("M_step:update pi")
            my_pi = np.zeros(K)
            for k in range(K):
                if k != K0:
                    my_pi[k] = cnt[k]
            my_pi[K0] = H * N_use - my_pi.sum()       # every kept datapoint distributes H latents over the values
            pi_new = my_pi / my_pi.sum()
            eps = 1e-6
            if np.any(pi_new < eps):
                which_lo = pi_new < eps
                which_hi = pi_new >= eps
                pi_new[which_lo] += eps - pi_new[which_lo]
                pi_new[which_hi] -= (eps * np.sum(which_lo)) / np.sum(which_hi)
            if 'penalty' in list(self.__dict__.keys()):
                if self.penalty > pi_new[K0]:
                    r = (1 - self.penalty) / (1 - pi_new[K0])
                    pi_new[pi_new != 0] = pi_new[pi_new != 0] * r
                    pi_new[K0] = self.penalty
                    pi_new /= pi_new.sum()
        else:
            pi_new = pi

        if 'sigma' in self.to_learn:
            tracing.tracepoint("M_step:update sigma")
            with np.errstate(divide='ignore', invalid='ignore'):
                sigma_new = np.sqrt(np.float64(my_sigma) / Nf)
        else:
            sigma_new = sigma

        dlog.append('N_use', N_use)
        return {'W': W_out, 'pi': pi_new, 'sigma': sigma_new, 'Q': 0.}

    def inference(self, anneal, model_params, test_data, topK=10, logprob=False, adaptive=True,
                  Hprime_max=None, gamma_max=None):
        """Top-K posterior states and marginals per datapoint (dsc_et.py:927-1059); same return dict
        (``s`` (N,topK,H) int8 with the latent VALUES, ``m`` (N,H), ``p`` (N,topK), ``gamma``, ``Hprime``).

        Upstream behaviour that is reproduced as it is: ``p`` without ``logprob`` is exp(logpj - max);
        re-run datapoints keep earlier entries of ``s``; the marginal of a candidate combines its FIRST
        non-zero value's singleton state with the multi-cause states in which it takes the value 1
        (:1010-1016); and in the adaptive re-runs ``state_abs`` is the 1-D count of non-zeros
        (``generate_state_matrix``, :1048), so every multi-cause state gets the same prior there.  Unlike
        upstream the proper (K, S) ``state_abs`` is restored afterwards (upstream leaves the 1-D one behind)."""
        assert 'y' in test_data, "Key 'y' in test_data dict not defined."
        model_params = self.check_params(model_params)
        comm = self.comm
        my_y = test_data['y']
        if isinstance(my_y, DeviceArray):
            my_y = my_y.tensor
        my_N, D = my_y.shape
        H, K = self.H, self.K
        nss = (K - 1) * H
        Hprime_start, gamma_start = self.Hprime, self.gamma
        if topK == -1:
            topK = self.state_matrix.shape[0]
        dev = self.device
        res_s = torch.zeros((my_N, topK, H), dtype=torch.int8, device=dev)
        res_m = torch.zeros((my_N, H), dtype=torch.float64, device=dev)
        res_p = torch.zeros((my_N, topK), dtype=torch.float64, device=dev)
        res_gamma = torch.zeros((my_N,), dtype=torch.float64, device=dev)
        res_Hprime = torch.zeros((my_N,), dtype=torch.float64, device=dev)
        nz_vals = torch.tensor([self.states[k] for k in range(K) if k != self._K_0], dtype=torch.int8, device=dev)

        cur_y = my_y
        which = torch.ones(my_N, dtype=torch.bool, device=dev)
        try:
            while bool(which.any()):
                ind_n = torch.nonzero(which).flatten()
                logpj, cand = self.compute_lpj(anneal, model_params, {'y': cur_y})
                lp = logpj.tensor if isinstance(logpj, DeviceArray) else torch.as_tensor(np.asarray(logpj)).to(dev)
                cd = (cand.tensor if isinstance(cand, DeviceArray) else torch.as_tensor(np.asarray(cand)).to(dev)).long()
                n_cur, Kt = lp.shape
                Hp = self.Hprime
                k_eff = min(topK, Kt)
                # normalisation, top-K columns and the marginals (:983-1016): one HIP pass over the rows
                # (pm_infer_topk_cols_f64; the marginal of a candidate combines its FIRST non-zero value's one-cause state
                # with the multi-cause states in which it takes the value 1: the masks mark exactly those)
                SMh = self.state_matrix if self.no_states else np.zeros((1, Hp))
                mk = ((SMh == 1).astype(np.int64) << np.arange(SMh.shape[1])[None, :]).sum(axis=1).astype(np.uint16)
                masks_d = torch.from_numpy(mk.view(np.int16).copy()).to(dev)
                lp = lp.contiguous() if lp.stride(1) != 1 else lp
                cd32 = cd.to(torch.int32).contiguous()
                top_idx32 = torch.empty((n_cur, k_eff), dtype=torch.int32, device=dev)
                top_val = torch.empty((n_cur, k_eff), dtype=torch.float64, device=dev)
                top_rel = torch.empty((n_cur, k_eff), dtype=torch.float64, device=dev)
                m_blk = torch.empty((n_cur, H), dtype=torch.float64, device=dev)
                self._call("infer_topk", "pm_infer_topk_cols_f64", _ptr(lp), lp.stride(0), _ptr(cd32), _ptr(masks_d), n_cur, H,
                           Hp, self.no_states, nss, k_eff, _ptr(top_idx32), _ptr(top_val), _ptr(top_rel), _ptr(m_blk), H,
                           self._stream())
                if bool((top_idx32 < 0).any()):
                    raise _lib.HipError("inference: non-finite log-joints (NaN) in %d datapoint(s)"
                                        % int((top_idx32 < 0).any(dim=1).sum()))
                top_idx = top_idx32.long()
                res_Hprime[ind_n] = float(self.Hprime)
                res_gamma[ind_n] = float(self.gamma)
                SM = torch.from_numpy(self.state_matrix.astype(np.int8)).to(dev) if self.no_states else \
                    torch.zeros((1, Hp), dtype=torch.int8, device=dev)
                s_blk = res_s[ind_n, :k_eff].clone()
                single = (top_idx >= 1) & (top_idx <= nss)
                if bool(single.any()):
                    nn_, mm_ = torch.nonzero(single, as_tuple=True)
                    si = top_idx[nn_, mm_] - 1
                    s_blk[nn_, mm_, si % H] = nz_vals[si // H]
                multi = top_idx > nss
                if bool(multi.any()):
                    nn_, mm_ = torch.nonzero(multi, as_tuple=True)
                    rows = SM[top_idx[nn_, mm_] - nss - 1]                           # (M, Hp) latent values
                    s_blk[nn_[:, None].expand(-1, Hp), mm_[:, None].expand(-1, Hp), cd[nn_]] = rows
                res_s[ind_n, :k_eff] = s_blk
                res_p[ind_n, :k_eff] = top_val if logprob else torch.exp(top_rel)
                res_m[ind_n] = m_blk
                if not adaptive:
                    break
                which = ((res_s[:, 0, :] != 0).sum(-1) == self.gamma)
                if not bool(which.any()):
                    break
                if (Hprime_max is not None and self.Hprime == Hprime_max) and \
                        (gamma_max is not None and self.gamma == gamma_max):
                    break
                cur_y = my_y[which.cpu().numpy()] if not torch.is_tensor(my_y) else my_y[which]
                print("Rank %i: For %i data points MAP state has activity equal to gamma." % (comm.rank, int(which.sum())))
                if not ((self.Hprime == self.H) or (Hprime_max is not None and self.Hprime == Hprime_max)):
                    self.Hprime += 1
                if (self.gamma == self.H) or (gamma_max is not None and self.gamma == gamma_max):
                    continue
                self.gamma += 1
                print("Rank %i: Updating state matrix and running again." % comm.rank)
                self.no_states, self.state_matrix, self.state_abs = generate_state_matrix(self.Hprime, self.gamma,
                                                                                           self.states)
        finally:
            self.Hprime, self.gamma = Hprime_start, gamma_start
            self._build_state_tables()
        m_out = res_m if logprob else torch.exp(res_m)
        return {'s': res_s.cpu().numpy(), 'm': m_out.cpu().numpy(), 'p': res_p.cpu().numpy(),
                'gamma': res_gamma.cpu().numpy(), 'Hprime': res_Hprime.cpu().numpy()}

    def calculate_respons(self, anneal, model_params, data):
        """Posterior over the truncated states (dsc_et.py:776-784)."""
        cand = np.sort(np.asarray(data['candidates']), axis=1)
        data['candidates'] = cand
        F = np.asarray(self.E_step(anneal, model_params, data)['logpj'])
        e = np.exp(F - F.max(axis=1)[:, None])
        return e / e.sum(axis=1).reshape(-1, 1)


class FixedT(dict):
    """Annealing stand-in for calls that need no temperature (selection)."""

    def __missing__(self, k):
        return 1.0 if k == 'T' else 0.0
