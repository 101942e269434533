"""Ternary Sparse Coding on the MI355X: the model of prosper/em/camodels/tsc_et.py.

Latents in {-1, 0, +1} with prior pi/2, 1-pi, pi/2 (scalar ``pi``), linear superposition, Gaussian noise.
Every truncated state lives in the H' candidate positions: the state table holds the null state and the
one-cause states too, ``logpj`` has one column per table row.  Upstream's class cannot be constructed
(``states`` is undefined in tsc_et.py:131); this one can, with the same constructor signature,
``select_Hprimes / E_step / M_step`` signatures, return keys and ``dlog`` side effects (``L``, ``N_use``), and it
reproduces what upstream's methods compute (checked against goldens minted by running them on an object
built without ``__init__``), including two behaviours a caller can observe:

  * candidates are the latents of the H' best one-cause STATES, so a latent can appear twice (tsc_et.py:208-211);
  * for a repeated candidate only its LAST position contributes to the W update (NumPy fancy-index ``+=``,
    tsc_et.py:471-475); pi and sigma see every position.

Kernels: the scores GEMM, ``pm_tsc_select_scores_f64`` + the 16-lane selection kernel over the 2H one-cause
states, and the DSC kernels (csrc/dsc_kernels.hip) with PM_DSC_TABLE_ONLY | PM_DSC_LAST_POSITION.
"""
import ctypes
import itertools as itls

import numpy as np
from scipy.special import comb

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


def generate_state_matrix(Hprime, gamma, H, states):
    """(single_state_matrix, state_matrix, no_states, state_abs) as tsc_et.py:23-80 -- incl. ``no_states`` =
    len(states)**Hprime (the size of the untruncated table) and ``state_abs`` over that untruncated table."""
    ss = np.concatenate([np.eye(H, dtype=np.int8) * v for v in states if v != 0])
    single_state_matrix = ss[np.sum(np.abs(ss), 1) == 1]
    s = np.array(list(itls.product(np.array(states), repeat=Hprime)), dtype=np.int8)
    states_abs = np.empty((len(states), s.shape[0]))
    for i in range(len(states)):
        states_abs[i, :] = (s == states[i]).sum(axis=1)
    state_matrix = s[np.sum(np.abs(s), axis=1) <= gamma]
    return single_state_matrix, state_matrix, s.shape[0], states_abs


class TSC_ET(DeviceCAModel):
    """Ternary Sparse Coding with Expectation Truncation."""

    def __init__(self, D, H, Hprime, gamma, to_learn=['W', 'pi', 'sigma'], comm=parallel.COMM_WORLD, device=None):
        DeviceCAModel.__init__(self, D, H, Hprime, gamma, to_learn, comm, device)
        self.states = np.array([-1., 0., 1.])
        (self.single_state_matrix, self.state_matrix, self.no_states,
         self.state_abs) = generate_state_matrix(Hprime, gamma, H, self.states)
        tol = 1e-5
        self.noise_policy = {
            'W': (-np.inf, +np.inf, False),
            'pi': (tol, 1. - tol, False),
            'sigma': (0., +np.inf, False),
        }
        self._tab = None

    def _draw_latents(self, model_params, my_N, g):
        pi = float(model_params['pi'])
        p = torch.rand((my_N, self.H), generator=g, device=self.device, dtype=torch.float64)
        one = torch.ones((), dtype=torch.float64, device=self.device)
        return torch.where(p < pi / 2, -one, torch.where(p < pi, one, 0 * one))

    def _generate_data_host(self, model_params, my_N):
        """s_h = -1 / +1 / 0 for p < pi/2, p < pi, else; y = s.W^T + noise.  RNG stream as upstream
        (tsc_et.py:215-275): ``random(H)`` per datapoint, then one ``normal((my_N, D))``."""
        pi = model_params['pi']
        W = model_params['W'].T
        p = np.random.random(size=(my_N, self.H))
        s = np.where(p < pi / 2, -1, np.where(p < pi, 1, 0)).astype(np.int8)
        y = s.astype(np.float64) @ W
        y += np.random.normal(scale=model_params['sigma'], size=(my_N, self.D))
        return {'y': y, 's': s}

    def inference(self, anneal, model_params, test_data, topK=10, logprob=False, abs_marginal=True,
                  adaptive=True, Hprime_max=None, gamma_max=None):
        """Top-K posterior states, signed marginal ``m`` and absolute marginal ``am`` per datapoint
        (tsc_et.py:546-680); same return dict.  As upstream: ``p`` is the normalised probability (its log with
        ``logprob``), a repeated candidate's LAST position wins the writes into ``s`` / ``m`` / ``am``, re-run
        datapoints keep earlier entries, and datapoints whose best state has exactly gamma non-zeros are re-run
        with Hprime+1 / gamma+1."""
        assert 'y' in test_data, "Key 'y' in test_data dict not defined."
        comm = self.comm
        my_y = test_data['y']
        if isinstance(my_y, DeviceArray):
            my_y = my_y.tensor
        my_N, D = my_y.shape
        H = self.H
        Hprime_start, gamma_start = self.Hprime, self.gamma
        if topK == -1:
            topK = self.state_matrix.shape[0]
        dev = self.device
        res_s = torch.zeros((my_N, topK, H), dtype=torch.int8, device=dev)
        res_m = torch.zeros((my_N, H), dtype=torch.float64, device=dev)
        res_am = torch.zeros((my_N, H), dtype=torch.float64, device=dev)
        res_p = torch.zeros((my_N, topK), dtype=torch.float64, device=dev)
        res_gamma = torch.zeros((my_N,), dtype=torch.float64, device=dev)
        res_Hprime = torch.zeros((my_N,), dtype=torch.float64, device=dev)

        def regenerate():
            (self.single_state_matrix, self.state_matrix, self.no_states,
             self.state_abs) = generate_state_matrix(self.Hprime, self.gamma, self.H, self.states)

        cur_y = my_y
        which = torch.ones(my_N, dtype=torch.bool, device=dev)
        try:
            while bool(which.any()):
                ind_n = torch.nonzero(which).flatten()
                logpj, cand = self.compute_lpj(anneal, model_params, {'y': cur_y})
                lp = logpj.tensor if isinstance(logpj, DeviceArray) else torch.as_tensor(np.asarray(logpj)).to(dev)
                cd = (cand.tensor if isinstance(cand, DeviceArray) else torch.as_tensor(np.asarray(cand)).to(dev)).long()
                n_cur, S = lp.shape
                Hp = self.Hprime
                k_eff = min(topK, S)
                # top-K states, signed / absolute marginals and the writes into s / m / am in position order: one HIP pass
                # (pm_infer_topk_signed_f64).  States that differ only in WHICH position of a repeated candidate is active tie
                # exactly, and the order NumPy's argsort()[::-1] (tsc_et.py:626, an introsort) gives exact ties is not a
                # function of (value, column): the kernel flags the rows with a tie among their topK + 1 best, those -- a
                # handful -- are ranked with NumPy itself, and the pass runs again with the ranking handed in.
                lp = lp.contiguous() if lp.stride(1) != 1 else lp
                cd32 = cd.to(torch.int32).contiguous()
                vals = torch.from_numpy(np.ascontiguousarray(self.state_matrix.astype(np.int8))).to(dev)
                top_idx = torch.empty((n_cur, k_eff), dtype=torch.int32, device=dev)
                top_lpc = torch.empty((n_cur, k_eff), dtype=torch.float64, device=dev)
                top_post = torch.empty((n_cur, k_eff), dtype=torch.float64, device=dev)
                tie = torch.zeros(n_cur, dtype=torch.int32, device=dev)
                s_blk = res_s[ind_n, :k_eff].contiguous()
                m_blk, am_blk = res_m[ind_n].contiguous(), res_am[ind_n].contiguous()

                def run(rank):
                    self._call("infer_topk", "pm_infer_topk_signed_f64", _ptr(lp), lp.stride(0), _ptr(cd32), _ptr(vals), n_cur,
                               H, Hp, S, k_eff, rank, _ptr(top_idx), _ptr(top_lpc), _ptr(top_post), _ptr(tie), _ptr(s_blk),
                               _ptr(m_blk), _ptr(am_blk) if abs_marginal else None, self._stream())
                run(1)
                if bool((top_idx < 0).any()):
                    raise _lib.HipError("inference: non-finite log-joints (NaN) in %d datapoint(s)"
                                        % int((top_idx < 0).any(dim=1).sum()))
                tied = torch.nonzero(tie).flatten()
                if tied.numel():
                    rows = lp[tied].cpu().numpy()
                    rel = rows - rows.max(axis=1, keepdims=True)
                    lpc = rel - np.log(np.exp(rel).sum(axis=1, keepdims=True))
                    order = np.argsort(lpc, axis=-1)[:, ::-1][:, :k_eff]
                    top_idx[tied] = torch.from_numpy(np.ascontiguousarray(order).astype(np.int32)).to(dev)
                    run(0)
                res_Hprime[ind_n] = float(self.Hprime)
                res_gamma[ind_n] = float(self.gamma)
                res_s[ind_n, :k_eff] = s_blk
                res_m[ind_n] = m_blk
                res_am[ind_n] = am_blk
                res_p[ind_n, :k_eff] = top_lpc if logprob else top_post
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
                regenerate()
        finally:
            comm.Barrier()
            self.Hprime, self.gamma = Hprime_start, gamma_start
            regenerate()
        with np.errstate(divide='ignore'):
            m_out = res_m.cpu().numpy()
            am_out = res_am.cpu().numpy()
            if logprob:
                m_out, am_out = np.log(m_out), np.log(am_out)
        return {'s': res_s.cpu().numpy(), 'm': m_out, 'am': am_out, 'p': res_p.cpu().numpy(),
                'gamma': res_gamma.cpu().numpy(), 'Hprime': res_Hprime.cpu().numpy()}

    # ------------------------------------------------------------------ plumbing
    def _tables(self):
        key = (self.Hprime, self.gamma, self.state_matrix.shape[0])
        if self._tab is None or self._tab[0] != key:
            if not _lib.load().pm_bsc_rows16_supported(2 * self.H, self.Hprime, 0):
                raise _lib.HipError("TSC_ET: 2 H = %d one-cause states exceed the selection kernel's range (<= 512)"
                                    % (2 * self.H))
            idx = (self.state_matrix.astype(np.int64) + 1).astype(np.uint8)       # -1, 0, +1 -> 0, 1, 2
            self._tab = (key, torch.from_numpy(np.ascontiguousarray(idx)).to(self.device))
        return self._tab[1]

    def _params(self, anneal, sigma):
        beta = 1. / anneal['T']
        pre1 = -1. / 2. / sigma / sigma
        P = _lib.DscParams(K=3, K0=1, pre1=float(pre1), ecoef=float(beta * pre1),
                           pscale=float(beta if anneal['anneal_prior'] else 1.0), flags=1 | 2)
        for k in range(3):
            P.values[k] = float(self.states[k])
        return P

    def _prior(self, pi):
        """log prior of every table row over the H' positions (tsc_et.py:327-337)."""
        pm = np.where(self.state_matrix != 0, pi / 2, 1 - pi)
        return np.log(pm).sum(axis=1)

    def _params_dev(self, W, res):
        """Device copy of W^T (H,D), the Gram matrix and the scores for the current W and data."""
        return self._scores_params(W, res)

    # ------------------------------------------------------------------ hot path
    @tracing.traced
    def select_Hprimes(self, model_params, data):
        """``data['candidates']`` (N, Hprime): latents of the Hprime best one-cause states, best last; a latent
        may repeat (tsc_et.py:142-213)."""
        res = self._resident(data['y'])
        N = res["Y"].shape[0]
        H, Hp = self.H, self.Hprime
        self._tables()
        par = self._params_dev(model_params['W'], res)
        cand = torch.empty((N, Hp), dtype=torch.int32, device=self.device)
        if N and _lib.load().pm_xsc_select_supported(H, Hp, 1):
            # the 2 H one-cause values formed and ranked in one pass over the scores; candidates come out as latents
            self._call("select", "pm_xsc_select_f64", _ptr(par["A"]), H, _ptr(par["G"]), None, N, H, Hp, _ptr(cand),
                       self._stream())
        elif N:
            R = self._buf("tsc_sel", (N, 2 * H))
            self._call("select_scores", "pm_tsc_select_scores_f64", _ptr(par["A"]), H, _ptr(par["G"]), N, H, _ptr(R),
                       2 * H, self._stream())
            # raw mode ranks R itself; the Gram / norm arguments only need to be valid memory
            gdummy = self._buf("tsc_gdummy", (2 * H, 2 * H))
            self._call("select", "pm_bsc_select_estep_f64", _ptr(R), 2 * H, _ptr(gdummy), _ptr(res["ynorm2"]), None, None,
                       None, None, None, 0, self.gamma, None, N, 2 * H, Hp, 1 | 8, _ptr(cand), None, 0, None,
                       self._stream())
            cand = torch.remainder(cand, H)            # state index -> latent index (tsc_et.py:210)
        data['candidates'] = DeviceArray(cand, np.int64)
        return data

    @tracing.traced
    def E_step(self, anneal, model_params, my_data):
        """Log-pseudo-joints ``{'logpj': (N, S)}``, one column per table row (tsc_et.py:277-356)."""
        res = self._resident(my_data['y'])
        N = res["Y"].shape[0]
        H, Hp, S = self.H, self.Hprime, self.state_matrix.shape[0]
        tab = self._tables()
        par = self._params_dev(model_params['W'], res)
        cand = self._device_candidates(my_data['candidates'], N)
        P = self._params(anneal, model_params['sigma'])
        prior = self._upload("tsc_prior", self._prior(model_params['pi']))
        tracing.tracepoint("E_step:iterating")
        return {'logpj': self._dsc_estep(anneal, "tsc_stats", par, res, cand, tab, S, prior, P, S, [model_params['pi']])}

    @tracing.traced
    def M_step(self, anneal, model_params, my_suff_stat, my_data):
        """New W, pi, sigma (tsc_et.py:359-542).  Logs ``L`` and ``N_use``."""
        comm = self.comm
        H, Hp, D, gamma = self.H, self.Hprime, self.D, self.gamma
        S = self.state_matrix.shape[0]
        pi, sigma = model_params['pi'], model_params['sigma']
        res = self._resident(my_data['y'])
        Y = res["Y"]
        my_N = Y.shape[0]
        tab = self._tables()
        cand = self._device_candidates(my_data['candidates'], my_N)

        logpj = my_suff_stat['logpj']
        if isinstance(logpj, DeviceArray) and getattr(logpj, "lse", None) is not None:
            lp, lse = logpj.tensor, logpj.lse
        else:
            lp = torch.from_numpy(np.ascontiguousarray(np.asarray(logpj), dtype=np.float64)).to(self.device)
            lse = torch.logsumexp(lp, dim=1)
        lp, lse = lp.contiguous(), lse.contiguous()
        assert tuple(lp.shape) == (my_N, S)
        N = self._global_count(res, my_N)

        # factors of the pi update (tsc_et.py:425-432)
        A_pi_gamma = 0.0
        B_pi_gamma = 0.0
        for gam1 in range(gamma + 1):
            for gam2 in range(gamma - gam1 + 1):
                cmb = comb(gam1, gam1) * comb(gam1 + gam2, gam2) * comb(H, H - gam1 - gam2)
                t = cmb * ((pi / 2) ** (gam1 + gam2)) * ((1 - pi) ** (H - gam1 - gam2))
                A_pi_gamma += t
                B_pi_gamma += (gam1 + gam2) * t
        E_pi_gamma = pi * H * A_pi_gamma / B_pi_gamma

        # data truncation (tsc_et.py:435-446): evidence >= the N_use-th largest
        lse_cut, cut_dev = float("-inf"), None
        if anneal['Ncut_factor'] > 0.0:
            tracing.tracepoint("M_step:truncating")
            N_use = int(N * (1 - (1 - A_pi_gamma) * anneal['Ncut_factor'])) or N    # (0: upstream's allsort(...)[-0] keeps everything)
            # the kernel keeps lse > cut; the un-stabilised sums upstream cuts on are exactly 0 below the
            # underflow boundary, where `>= 0` keeps every datapoint
            if lse.is_cuda and my_N:      # (the cut stays on the device: the row pass reads it there)
                c = self._kth_select_dev(lse, N_use)
                ninf = torch.full_like(c, float("-inf"))
                cut_dev = torch.where(c < _LOG_UNDERFLOW, ninf, torch.nextafter(c, ninf))
                lse_cut = float("nan")    # (not -inf: statistics a fused E-step pass may have left do not apply)
            else:
                cut = self._kth_largest_global(lse, N_use)
                lse_cut = float("-inf") if cut < _LOG_UNDERFLOW else float(np.nextafter(cut, -np.inf))

        tracing.tracepoint("M_step:iterating")
        lib = _lib.load()
        P = self._params(anneal, sigma)
        fused = self._dsc_fused_stats(logpj, res, cand, P, [pi], lse_cut) if my_N else None
        stats = fused["stats"] if fused else self._buf("tsc_stats", (lib.pm_dsc_stats_len(H, D),))
        if not fused:
            stats.zero_()
        expect = self._buf("expect", (my_N, H))
        prior = None if fused else self._upload("tsc_prior", self._prior(pi))
        if my_N:
            self._rows_and_wp((_ptr(lp), S, _ptr(lse), ctypes.c_double(lse_cut), _ptr(cand), _ptr(tab), S,
                               _ptr(prior) if prior is not None else None,
                               ctypes.byref(P), my_N, H, D, Hp, _ptr(expect), H, _ptr(stats)),
                              S, expect, Y, stats, my_N, int(P.K), int(P.flags), Hp, S, fused=fused, cut_dev=cut_dev)
        comm.allreduce_device(stats)      # replaces tsc_et.py:412,446,453,486,487,497,527
        self._mstep_res = res
        return self._finalize(stats, model_params, A_pi_gamma, E_pi_gamma)

    def _finalize(self, stats, model_params, A_pi_gamma, E_pi_gamma):
        """Parameter updates from the all-reduced statistics (tsc_et.py:448-542), one device->host copy."""
        H, D = self.H, self.D
        pi, sigma = model_params['pi'], model_params['sigma']
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
        my_sigma, Fs, N_use = float(host[8]), float(host[9]), int(round(host[10]))

        L = -0.5 * D * np.log(2 * np.pi * sigma ** 2) - np.log(A_pi_gamma) + Fs / N_use      # tsc_et.py:449-453
        dlog.append('L', L)

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
            else:   # singular Wq: the reference's pseudo-inverse (tsc_et.py:488)
                self._seed_rec = None
                self._winv_prev = None        # never warm-start the next inverse from a rejected one
                with small_blas():
                    W_new = np.dot(np.linalg.pinv(Wq.cpu().numpy()), Wp.cpu().numpy())
            W_out = W_new.transpose()
        else:
            W_out = np.asarray(model_params['W'])
        if 'pi' in self.to_learn:
            tracing.tracepoint("M_step:update pi")
            pi_new = E_pi_gamma * (cnt[0] + cnt[2]) / H / N_use          # expected number of non-zero latents
        else:
            pi_new = pi
        if 'sigma' in self.to_learn:
            tracing.tracepoint("M_step:update sigma")
            sigma_new = np.sqrt(my_sigma / D / N_use)
        else:
            sigma_new = sigma
        dlog.append('N_use', N_use)
        return {'W': W_out, 'pi': pi_new, 'sigma': sigma_new, 'Q': 0.}
