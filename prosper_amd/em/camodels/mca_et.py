"""Maximal Causes Analysis on the MI355X: drop-in for prosper/em/camodels/mca_et.py.

Same constructor, ``check_params`` (W >= 1e-4), ``select_Hprimes / E_step / M_step``
signatures, return keys (``W, pi, sigma, Q``) and ``dlog`` side effect (``N_use``) as the
reference's ``MCA_ET`` (mca_et.py:22-389).  Kernels (prosper_amd/csrc/mca_kernels.hip):

  select_Hprimes  sim[n,h] = sum_d |max(W_hd, y_d) - y_d| (register-tiled max-plus contraction),
                  the H' smallest per datapoint by the 16-lane DPP selection kernel
  E_step          singleton energies from the f64 MFMA scores GEMM, multi-cause states from
                  Wbar = (sum W^rho)^(1/rho) (one f64 log+exp per state and dimension)
  M_step          posterior q ~ exp(beta*logpj); Q1^T.Y as an f64 MFMA GEMM, the multi-cause
                  terms Aid scattered with f64 atomics, ONE all-reduce of the packed statistics
"""
import ctypes
from math import pi as _PI

import numpy as np
from scipy.special import comb

from ._device import DeviceCAModel, DeviceArray, _ptr
from ... import _lib
from ...utils import parallel
from ...utils import tracing
from ...utils.datalog import dlog

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class MCA_ET(DeviceCAModel):
    """Maximal Causes Analysis (max-superposition) with Expectation Truncation."""

    def __init__(self, D, H, Hprime, gamma, to_learn=['W', 'pi', 'sigma'], comm=parallel.COMM_WORLD,
                 device=None):
        DeviceCAModel.__init__(self, D, H, Hprime, gamma, to_learn, comm, device)
        self.rho_temp_bound = 1.05    # for rho: never use a T smaller than this (mca_et.py:31)
        self.W_tol = 1e-4             # for W: ensure W[W<W_tol] = W_tol       (mca_et.py:32)
        W_tol = self.W_tol
        self.noise_policy = {
            'W':     (W_tol, +np.inf, True),
            'pi':    (W_tol, 1 - W_tol, False),
            'sigma': (W_tol, +np.inf, False),
        }
        self._masks_dev = None
        self.signed_w = 0.0           # 1.0 in MMCA_ET: signed W, see pm_mca_params
        self.fuse_em = True           # E_step also produces the M-step's per-datapoint statistics when it can
        self.defer_stats = True       # ... on data-truncation steps as per-datapoint records, added once the cut is known
        self.defer_max_bytes = 32 << 30     # (N x H' x D doubles of records: beyond this the M-step runs its own pass)
        self.overlap_scores = True    # the next step's scores GEMM on a second stream beside the selection pass

    @tracing.traced
    def check_params(self, model_params):
        """Clamp W to >= W_tol (mca_et.py:44-55)."""
        model_params['W'] = np.maximum(model_params['W'], self.W_tol)
        return model_params

    @tracing.traced
    def _superpose(self, model_params, s, g):
        """y_d = max(0, max over the active causes of W_dh) (mca_et.py:76-80), in chunks of datapoints."""
        W = torch.from_numpy(np.ascontiguousarray(np.asarray(model_params['W'], dtype=np.float64))).to(self.device)   # (D,H)
        y = torch.zeros((s.shape[0], self.D), dtype=torch.float64, device=self.device)
        neg = torch.full((), float("-inf"), dtype=torch.float64, device=self.device)
        step = max(1, (1 << 25) // (self.D * self.H))
        for lo in range(0, s.shape[0], step):
            blk = s[lo:lo + step]
            y[lo:lo + step] = torch.where(blk[:, None, :], W[None, :, :], neg).amax(dim=2).clamp_min(0.0)
        return y

    def _generate_data_host(self, model_params, my_N):
        """Max-rule superposition + Gaussian noise; RNG stream as upstream (mca_et.py:58-85):
        one ``random(H)`` per datapoint, then one ``normal((my_N, D))``.  Does not obey gamma."""
        H, D = self.H, self.D
        W = model_params['W'].T
        y = np.zeros((my_N, D))
        s = np.zeros((my_N, H), dtype=bool)
        for n in range(my_N):
            s[n] = np.random.random(H) < model_params['pi']
            if s[n].any():
                y[n] = np.maximum(0.0, W[s[n]].max(axis=0))
        y += np.random.normal(scale=model_params['sigma'], size=(my_N, D))
        return {'y': y, 's': s}

    # ------------------------------------------------------------------ plumbing
    def _masks(self):
        if self._masks_dev is None or self._masks_dev[0] != (self.Hprime, self.gamma):
            lib = _lib.load()
            if not lib.pm_bsc_rows16_supported(self.H, self.Hprime, 0):
                raise _lib.HipError("MCA_ET: H = %d is outside the selection kernel's range (H <= 512)" % self.H)
            self._masks_dev = ((self.Hprime, self.gamma), self._u16_dev(self._state_masks()))
        return self._masks_dev[1]

    def _rho(self, T):
        T_rho = np.maximum(T, self.rho_temp_bound)
        return float(1. / (1. - 1. / T_rho))

    def _tables_on_device(self, Wt_dev, rho):
        """[W^T | sign(W)|W|^rho | |W|^(rho-1)] and |W_h|^2 from a device W^T (pm_mca_tables_f64), into the buffer set that
        is NOT the current step's (a speculative build for the next step must not touch tables in use)."""
        H, D = self.H, self.D
        self._tab_flip = 1 - getattr(self, "_tab_flip", 0)
        tabs = self._buf("mca_tabs%d" % self._tab_flip, (3, H, D))
        wnorm2 = self._buf("mca_wn%d" % self._tab_flip, (H,))
        self._call("tables", "pm_mca_tables_f64", _ptr(Wt_dev), H, D, ctypes.c_double(rho), _ptr(tabs), _ptr(wnorm2),
                   self._stream())
        return tabs, wnorm2

    def _tables_for(self, W_DH, T, res):
        """Device copies of the per-step tables: W (H,D), |W_h|^2, sign(W)|W|^rho, |W|^(rho-1) (mca_et.py:218-227,
        mmca_et.py:250-260 -- there NumPy on the host; here pm_mca_tables_f64, so that an EM loop does not wait for the
        host's log / exp over H x D elements: round 5).  The reference's assertions are kept, on W itself."""
        W = np.asarray(W_DH, dtype=np.float64)
        par = self._par
        if par.get("ykey") == res["key"] and par.get("T") == T and par.get("W") is not None \
                and par["W"].shape == W.shape and np.array_equal(par["W"], W):
            return par
        rho = self._rho(T)
        aW = np.abs(W) if self.signed_w else W
        lo, hi = float(aW.min()), float(aW.max())
        # finite logarithms; W^rho finite and > 1e-86 (mca_et.py:224-227, mmca_et.py:257-260: monotone in |W|)
        assert np.isfinite(hi) and lo > 0.0 and rho * np.log(lo) > np.log(1e-86) and rho * np.log(hi) < 709.0
        nxt, self._next_tabs = getattr(self, "_next_tabs", None), None
        if nxt is not None and nxt["ykey"] == res["key"] and nxt["rho"] == rho and nxt["W"] is not None \
                and nxt["W"].shape == W.shape and np.array_equal(nxt["W"], W):
            tabs, wnorm2 = nxt["tabs"], nxt["wnorm2"]      # the last M-step built them from its own result, on the device
            A_spec = nxt.get("A")                          # ... and the scores Y . W^T (good for one pass)
        else:
            tabs, wnorm2 = self._tables_on_device(self._upload("mca_Wt_in", np.ascontiguousarray(W.T)), rho)
            A_spec = None
        self._par = {"ykey": res["key"], "T": T, "W": W.copy(order='K'), "Wt": tabs[0], "Wrho": tabs[1], "Wrm1": tabs[2],
                     "wnorm2": wnorm2, "rho": rho, "A": A_spec}
        return self._par

    def _params(self, anneal, pies, sigma, rho):
        return _lib.McaParams(pil_bar=float(np.log(pies / (1. - pies))), pre1=float(-1. / 2. / sigma / sigma),
                              beta=float(1. / anneal['T']), inv_rho=float(1. / rho), signed_w=self.signed_w)

    # ------------------------------------------------------------------ hot path
    def step(self, anneal, model_params, my_data):
        """CAModel.step; the M-step builds the next step's power tables for the NEXT annealing point's rho."""
        self._next_anneal = self._predict_anneal(anneal)
        return DeviceCAModel.step(self, anneal, model_params, my_data)

    @tracing.traced
    def select_Hprimes(self, model_params, data):
        """``data['candidates']`` (N, Hprime): the latents with the smallest
        sum_d |max(W_hd, y_d) - y_d|, ascending (mca_et.py:88-111)."""
        res = self._resident(data['y'])
        Y = res["Y"]
        N, D = Y.shape
        H, Hp = self.H, self.Hprime
        self._masks()
        W = np.asarray(model_params['W'], dtype=np.float64)
        seed, self._sel_seed = getattr(self, "_sel_seed", None), None
        if seed is not None and seed["W"] is not None and seed["ykey"] == res["key"] and seed["W"].shape == W.shape \
                and np.array_equal(seed["W"], W):
            # the last M-step ranked the candidates for exactly this W behind its download (``_seed_select``)
            data['candidates'] = DeviceArray(seed["cand"], np.int64)
            return data
        Wt = self._upload("mca_W", np.ascontiguousarray(W.T))
        data['candidates'] = DeviceArray(self._select_on_device(res, Wt), np.int64)
        return data

    def _select_on_device(self, res, Wt):
        """The two selection launches for W^T (H,D) on the device; returns the candidates tensor."""
        Y = res["Y"]
        N, D = Y.shape
        H, Hp = self.H, self.Hprime
        cand = torch.empty((N, Hp), dtype=torch.int32, device=self.device)
        if N:
            R = self._buf("mca_sim", (N, H))
            self._call("select_scores", "pm_mca_select_scores_f64", _ptr(Y), D, _ptr(Wt), D, _ptr(R), H, N, H, D,
                       self._stream())
            # raw mode ranks R itself: the Gram / norm arguments only need to be valid memory
            gdummy = self._buf("mca_gdummy", (H, H))
            self._call("select", "pm_bsc_select_estep_f64", _ptr(R), H, _ptr(gdummy), _ptr(res["ynorm2"]), None, None,
                       None, None, None, 0, self.gamma, None, N, H, Hp, 1 | 4 | 8, _ptr(cand), None, 0, None,
                       self._stream())
        return cand

    def _seed_select(self, res, Wt_new):      # Wt_new: max(W_new^T, W_tol)
        """Next step's candidates from the W^T the M-step has just formed on the device, enqueued behind its download: the
        max-plus selection pass (0.45 ms at config 5) runs while the host unpacks the result and prepares the next step's
        power tables, instead of after both (the device idled ~0.2 ms per 8.4 ms iteration there).  ``select_Hprimes``
        adopts them iff it is called with the W this M-step returns AFTER ``check_params`` -- the clamp to W_tol that
        ``CAModel.step`` applies first (mca_et.py:44-55) is applied here too -- compared with a private snapshot."""
        self._sel_seed = {"ykey": res["key"], "cand": self._select_on_device(res, Wt_new), "W": None, "Wt": Wt_new}

    @tracing.traced
    def E_step(self, anneal, model_params, my_data):
        """Log-pseudo-joints ``{'logpj': (N, 1+H+S)}`` (mca_et.py:114-179; no beta here)."""
        res = self._resident(my_data['y'])
        Y = res["Y"]
        N, D = Y.shape
        H, Hp, S = self.H, self.Hprime, self.no_states
        T = anneal['T']
        par = self._tables_for(model_params['W'], T, res)
        masks = self._masks()
        cand = self._device_candidates(my_data['candidates'], N)
        P = self._params(anneal, model_params['pi'], model_params['sigma'], par["rho"])
        K = 1 + H + S
        logpj = torch.empty((N, K), dtype=torch.float64, device=self.device)
        lse1 = torch.empty((N,), dtype=torch.float64, device=self.device)
        lseb = torch.empty((N,), dtype=torch.float64, device=self.device)
        tracing.tracepoint("E_step:iterating")
        fused = None
        if N and self.deterministic:
            self._det_quanta(res, model_params, P, K)
        if N:
            A_spec, par["A"] = par.get("A"), None
            if A_spec is not None and tuple(A_spec[0].shape) == (N, H):
                torch.cuda.current_stream(self.device).wait_event(A_spec[1])
                A = A_spec[0]
            else:
                A = self._gemm_nt(Y, par["Wt"], self._buf("scores", (N, H)), "scores_gemm")
            hp_tile = 4 if Hp <= 4 else 8 if Hp <= 8 else 12
            dpl = 1 if D <= 64 else 2 if D <= 128 else 4 if D <= 256 else 8
            ncut = anneal['Ncut_factor'] > 0.0
            defer = (ncut and self.defer_stats and getattr(self, "_in_step", False) and H <= 512
                     and 8 * N * Hp * D <= self.defer_max_bytes)
            if self.fuse_em and (not ncut or defer) and D <= 512 and Hp <= 12 and dpl * hp_tile <= 48:
                # the M-step's per-datapoint statistics come out of the same pass (every multi-cause power is evaluated once
                # instead of twice): accumulated in the pass when no data truncation is ahead; with one ahead (49 of the 50
                # steps of the reference's schedules) left as per-datapoint records that M_step adds once the cut is known
                stats = torch.zeros(_lib.load().pm_mca_stats_len(H, D), dtype=torch.float64, device=self.device)
                q1 = torch.empty((N, H), dtype=torch.float64, device=self.device)
                rec = (self._buf("mca_defer_rec", (N, Hp, D)), self._buf("mca_defer_sc", (N, 4))) if defer else None
                self._call("estep_mstats", "pm_mca_estep_mstats_defer_f64", _ptr(A), H, _ptr(par["wnorm2"]),
                           _ptr(res["ynorm2"]), _ptr(Y), D, _ptr(par["Wrho"]), _ptr(par["Wrm1"]), _ptr(cand),
                           _ptr(masks), S, ctypes.byref(P), N, H, D, Hp, _ptr(logpj), K, _ptr(lse1), _ptr(lseb),
                           _ptr(q1), H, _ptr(stats), _ptr(rec[0]) if rec else None, _ptr(rec[1]) if rec else None,
                           self._stream())
                fused = {"stats": stats, "q1": q1, "par": par, "res": res, "cand": my_data['candidates'],
                         "pi": model_params['pi'], "sigma": model_params['sigma'], "defer": rec}
            else:
                self._call("estep", "pm_mca_estep_f64", _ptr(A), H, _ptr(par["wnorm2"]), _ptr(res["ynorm2"]), _ptr(Y),
                           D, _ptr(par["Wrho"]), _ptr(cand), _ptr(masks), S, ctypes.byref(P), N, H, D, Hp,
                           _ptr(logpj), K, _ptr(lse1), _ptr(lseb), self._stream())
        out = DeviceArray(logpj)
        out.fused = fused
        out.lse = lseb          # log sum exp(beta * logpj): weights and truncation
        out.lse1 = lse1         # log sum exp(logpj): the likelihood term Q
        out.T = T
        return {'logpj': out}

    @tracing.traced
    def M_step(self, anneal, model_params, my_suff_stat, my_data):
        """New W, pi, sigma and the ET log-likelihood Q (mca_et.py:182-377)."""
        comm = self.comm
        H, Hp, D, gamma, S = self.H, self.Hprime, self.D, self.gamma, self.no_states
        pies, sigma = model_params['pi'], model_params['sigma']
        res = self._resident(my_data['y'])
        Y = res["Y"]
        my_N = Y.shape[0]
        T = anneal['T']
        beta = 1. / T
        par = self._tables_for(model_params['W'], T, res)
        masks = self._masks()
        cand = self._device_candidates(my_data['candidates'], my_N)
        K = 1 + H + S

        logpj = my_suff_stat['logpj']
        if isinstance(logpj, DeviceArray) and getattr(logpj, "lse1", None) is not None and logpj.T == T:
            lp, lseb, lse1 = logpj.tensor, logpj.lse, logpj.lse1
        else:   # foreign log-joints (or another temperature): recompute the log-evidences
            lp = logpj.tensor if isinstance(logpj, DeviceArray) else \
                torch.from_numpy(np.ascontiguousarray(np.asarray(logpj), dtype=np.float64)).to(self.device)
            lse1 = torch.logsumexp(lp, dim=1)
            lseb = torch.logsumexp(beta * lp, dim=1)
        lp, lseb, lse1 = lp.contiguous(), lseb.contiguous(), lse1.contiguous()
        assert tuple(lp.shape) == (my_N, K)
        N = self._global_count(res, my_N)

        A_pi_gamma = 0.
        B_pi_gamma = 0.
        for gp in range(0, gamma + 1):
            a = comb(H, gp, exact=1) * pies ** gp * (1. - pies) ** (H - gp)
            A_pi_gamma += a
            B_pi_gamma += gp * a

        ncut = anneal['Ncut_factor'] > 0.0
        N_use = (int(N * (1 - (1 - A_pi_gamma) * anneal['Ncut_factor'])) or N) if ncut else N
        # (int(...) == 0 on a handful of datapoints: upstream's allsort(...)[-0] is the SMALLEST value -- everything is kept,
        # mca_et.py:255-258)
        lib = _lib.load()
        n_stats = lib.pm_mca_stats_len(H, D)
        fz = getattr(logpj, "fused", None) if isinstance(logpj, DeviceArray) else None
        mine = (fz is not None and fz["par"] is par and fz["res"] is res and fz["cand"] is my_data['candidates']
                and fz["pi"] == pies and fz["sigma"] == sigma and logpj.T == T)
        rec = fz.get("defer") if mine else None
        lse_cut = float("-inf")
        if ncut and rec is None:
            tracing.tracepoint("M_step:truncating")
            lse_cut = self._kth_largest_global(lseb, N_use)

        tracing.tracepoint("M_step:iterating")
        if mine and rec is not None:
            # the pass left every datapoint's statistics as a record (a data-truncation step): the cut is selected on the
            # device and stays there; pm_mca_defer_apply_f64 adds the records of the datapoints above it (round 6: no second
            # evaluation of the S x D powers -- pm_mca_mstep_rows_f64, 5 ms at config 5 -- and no host round trip)
            stats, q1 = fz["stats"], fz["q1"]
            logpj.fused = None
            if ncut:
                cut_dev = self._kth_select_dev(lseb, N_use)
            else:
                cut_dev = torch.full((1,), float("-inf"), dtype=torch.float64, device=self.device)
            if my_N and self.deterministic:
                self._det_quanta(res, model_params, self._params(anneal, pies, sigma, par["rho"]), K)
            if my_N:
                work = self._buf("mca_defer_work", (int(lib.pm_mca_defer_apply_work_len(H, D)),))
                self._call("defer_apply", "pm_mca_defer_apply_f64", _ptr(lseb), _ptr(cut_dev), _ptr(Y), D, _ptr(cand),
                           _ptr(rec[0]), _ptr(rec[1]), _ptr(q1), H, _ptr(stats), _ptr(work), my_N, H, D, Hp, self._stream())
        elif mine and lse_cut == float("-inf") and not ncut:
            # E_step already accumulated the per-datapoint statistics for exactly these inputs
            stats, q1 = fz["stats"], fz["q1"]
            logpj.fused = None            # consumed: the buffer is all-reduced in place below
        else:
            stats = self._buf("mca_stats", (n_stats,))
            stats.zero_()
            q1 = self._buf("mca_q1", (my_N, H))
            P = self._params(anneal, pies, sigma, par["rho"])
            if my_N and self.deterministic:
                self._det_quanta(res, model_params, P, K)
            if my_N:
                self._call("mstep_rows", "pm_mca_mstep_rows_f64", _ptr(lp), K, _ptr(lse1), _ptr(lseb),
                           ctypes.c_double(lse_cut), _ptr(Y), D, _ptr(par["Wrho"]), _ptr(par["Wrm1"]), _ptr(cand),
                           _ptr(masks), S, ctypes.byref(P), my_N, H, D, Hp, _ptr(q1), H, _ptr(stats), self._stream())
        if my_N:
            self._call("stats_gemm", "pm_gemm_tn_acc_f64", _ptr(q1), H, _ptr(Y), D, _ptr(stats), D, H, D, my_N,
                       self._stream())
        stats = stats[:3 * H * D + H + 4]   # the per-XCD scratch tail is already folded in (and zero)
        comm.allreduce_device(stats)      # replaces mca_et.py:208,253,340,341,357,366,371
        self._mstep_res = res
        return self._finalize(stats, model_params, par, A_pi_gamma, B_pi_gamma)

    def _det_quanta(self, res, model_params, P, K):
        """Deterministic mode: bounds of the statistics' partial sums -> quanta (pm_common.h, PM_Q).  Every M-step weight
        q_s (W_j / Wbar_s)^(rho-1) is <= 1 (Wbar_s >= each of its terms), so Wq <= N and |Wp| <= N max|y|; energies
        |Wbar_s - y|^2 <= (|y| + gamma max|W_h|)^2."""
        ymax, ynmax = self._det_data_bounds(res)
        W = np.maximum(np.abs(np.asarray(model_params['W'], dtype=np.float64)), self.W_tol)
        wn = float(np.sqrt((W * W).sum(axis=0)).max())
        emax = (ynmax + self.gamma * wn) ** 2
        lpmax = (abs(P.pil_bar) * self.gamma + abs(P.pre1) * emax) * max(1.0, abs(P.beta)) + np.log(max(K, 2))
        n = float(res["Y"].shape[0])
        self._det_set("mca", [n, n * ymax, n * self.gamma, n * emax, n * lpmax])
        self._det_set("gemm", [n * ymax, n * ymax])

    def _finalize(self, stats, model_params, par, A_pi_gamma, B_pi_gamma):
        """Element-wise W update and the scalars (mca_et.py:333-377), one device->host copy."""
        H, D = self.H, self.D
        pies, sigma = model_params['pi'], model_params['sigma']
        HD = H * D
        scal = stats[3 * HD + H:3 * HD + H + 4]          # behind [G1 | Wp_m | Wq_m] (H,D each) and q1sum (H)
        parts = [scal]
        learn_W = 'W' in self.to_learn
        if learn_W:
            tracing.tracepoint("M_step:update W")
            # one launch (pm_mca_w_update_f64) instead of eleven tensor operations; the clamped copy is what the next step's
            # check_params will make of W: the seeded selection ranks that
            Wt_new = torch.empty((H, D), dtype=torch.float64, device=stats.device)
            Wt_cl = torch.empty_like(Wt_new)
            if not stats.is_cuda:
                raise _lib.HipError("MCA_ET.M_step: the statistics must be a device tensor (no CPU path)")
            self._call("w_update", "pm_mca_w_update_f64", _ptr(stats), _ptr(par["Wt"]), H, D, ctypes.c_double(self.W_tol),
                       _ptr(Wt_new), _ptr(Wt_cl), self._stream())
            parts.append(Wt_new.reshape(-1))
        flat = torch.cat(parts)
        self._sel_seed = None
        res = getattr(self, "_mstep_res", None)
        seedable = (flat.is_cuda and learn_W and self.speculate and res is not None and res["Y"].shape[0] > 0
                    and type(self).select_Hprimes is MCA_ET.select_Hprimes)
        self._next_tabs = None
        if seedable:
            # behind the download: the next step's candidates AND its power tables from the clamped W^T on the device
            # (valid if the caller hands this W back at the same temperature: checked by value in _tables_for)
            # (the NEXT step's rho where the schedule's next point is known -- _predict_anneal --, else this step's)
            nxt = getattr(self, "_next_anneal", None)
            rho_next = self._rho(nxt['T']) if nxt is not None else par["rho"]

            def ahead():
                A_next = None
                if self.overlap_scores and self.timer is None:
                    # the next E-step's scores GEMM on a second stream BESIDE the max-plus selection pass (round 6; it used to
                    # follow it from E_step).  Measured: the two share more than their pipes -- side by side the selection takes
                    # 0.38 ms instead of 0.27 and the GEMM 0.35 instead of 0.13: 0.62 instead of 0.64 ms between two passes
                    side = getattr(self, "_side_stream", None)
                    if side is None:
                        side = self._side_stream = torch.cuda.Stream(device=self.device)
                    fork = torch.cuda.Event()
                    fork.record()
                    side.wait_event(fork)
                    with torch.cuda.stream(side):
                        Yr = res["Y"]
                        A_next = (self._gemm_nt(Yr, Wt_cl, self._buf("scores_spec", (Yr.shape[0], H)), "scores_gemm"),
                                  torch.cuda.Event())
                        A_next[1].record(side)
                self._seed_select(res, Wt_cl)
                tabs, wnorm2 = self._tables_on_device(Wt_cl, rho_next)
                self._next_tabs = {"ykey": res["key"], "rho": rho_next, "tabs": tabs, "wnorm2": wnorm2, "W": None, "A": A_next}
            host = self._download(flat, then=ahead)
        else:
            host = self._download(flat) if flat.is_cuda else flat.numpy()
        my_pi, my_sigma, ldenom_sum, N_use = float(host[0]), float(host[1]), float(host[2]), int(round(host[3]))
        dlog.append('N_use', N_use)

        if learn_W:
            # (D, H) as a transposed view of a contiguous (H, D) copy: no strided copy here, none when the next step takes
            # W^T again; the seed's snapshot is a second copy in the same memory order
            W_new = host[4:4 + HD].reshape(H, D).copy().T
            if self._sel_seed is not None:
                self._sel_seed["W"] = np.maximum(host[4:4 + HD].reshape(H, D), self.W_tol).T
                if self._next_tabs is not None:
                    self._next_tabs["W"] = self._sel_seed["W"]
        else:
            W_new = np.asarray(model_params['W'])
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

    def calculate_respons(self, anneal, model_params, data):
        """Posterior over the truncated states (mca_et.py:380-389)."""
        cand = np.sort(np.asarray(data['candidates']), axis=1)
        data['candidates'] = cand
        F = np.asarray(self.E_step(anneal, model_params, data)['logpj'])
        e = np.exp(F - F.max(axis=1)[:, None])
        return e / e.sum(axis=1).reshape(-1, 1)
