"""Gaussian (spike-and-slab) Sparse Coding on the MI355X: drop-in for
prosper/em/camodels/gsc_et.py (class ``GSC``), scalar observation noise.

Same constructor (incl. the silent gamma / Hprime reset, gsc_et.py:45-50), ``standard_init``,
``check_params``, ``generate_data``, ``select_Hprimes / E_step / M_step`` and parameter keys
(``W, pi, mu, psi_sq, sigma_sq``).  One HIP kernel per E-step (prosper_amd/csrc/gsc_kernels.hip)
computes the component scores, the candidates and every truncated state's posterior moments from
the f64 MFMA scores GEMM ``Y.W`` and ``G = W^T.W``; the M-step's contractions
(``Y^T.xpt_sz``, ``xpt_s^T.xpt_sz``, ``xpt_sz^T.xpt_sz``) are f64 MFMA GEMMs, its H x H algebra runs on
the host exactly as upstream (gsc_et.py:584-718).

Differences a caller can observe: datapoints keep their order (the reference returns
its statistics in candidate-bucket order and rewrites ``my_data['y']``, gsc_et.py:572-573; every
consumer only sums over datapoints), and the per-datapoint (N,H,H) moments ``xpt_ss`` / ``xpt_szsz``
are handed over as ``SummedMoments`` (their sum over datapoints), never materialised.
``sigma_sq_type`` 'diagonal' / 'full' run through the same kernel on Sigma^-1-weighted scores, Gram matrix
and norms (the kernel then sees sigma^2 = 1).
"""
import ctypes
import os

import numpy as np

from ._device import DeviceCAModel, DeviceArray, _ptr, small_blas
from . import CAModel
from ... import _lib
from ...utils import parallel
from ...utils import tracing

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class SummedMoments(object):
    """sum over datapoints of a per-datapoint (H,H) moment (what the M-step consumes of
    ``xpt_ss`` / ``xpt_szsz``).  ``.sum(axis=0)`` returns the (H,H) ndarray."""

    def __init__(self, total, N, shape=None):
        self._total = total                 # the (H,H) tensor, or a function that assembles it on first use
        self.shape = (N,) + tuple(shape if shape is not None else total.shape)

    def sum(self, axis=0):
        assert axis == 0, "only the sum over datapoints exists"
        if callable(self._total):
            self._total = self._total()
        return self._total


class LazyClusters(dict):
    """``my_data['data_clusters']`` as the reference's ``select_Hprimes`` leaves it (gsc_et.py:731-747): one entry per
    distinct sorted candidate set, keyed by ``str(candidates)``, holding ``'hprimes'`` (the set), ``'data'`` (its
    datapoints, rows of y) and ``'ind'`` (their indices), in order of first appearance.  The kernels never need the
    bucketing (they walk datapoints, not clusters), so it is built only when somebody looks: the first access runs the
    selection pass and groups the rows on the host."""

    def __init__(self, model, model_params, my_data):
        dict.__init__(self)
        self._build = (model, model_params, my_data)

    def _fill(self):
        if self._build is None:
            return
        (model, model_params, my_data), self._build = self._build, None
        y = my_data['y']
        y_host = np.asarray(y)
        cands = np.asarray(model.candidates(model_params, {'y': y})).astype(np.int64)
        for ind in range(cands.shape[0]):
            key = str(cands[ind, :])
            c = dict.get(self, key)
            if c is None:
                c = {'hprimes': cands[ind, :], 'data': [], 'ind': []}
                dict.__setitem__(self, key, c)
            c['data'].append(y_host[ind])
            c['ind'].append(ind)
        for c in dict.values(self):
            c['data'] = np.array(c['data']).reshape((len(c['ind']), y_host.shape[1]))

    def order(self):
        """Datapoint indices in the reference's cluster order (the row order of its E_step outputs)."""
        self._fill()
        return np.concatenate([np.asarray(c['ind'], dtype=np.int64) for c in dict.values(self)]) if len(self) else \
            np.zeros(0, dtype=np.int64)


def _lazy(name):
    def method(self, *a, **k):
        self._fill()
        return getattr(dict, name)(self, *a, **k)
    method.__name__ = name
    return method


for _name in ('__getitem__', '__iter__', '__len__', '__contains__', 'keys', 'values', 'items', 'get', '__repr__',
              '__eq__', 'copy'):
    setattr(LazyClusters, _name, _lazy(_name))


class GSC(DeviceCAModel):
    def __init__(self, D, H, Hprime=0, gamma=0, sigma_sq_type='scalar',
                 to_learn=['W', 'pi', 'mu', 'sigma_sq', 'psi_sq'], comm=parallel.COMM_WORLD, device=None):
        # CAModel asserts Hprime <= H, gamma <= Hprime on the raw arguments (camodels/__init__.py:90-91)
        DeviceCAModel.__init__(self, D, H, Hprime, gamma, to_learn, comm, device)
        tol = 1e-5
        self.noise_policy = {
            'W':        (-np.inf, +np.inf, False),
            'pi':       (tol, 1. - tol, False),
            'sigma_sq': (0., +np.inf, False),
            'mu':       (-np.inf, +np.inf, False),
            'psi_sq':   (0., +np.inf, False),
        }
        # gsc_et.py:45-50 -- note: the state table built by CAModel.__init__ is NOT regenerated upstream
        if gamma <= 0 or gamma > H:
            self.gamma = self.H
        if Hprime <= 0 or Hprime > H:
            self.Hprime = self.H
        elif Hprime < gamma:
            self.gamma = self.Hprime
        self.sigma_sq_type = sigma_sq_type
        self.dtype_precision = np.float64
        self._masks_dev = None
        self._seed = None        # next step's W^T / Gram / scores left on the device by M_step (_speculate)
        self.speculate = os.environ.get('PM_SPECULATE', '1') == '1'
        self.fuse_moment_gemm = True      # [Y | xs | xsz]^T xsz as one GEMM (a plain attribute: tests flip it)
        self.overlap_moments = True       # ... the two parts on two streams (an HBM stream beside an MFMA GEMM)
        self.list_pairs = False           # the H x H blocks of listed datapoints from their lists, the sparse product over Y only:
                                          # sparse 0.22 -> 0.13 ms + pairs kernel 0.07 -- the EM iteration does not move (rounds 4, 5)
        self.early_inverse = True         # one rank: the inverse chain on a stream of its own beside the contraction
        self.overlap_scores = True        # the next step's scores GEMM on a second stream beside Gram / finish kernel / download
        self.sparse_moments = True        # ... split into listed rows (sparse product) + gathered dense rows, when the
                                          # M-step itself launched the E-step (pm_gsc_estep_lists_f64)
        self._spec = None        # next step's whole E-step, launched by M_step from device-side parameters
        self.speculate_estep = os.environ.get('PM_SPECULATE_ESTEP', '1') == '1'
        self.spec_hits = 0
        self.inverse_fallbacks = 0    # EM steps whose device inverses were rejected (host LAPACK took over, speculation void)
        self._in_step = False
        self._next_anneal = None     # the next step's annealing point (_predict_anneal), or None
        self._flat_schedule = False
        # True: E_step returns its statistics and leaves my_data['y'] / ['candidates'] in the reference's cluster order
        # (gsc_et.py:572-573) instead of datapoint order -- for callers that walk my_data['data_clusters'] alongside them
        self.reference_order = False

    # ------------------------------------------------------------------ host-side mirror
    @tracing.traced
    def standard_init(self, my_data):
        """gsc_et.py:59-114 (same RNG draw order: W noise, pi, [mu], [psi_sq])."""
        comm = self.comm
        temp = CAModel.standard_init(self, my_data)
        model_params = {'W': temp['W'].copy()}
        pi = comm.bcast(np.random.rand(self.H)) * 0.95
        pi[pi < 0.05] = 0.05
        model_params['pi'] = pi
        my_y = np.asarray(my_data['y'])
        W_mean = parallel.allmean(my_y, axis=0, comm=comm)
        sigma_sq_sq = parallel.allmean((my_y - W_mean) ** 2, axis=0, comm=comm)
        if self.sigma_sq_type == 'full':
            model_params['sigma_sq'] = np.diag(np.diag(sigma_sq_sq)) + (0.001 * np.eye(self.D))
        elif self.sigma_sq_type == 'diagonal':
            model_params['sigma_sq'] = sigma_sq_sq + 0.001
        else:
            model_params['sigma_sq'] = np.mean(sigma_sq_sq) + 0.001
        if 'mu' in self.to_learn:
            mu = comm.bcast(np.random.normal(0, 1, [self.H]))
        else:
            mu = np.zeros(self.H)
        model_params['mu'] = mu
        if 'psi_sq' in self.to_learn:
            psi_sq_diag = comm.bcast(np.random.rand(self.H)) * 2
            psi_sq_diag[psi_sq_diag < 0.05] = 0.05
            psi_sq = np.diag(psi_sq_diag)
        else:
            psi_sq = np.eye(self.H)
        model_params['psi_sq'] = psi_sq
        return comm.bcast(model_params)

    @tracing.traced
    def resume_init(self, h5_result_file):
        """Model parameters from the last logged EM step of a ``result.h5`` (gsc_et.py:112-160; upstream's version
        refers to undefined names and cannot run -- this is what it sets out to do): W, pi, mu, psi_sq as stored,
        sigma_sq converted to this model's ``sigma_sq_type`` when the file holds another one."""
        from ...utils.datalog import resume_params
        last = resume_params(h5_result_file, ('W', 'pi', 'mu', 'psi_sq', 'sigma_sq'))
        sigma_sq, D = np.asarray(last['sigma_sq']), self.D
        if sigma_sq.ndim == 2:
            if self.sigma_sq_type == 'diagonal':
                sigma_sq = sigma_sq.diagonal()
            elif self.sigma_sq_type == 'scalar':
                sigma_sq = np.mean(sigma_sq.diagonal())
        elif sigma_sq.ndim == 1:
            if self.sigma_sq_type == 'full':
                sigma_sq = np.diag(sigma_sq)
            elif self.sigma_sq_type == 'scalar':
                sigma_sq = np.mean(sigma_sq)
        else:
            if self.sigma_sq_type == 'full':
                sigma_sq = sigma_sq * np.eye(D)
            elif self.sigma_sq_type == 'diagonal':
                sigma_sq = sigma_sq * np.ones(D)
        if np.ndim(sigma_sq) == 0:
            sigma_sq = float(sigma_sq)
        model_params = {'W': last['W'], 'pi': last['pi'], 'sigma_sq': sigma_sq, 'mu': last['mu'], 'psi_sq': last['psi_sq']}
        return self.comm.bcast(model_params)

    @tracing.traced
    def check_params(self, model_params):
        """gsc_et.py:164-184: finiteness / positivity asserts on rank 0."""
        if self.comm.rank == 0:
            for k in ('W', 'mu', 'pi', 'psi_sq'):
                assert np.isfinite(model_params[k]).all()
            assert np.isfinite(model_params['sigma_sq']).all()
            assert np.all(np.asarray(model_params['sigma_sq']) > 0) or self.sigma_sq_type == 'full'
        return model_params

    def generate_data_device(self, model_params, my_N, seed=None):
        """Spike-and-slab data on the device: s_h ~ Bernoulli(pi_h) (``<=`` as upstream), z = s * (mu + L n) with
        L L^T = psi_sq -- the marginal of N(mu, psi_sq) over the active latents is the Gaussian the reference draws for
        them (gsc_et.py:204-257) -- y = z.W^T + sd * n'.  Kept quirks: with a full noise covariance only its diagonal
        is used, and a datapoint whose active INDICES sum to 0 (none, or only latent 0) stays all zero."""
        g = self._gen(seed)
        dev, H, D = self.device, self.H, self.D
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)).to(dev)
        pi, mu, psi = t(model_params['pi']), t(model_params['mu']), t(model_params['psi_sq'])
        s = torch.rand((my_N, H), generator=g, device=dev, dtype=torch.float64) <= pi
        L = t(np.linalg.cholesky(np.asarray(model_params['psi_sq'], dtype=np.float64)))     # (H x H, on the host)
        zfull = mu[None, :] + torch.randn((my_N, H), generator=g, device=dev, dtype=torch.float64) @ L.t()
        live = (s.to(torch.float64) @ torch.arange(H, dtype=torch.float64, device=dev)) != 0
        z = torch.where(s & live[:, None], zfull, torch.zeros((), dtype=torch.float64, device=dev))
        sig = np.asarray(model_params['sigma_sq'], dtype=np.float64)
        sd = t(np.sqrt(np.diagonal(sig)) if sig.ndim == 2 else np.sqrt(sig) * np.ones(D))
        y = self._mix_linear(z, model_params['W'])
        y += live[:, None].to(torch.float64) * sd[None, :] * torch.randn((my_N, D), generator=g, device=dev, dtype=torch.float64)
        return {'y': DeviceArray(y), 's': DeviceArray(s), 'z': DeviceArray(z)}

    @tracing.traced
    def _generate_data_host(self, model_params, my_N):
        """gsc_et.py:186-202: one ``random(H)`` draw per datapoint, ``p <= pi``."""
        s = np.zeros((my_N, self.H), dtype=bool)
        for n in range(my_N):
            s[n] = np.random.random(self.H) <= model_params['pi']
        return self.generate_from_hidden(model_params, {'s': s})

    @tracing.traced
    def generate_from_hidden(self, model_params, my_hdata):
        """gsc_et.py:204-257 incl. its quirk of skipping a datapoint whose active INDICES sum to 0."""
        D, H = self.D, self.H
        s = my_hdata['s']
        my_N = s.shape[0]
        y = np.zeros((my_N, D))
        z = np.zeros((my_N, H))
        if self.sigma_sq_type == 'full':
            sd = np.sqrt(model_params['sigma_sq'].diagonal())
        elif self.sigma_sq_type == 'diagonal':
            sd = np.sqrt(model_params['sigma_sq'])
        else:
            sd = np.sqrt(model_params['sigma_sq']) * np.ones(D)
        for n in range(my_N):
            act = np.nonzero(s[n])[0]
            if np.sum(act) == 0:
                continue
            Ws = model_params['W'][:, act]
            z_n = np.random.multivariate_normal(model_params['mu'][act], (model_params['psi_sq'][act, :])[:, act],
                                                1).flatten()
            z[n, act] = z_n
            y[n] = Ws @ z_n + sd * np.random.randn(D)
        return {'y': y, 's': s, 'z': z}

    # ------------------------------------------------------------------ device tables
    def _require_scalar(self):
        if self.sigma_sq_type not in ('scalar', 'diagonal', 'full'):
            raise _lib.HipError("GSC: unknown sigma_sq_type %r" % (self.sigma_sq_type,))
        if not _lib.load().pm_gsc_supported(self.H, self.Hprime, self.gamma):
            raise _lib.HipError("GSC kernel range: H <= 512, gamma <= 8 (got H=%d Hprime=%d gamma=%d)"
                                % (self.H, self.Hprime, self.gamma))

    def _masks(self):
        key = (self.Hprime, self.gamma, self.no_states)
        if self._masks_dev is None or self._masks_dev[0] != key:
            SM = self.state_matrix.astype(np.int64)
            m = (SM << np.arange(SM.shape[1])[None, :]).sum(axis=1).astype(np.uint16) if SM.size else np.zeros(0, np.uint16)
            self._masks_dev = (key, self._u16_dev(m))
        return self._masks_dev[1]

    def _tables_for(self, model_params, res):
        """Per-step device tables.  A diagonal / full noise covariance Sigma turns every inner product into a
        Sigma^-1-weighted one (gsc_et.py:321-346, 414-425): the scores GEMM runs against Sigma^-1 W, the Gram
        matrix is W^T Sigma^-1 W, |y|^2 becomes y^T Sigma^-1 y, and the kernel sees sigma^2 = 1."""
        W = np.asarray(model_params['W'], dtype=np.float64)
        mu = np.asarray(model_params['mu'], dtype=np.float64)
        psi = np.asarray(model_params['psi_sq'], dtype=np.float64)
        pi = np.asarray(model_params['pi'], dtype=np.float64)
        sig = np.asarray(model_params['sigma_sq'], dtype=np.float64)
        par = self._par
        same = (par.get("ykey") == res["key"] and par.get("W") is not None and par["W"].shape == W.shape
                and np.array_equal(par["W"], W) and np.array_equal(par["mu"], mu) and np.array_equal(par["psi"], psi)
                and np.array_equal(par["pi"], pi) and par["sig"].shape == sig.shape and np.array_equal(par["sig"], sig))
        if same:
            return par
        Y = res["Y"]
        N, D = Y.shape
        seed, self._seed = self._seed, None
        scores = None
        if seed is not None and seed["ykey"] == res["key"] and sig.ndim == 0 and seed["W_host"] is not None \
                and seed["W_host"].shape == W.shape and np.array_equal(seed["W_host"], W):
            # the W the last M-step returned: its transpose, Gram matrix and scores are already on the device
            Wt, G, scores = seed["Wt"], seed["G"], seed["A"]
            s2, Wst, Gd, yn = float(sig), Wt, (W * W).sum(axis=0), res["ynorm2"]
        else:
            Wt = self._upload("gsc_W", W).t().contiguous()      # (H, D): rows = latents
        if scores is not None:
            pass
        elif sig.ndim == 0:
            s2, Wst, Gd, yn = float(sig), Wt, (W * W).sum(axis=0), res["ynorm2"]
            G = self._gemm_nt(Wt, Wt, self._buf("gram", (self.H, self.H)), "gram_gemm")
        else:
            s2 = 1.0
            if sig.ndim == 1:
                assert sig.shape == (D,)
                sinv = 1. / sig
                Ws = W * sinv[:, None]
                yn = torch.empty(N, dtype=torch.float64, device=self.device)
                if N:
                    self._call("row_wsqnorm", "pm_row_wsqnorm_f64", _ptr(Y), Y.stride(0), N, D,
                               _ptr(self._upload("gsc_sinv", sinv)), _ptr(yn), self._stream())
            else:
                assert sig.shape == (D, D)
                with small_blas():
                    Sinv = np.linalg.inv(sig)
                    Sinv = 0.5 * (Sinv + Sinv.T)
                    C = np.linalg.cholesky(Sinv)                # Sinv = C C^T  ->  y^T Sinv y = |C^T y|^2
                    Ws = Sinv @ W
                yn = torch.empty(N, dtype=torch.float64, device=self.device)
                if N:
                    Ct = self._upload("gsc_chol", np.ascontiguousarray(C.T))
                    Z = self._gemm_nt(Y, Ct, self._buf("gsc_white", (N, D)), "whiten_gemm")
                    self._call("row_sqnorm", "pm_row_sqnorm_f64", _ptr(Z), D, N, D, _ptr(yn), self._stream())
            Gd = (Ws * W).sum(axis=0)
            Wst = self._upload("gsc_Ws", Ws).t().contiguous()
            G = self._gemm_nt(Wst, Wt, self._buf("gram", (self.H, self.H)), "gram_gemm")
            G.copy_(0.5 * (G + G.t()))                           # symmetric up to rounding; the kernel reads both halves
        psid = np.diag(psi)
        lam = Gd / s2 + 1. / psid
        with np.errstate(divide='ignore', invalid='ignore'):
            c0 = -(np.log(psid) + np.log(lam)) - mu * mu * Gd / s2
            lpi = np.log(pi) - np.log(1 - pi)
        tables = np.stack([c0, 2. * mu / s2, Gd * mu, 1. / (lam * s2 * s2), 1. / (lam * s2), 1. / lam, mu, lpi])
        self._par = {"ykey": res["key"], "W": W.copy(), "mu": mu.copy(), "psi": psi.copy(), "pi": pi.copy(),
                     "sig": sig.copy(), "s2": s2, "Wt": Wt, "Wst": Wst, "G": G, "yn": yn,
                     "psi_d": self._upload("gsc_psi", psi), "tables": self._upload("gsc_tab", tables),
                     "scores": scores}
        return self._par

    def _speculate(self, res, Wt):
        """Next step's Gram matrix and scores from the W^T the M-step has just solved on the device, enqueued
        behind the M-step's download: they run while the host unpacks the result and prepares the next tables.
        ``_tables_for`` picks them up if the caller hands the returned W back unchanged (plain EM)."""
        Y = res["Y"]
        N, H = Y.shape[0], self.H
        G = self._gemm_nt(Wt, Wt, torch.empty((H, H), dtype=torch.float64, device=self.device), "gram_gemm")
        A = self._gemm_nt(Y, Wt, self._buf("scores_spec", (N, H)), "scores_gemm") if N else None
        self._seed = {"ykey": res["key"], "Wt": Wt, "G": G, "A": A, "W_host": None}

    # ------------------------------------------------------------------ hot path
    def _run(self, anneal_T, model_params, res, cand_in, logpj=None):
        """Scores GEMM + the fused select / E-step kernel; returns (cand, xpt_s, xpt_sz, stats)."""
        self._require_scalar()
        Y = res["Y"]
        N, D = Y.shape
        H, Hp, S = self.H, self.Hprime, self.no_states
        par = self._tables_for(model_params, res)
        if self.deterministic and N:
            self._det_quanta(res, model_params)
        A = None
        if N:
            A = par.pop("scores", None)          # left by the previous M-step (_speculate); good for one pass
            par["scores"] = None
            if A is None:
                A = self._gemm_nt(Y, par["Wst"], self._buf("scores", (N, H)), "scores_gemm")
        return self._launch_estep(res, A, par["G"], par["psi_d"], par["yn"], par["tables"], par["s2"], anneal_T, cand_in,
                                  logpj)

    def _launch_estep(self, res, A, G, psi_d, yn, tables, s2, anneal_T, cand_in, logpj=None, lists=False, zeros=None):
        """The fused select / E-step kernel on scores ``A``; ``s2 == 0.0``: 1/sigma_sq sits in the ninth row of
        ``tables`` (an M-step that finished on the device).  Returns (cand, xpt_s, xpt_sz, stats)."""
        N = res["Y"].shape[0]
        H, Hp, S = self.H, self.Hprime, self.no_states
        masks = self._masks()
        n_stats = _lib.load().pm_gsc_stats_len(H)
        # (`zeros`: a zeroed statistics buffer and dense-row counter the caller filled earlier, off the critical path)
        stats = zeros[0] if zeros is not None else torch.zeros(n_stats, dtype=torch.float64, device=self.device)
        # xpt_s and xpt_sz side by side, and BEHIND A COPY OF Y, in one (N, D + 2H) buffer: the M-step then gets all three
        # contractions over the datapoints -- [Y | xs | xsz]^T . xsz = [Wp ; xs^T xsz ; xsz^T xsz] -- from a single GEMM
        # launch (0.43 instead of 0.56 ms at config 4: one (D + 2H) x H output keeps the chip fuller than two D x H
        # ones).  Two such buffers alternate, so the moments handed out by one E-step survive the next one (the M-step
        # launches it early).
        D = res["Y"].shape[1]
        if N and self.fuse_moment_gemm:
            bufs = res.setdefault("gsc_big", [None, None])
            k = res["gsc_flip"] = 1 - res.get("gsc_flip", 1)
            if bufs[k] is None:      # (+ a row of zeros: what the gathered GEMM reads past the end of its row list)
                bufs[k] = torch.empty((N + 1, D + 2 * H), dtype=torch.float64, device=self.device)
                bufs[k][:N, :D] = res["Y"]
                bufs[k][N].zero_()
            both = bufs[k][:N, D:]
        else:
            both = torch.empty((N, 2 * H), dtype=torch.float64, device=self.device)
        xs, xsz = both[:, :H], both[:, H:]
        if cand_in is None:
            cand = torch.empty((N, Hp), dtype=torch.int32, device=self.device)
            do_select = 1
        else:
            cand, do_select = cand_in, 0
        if N and logpj is not None:      # compute_lpj: the same pass also writes every state's log-joint
            self._call("estep", "pm_gsc_estep_lpj_f64", _ptr(A), H, _ptr(G), _ptr(psi_d), _ptr(yn),
                       _ptr(tables), _ptr(masks), S, self.gamma, ctypes.c_double(1. / anneal_T),
                       ctypes.c_double(s2), N, H, Hp, do_select, _ptr(cand), _ptr(xs), _ptr(xsz), both.stride(0),
                       _ptr(stats), _ptr(logpj), logpj.stride(0), self._stream())
        elif N and lists and self.fuse_moment_gemm and self.sparse_moments and cand_in is None \
                and _lib.load().pm_gsc_lists_supported(H, Hp, self.gamma, D):
            # (`tables` comes from pm_gsc_mstep_finish_f64 here: its slot 8 H + 1 holds the list threshold)
            lb = res.setdefault("gsc_lists", [None, None])
            if lb[k] is None:
                lb[k] = (torch.empty((N, 16), dtype=torch.int16, device=self.device),
                         torch.empty((2, N, 16), dtype=torch.float64, device=self.device),   # xpt_sz | xpt_s at the listed entries
                         torch.empty(N, dtype=torch.int32, device=self.device))
            nz_idx, nz_val, dense_rows = lb[k]
            dense_count = zeros[1] if zeros is not None else torch.zeros(1, dtype=torch.int32, device=self.device)
            self._call("estep", "pm_gsc_estep_lists_f64", _ptr(A), H, _ptr(G), _ptr(psi_d), _ptr(yn),
                       _ptr(tables), _ptr(masks), S, self.gamma, ctypes.c_double(1. / anneal_T),
                       ctypes.c_double(s2), N, H, Hp, do_select, _ptr(cand), _ptr(xs), _ptr(xsz), both.stride(0),
                       _ptr(stats), _ptr(nz_idx), _ptr(nz_val), _ptr(dense_rows), _ptr(dense_count), self._stream())
            if self.deterministic:
                # the dense rows were listed in the order the workgroups finished: ascending, so that the gathered GEMM's
                # K-slices hold the same rows in the same order in every run.  (Right behind the pass, on an idle device: beside
                # the sparse product the one-workgroup kernel waited 0.2 ms for room on a CU, and the GEMM for it.)
                flags = res.get("gsc_row_flags")
                if flags is None:      # (zero between calls: the compaction clears what it reads)
                    flags = res["gsc_row_flags"] = torch.zeros((N + 7) // 8 * 8 + 4 * ((N + 8191) // 8192), dtype=torch.uint8,
                                                               device=self.device)
                self._call("estep", "pm_sort_row_list_i32", _ptr(dense_rows), _ptr(dense_count), N, _ptr(flags), self._stream())
            stats._pm_lists = (nz_idx, nz_val, dense_rows, dense_count, bufs[k])
        elif N:
            self._call("estep", "pm_gsc_estep_f64", _ptr(A), H, _ptr(G), _ptr(psi_d), _ptr(yn),
                       _ptr(tables), _ptr(masks), S, self.gamma, ctypes.c_double(1. / anneal_T),
                       ctypes.c_double(s2), N, H, Hp, do_select, _ptr(cand), _ptr(xs), _ptr(xsz), both.stride(0),
                       _ptr(stats), self._stream())
        return cand, xs, xsz, stats

    def step(self, anneal, model_params, my_data):
        """CAModel.step; M_step knows that an E_step of the next EM step is likely to follow."""
        self._in_step = True
        self._next_anneal = self._predict_anneal(anneal)     # (round 6: across a temperature ramp too, not only a flat schedule)
        try:
            return DeviceCAModel.step(self, anneal, model_params, my_data)
        finally:
            self._in_step = False

    _PARAM_KEYS = ('W', 'pi', 'mu', 'psi_sq', 'sigma_sq')

    def _spec_matches(self, sp, anneal, model_params, res):
        if sp["res"] is not res or sp["T"] != anneal['T']:
            return False
        for k in self._PARAM_KEYS:
            a, b = np.asarray(model_params[k]), sp["params"][k]
            if a.shape != b.shape or not np.array_equal(a, b):
                return False
        return True

    @tracing.traced
    def select_Hprimes(self, model_params, my_data):
        """Candidates = the Hprime latents with the best singleton log-posterior, sorted by index
        (gsc_et.py:721-728).  The selection itself is fused into the E-step kernel; ``data_clusters`` -- the reference's
        bucketing of the datapoints by candidate set (gsc_et.py:731-747) -- is a ``LazyClusters`` that is filled when
        somebody looks at it.  Statistics stay in datapoint order unless ``self.reference_order`` is set."""
        self._require_scalar()
        res = self._resident(my_data['y'])
        my_data['data_clusters'] = LazyClusters(self, dict(model_params), my_data)     # (shallow: nothing is copied)
        my_data.pop('candidates', None)
        return my_data

    @tracing.traced
    def component_scores(self, model_params, my_data):
        """Singleton log-posterior of every latent without the prior, (N, H), clamped as upstream clamps it
        (gsc_et.py:752-809): what ``select_Hprimes`` ranks.  Scores GEMM + one element-wise HIP kernel."""
        self._require_scalar()
        res = self._resident(my_data['y'])
        N, H = res["Y"].shape[0], self.H
        par = self._tables_for(model_params, res)
        out = torch.empty((N, H), dtype=torch.float64, device=self.device)
        if N:
            A = self._gemm_nt(res["Y"], par["Wst"], self._buf("scores", (N, H)), "scores_gemm")
            self._call("component_scores", "pm_gsc_component_scores_f64", _ptr(A), H, _ptr(par["yn"]), _ptr(par["tables"]),
                       ctypes.c_double(par["s2"]), N, H, _ptr(out), H, self._stream())
        return DeviceArray(out)

    def compute_lpj(self, anneal, model_params, my_data):
        """Candidates and log-pseudo-joints of the truncated states (gsc_et.py:811-944): ``(logpj (N, 1+H+S),
        candidates (N, Hprime))`` in datapoint order -- no annealing, prior odds included; columns [null ; singletons ;
        multi-cause states in ``state_matrix`` order over the sorted candidates].  ``CAModel.inference`` consumes it."""
        assert 'y' in my_data, "Key 'y' in test_data dict not defined."
        self._require_scalar()
        res = self._resident(my_data['y'])
        N = res["Y"].shape[0]
        K = 1 + self.H + self.no_states
        logpj = torch.empty((N, K), dtype=torch.float64, device=self.device)
        cand, _, _, _ = self._run(1.0, model_params, res, None, logpj=logpj)
        return DeviceArray(logpj), DeviceArray(cand, np.int64)

    @tracing.traced
    def compute_posterior_hprime(self, anneal, model_params, my_data):
        """gsc_et.py:260-398: for ONE data cluster -- ``my_data['y']`` (n, D), all rows sharing the candidate set
        ``my_data['candidates']`` (Hprime,) -- the un-normalised sums over the multi-cause states of the truncated space:
        ``{'pstr_s' (n,H), 'pstr_ss' (n,H,H), 'pstr_sz' (n,H), 'pstr_szsz' (n,H,H), 'post_nfac_n' (n,)}`` as NumPy arrays
        (the reference's E_step accumulates them per cluster, :550, then adds the null and one-cause states and normalises).
        The training path never calls it: ``E_step`` produces the normalised moments of every datapoint in one kernel pass
        and only ever forms the (H,H) SUMS of the second moments -- this method materialises them per datapoint, so it is
        for cluster-sized inputs.  The state weights exp(beta lp) come from the E-step kernel's log-joint output
        (pm_gsc_estep_lpj_f64, candidates handed in); kappa and Lambda^-1 of a state depend on the cluster only through
        its candidates and are small dense algebra on the device."""
        self._require_scalar()
        y = np.asarray(my_data['y'], dtype=np.float64)
        if y.ndim == 1:
            y = y[None, :] if y.shape[0] == self.D else y[:, None]
        n, D, H, Hp, S = y.shape[0], self.D, self.H, self.Hprime, self.no_states
        comps = np.asarray(my_data['candidates']).astype(np.int64).reshape(-1)
        assert comps.shape == (Hp,) and y.shape[1] == D
        beta = 1. / anneal['T']
        tiny = np.finfo(np.float64).tiny
        dev = self.device
        out = {'pstr_s': np.zeros((n, H)), 'pstr_ss': np.zeros((n, H, H)), 'pstr_sz': np.zeros((n, H)),
               'pstr_szsz': np.zeros((n, H, H)), 'post_nfac_n': np.zeros(n)}
        if n == 0 or S == 0:
            return out
        # One pass of the E-step kernel with the candidates handed in, SORTED (its enumeration): besides the log-joints it leaves
        # every datapoint's sums over the multi-cause states as they stand in its LDS blocks (pm_gsc_estep_lpj_blocks_f64) --
        # the state weights exp(beta lp) with the reference's clamp, kappa and Lambda^-1 are the kernel's own.  The blocks are
        # indexed by candidate POSITION; scattering them by latent index makes the caller's order of the candidates immaterial.
        sorted_c = np.sort(comps, kind="stable")
        res = self._resident(y)
        par = self._tables_for(model_params, res)
        cand_in = self._device_candidates(np.tile(sorted_c[None, :], (n, 1)), n)
        A = self._gemm_nt(res["Y"], par["Wst"], self._buf("scores", (n, H)), "scores_gemm")
        ldb = 2 * Hp * Hp + 2 * Hp + 1
        blocks = torch.empty((n, ldb), dtype=torch.float64, device=dev)
        logpj = torch.empty((n, 1 + H + S), dtype=torch.float64, device=dev)
        both = torch.empty((n, 2 * H), dtype=torch.float64, device=dev)
        stats = torch.zeros(int(_lib.load().pm_gsc_stats_len(H)), dtype=torch.float64, device=dev)
        self._call("estep", "pm_gsc_estep_lpj_blocks_f64", _ptr(A), H, _ptr(par["G"]), _ptr(par["psi_d"]), _ptr(par["yn"]),
                   _ptr(par["tables"]), _ptr(self._masks()), S, self.gamma, ctypes.c_double(beta), ctypes.c_double(par["s2"]),
                   n, H, Hp, 0, _ptr(cand_in), _ptr(both), ctypes.c_void_p(both.data_ptr() + 8 * H), 2 * H, _ptr(stats),
                   _ptr(logpj), logpj.stride(0), _ptr(blocks), ldb, self._stream())
        blk = blocks.cpu().numpy()
        HH = Hp * Hp
        ii, kk = np.meshgrid(sorted_c, sorted_c, indexing="ij")
        out['pstr_ss'][:, ii, kk] = blk[:, :HH].reshape(n, Hp, Hp)
        out['pstr_szsz'][:, ii, kk] = blk[:, HH:2 * HH].reshape(n, Hp, Hp)
        out['pstr_s'][:, sorted_c] = blk[:, 2 * HH:2 * HH + Hp]
        out['pstr_sz'][:, sorted_c] = blk[:, 2 * HH + Hp:2 * HH + 2 * Hp]
        out['post_nfac_n'][:] = blk[:, 2 * HH + 2 * Hp]
        return out

    def candidates(self, model_params, my_data):
        """Sorted candidates (N, Hprime) on their own (np.asarray-able)."""
        res = self._resident(my_data['y'])
        cand, _, _, _ = self._run(1.0, model_params, res, None)
        return DeviceArray(cand, np.int64)

    @tracing.traced
    def E_step(self, anneal, model_params, my_data):
        """Posterior moments over the truncated state set (gsc_et.py:401-580): ``xpt_s``, ``xpt_sz``
        (N,H) device handles, ``xpt_ss`` / ``xpt_szsz`` as sums over datapoints."""
        res = self._resident(my_data['y'])
        N = res["Y"].shape[0]
        cand_in = None
        if 'candidates' in my_data and 'data_clusters' not in my_data:   # candidates handed in (sorted, as upstream)
            cand_in = self._device_candidates(np.sort(np.asarray(my_data['candidates']).astype(np.int64), axis=1), N)
        tracing.tracepoint("E_step:iterating")
        sp, self._spec = self._spec, None
        if sp is not None and cand_in is None and self._spec_matches(sp, anneal, model_params, res):
            cand, xs, xsz, stats = sp["out"]       # the previous M-step has already launched exactly this pass
            self.spec_hits += 1
        else:
            cand, xs, xsz, stats = self._run(anneal['T'], model_params, res, cand_in)
        my_data['candidates'] = DeviceArray(cand, np.float64)             # float array upstream (gsc_et.py:431)
        H = self.H
        st = stats
        U_ss = st[:H * H].view(H, H)
        U_zz = st[H * H:2 * H * H].view(H, H)
        cs, csz, dzz = (st[2 * H * H + i * H:2 * H * H + (i + 1) * H] for i in range(3))   # per-XCD scratch follows
        def sum_ss():        # (assembled only for a caller that asks: M_step packs both straight from `stats`)
            off = torch.triu(U_ss, 1)
            return off + off.t() + torch.diag(cs)                           # diag(sum xpt_ss) = sum xpt_s
        # xpt_szsz = kappa kappa^T + Lambda^-1 is symmetric only while psi_sq is (gsc_et.py:660-675 returns a non-symmetric
        # one): the kernel accumulates both triangles as they are
        sum_zz = lambda: U_zz + torch.diag(dzz)
        if self.reference_order and isinstance(my_data.get('data_clusters'), LazyClusters):
            # the reference's row order (gsc_et.py:572-573): clusters in order of first appearance
            order = my_data['data_clusters'].order()
            idx = torch.from_numpy(order).to(self.device)
            xs, xsz = xs.index_select(0, idx), xsz.index_select(0, idx)
            my_data['candidates'] = DeviceArray(cand.index_select(0, idx), np.float64)
            y = my_data['y']
            my_data['y'] = DeviceArray(y.tensor.index_select(0, idx)) if isinstance(y, DeviceArray) else np.asarray(y)[order]
        out = {'xpt_s': DeviceArray(xs), 'xpt_sz': DeviceArray(xsz),
               'xpt_ss': SummedMoments(sum_ss, N, (H, H)), 'xpt_szsz': SummedMoments(sum_zz, N, (H, H))}
        out['_sums'] = (cs, csz)
        out['_stats'] = (stats, out['xpt_ss'], out['xpt_szsz'])      # M_step: pack from the kernel's buffer in one launch
        return out

    @tracing.traced
    def M_step(self, anneal, model_params, suff_stats, my_data):
        """gsc_et.py:584-718 for scalar sigma_sq: the three contractions over datapoints are f64 MFMA
        GEMMs into one packed buffer (ONE all-reduce), the H x H algebra follows upstream on the host."""
        comm = self.comm
        H, D = self.H, self.D
        res = self._resident(my_data['y'])
        Y = res["Y"]
        my_N = Y.shape[0]
        N = self._global_count(res, my_N)
        eps = 1e-5

        def dev(x):
            if isinstance(x, DeviceArray):
                return x.tensor
            if torch.is_tensor(x):
                return x.to(self.device)
            return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float64)).to(self.device)

        old_dev = None
        if self.deterministic and my_N:
            # (the quanta of a pass the previous M-step launched were derived on the device, for these very parameters:
            # _det_dev_quanta; they are installed again only if another model's launches came in between)
            made = getattr(suff_stats.get('_stats', (None,))[0], "_pm_det", None)
            if made is not None:
                self._det_dev_install(made)
            else:
                self._det_quanta(res, model_params)
        if self.speculate and self.speculate_estep and self.sigma_sq_type == 'scalar' and my_N and 'W' in self.to_learn:
            # the old parameters for pm_gsc_mstep_finish_f64, uploaded NOW: enqueued in front of the contraction they are on
            # the device long before the finish kernel wants them (uploaded next to it, the copy and its latency sat in the
            # middle of the M-step's chain of small launches: ~30 us of idle device per step)
            # (inside an EM loop they ARE on the device: the previous M-step's own result, if the caller handed it back
            # unchanged -- an upload here, which the contraction's stream waits for, left the device idle for ~40 us per step)
            kept, self._dev_params = getattr(self, "_dev_params", None), None
            if kept is not None and all(np.array_equal(np.asarray(model_params[k]), kept[1][k])
                                        for k in ('pi', 'mu', 'psi_sq', 'sigma_sq')):
                old_dev = kept[0]
            else:
                old_dev = self._upload("gsc_old", np.concatenate([np.asarray(model_params[k], dtype=np.float64).reshape(-1)
                                                                   for k in ('pi', 'mu', 'psi_sq', 'sigma_sq')]))
        xs, xsz = dev(suff_stats['xpt_s']), dev(suff_stats['xpt_sz'])
        ld = xs.stride(0) if xs.dim() == 2 else 0
        paired = (xs.dim() == 2 and xs.stride(1) == 1 and xsz.stride() == (ld, 1) and H % 2 == 0 and ld % 2 == 0
                  and xsz.data_ptr() == xs.data_ptr() + 8 * H)       # the E-step's own [xs | xsz] buffer
        big = None                                                   # ... which sits behind a copy of Y
        if paired and ld == D + 2 * H:
            for b in res.get("gsc_big", ()):
                if b is not None and b.shape[0] == my_N + 1 and xs.data_ptr() == b.data_ptr() + 8 * D:
                    big = b
        if not paired:
            xs, xsz = xs.contiguous(), xsz.contiguous()
        ldx = xs.stride(0) if my_N else H
        raw = suff_stats.get('_stats')
        if not (raw is not None and raw[1] is suff_stats['xpt_ss'] and raw[2] is suff_stats['xpt_szsz'] and my_N
                and suff_stats.get('_sums') is not None):
            raw = None                      # moments from outside (or replaced by the caller): the general path
            sum_ss = dev(suff_stats['xpt_ss'].sum(axis=0))
            sum_zz = dev(suff_stats['xpt_szsz'].sum(axis=0))
        # packed: [Wp (D,H) | xs^T xsz (H,H) | xsz^T xsz (H,H) | sum_ss | sum_zz | sum_s | sum_sz | sum |y|^2]
        nWp, nHH = D * H, H * H
        n_stat = nWp + 4 * nHH + 2 * H + 1
        # one buffer, one download: [statistics (all-reduced) | 2 inverses + 4 pivots | W_new^T | pi mu psi_sq sigma_sq]
        n_inv, n_par = 2 * nHH + 4 + 2, 2 * H + nHH + 1          # (+ 2: the warm starts' accepted flags)
        o_inv = n_stat + (n_stat & 1)               # (16-byte alignment for the GEMM's vector loads)
        n_whole = o_inv + n_inv + nWp + n_par
        whole, self._whole_next = getattr(self, "_whole_next", None), None        # (zeroed by the previous M-step, off the
        if whole is None or whole.numel() != n_whole or whole.device != torch.device(self.device):  # critical path)
            whole = torch.zeros(n_whole, dtype=torch.float64, device=self.device)
        packed = whole[:n_stat]
        o = nWp + 2 * nHH
        o2 = o + 2 * nHH
        o_wt, o_par = o_inv + n_inv, o_inv + n_inv + nWp
        at = lambda off: ctypes.c_void_p(whole.data_ptr() + 8 * off)
        yy = res.get("ynorm2_sum")                   # sum_n |y_n|^2: a constant of the resident shard
        if yy is None:
            yy = res["ynorm2_sum"] = res["ynorm2"].sum().reshape(1)

        def invert(st):
            """The two H x H inverses of the update (gsc_et.py:625, 673) on the device, ahead of the download: a 128 x 128 LAPACK
            inverse costs 0.4 ms of host time each while the GPU idles.  [sum_ss ; sum_zz] sit back to back in the statistics;
            inverses are stored in that order: [(sum_ss + eps I)^-1 ; sum_zz^-1 ; 4 pivots].  sum_zz is a GENERAL matrix from the
            second EM step on (psi_sq is not symmetric any more: gsc_et.py:660-675) and the reference inverts it as it is (:625):
            the left-sided Newton-Schulz refinement of pm_inverse_warm_batch_f64 (its result is the inverse of the transpose --
            what W_new^T = (A^-1)^T Wp^T needs), started from the previous EM step's inverses or, cold, from the sweep's inverses
            of the upper-mirrored matrices (within ~1e-6).  (PM_WARM_INVERSE=0 only forces the cold START; the Newton-Schulz pass
            itself is not optional here -- it is what inverts the general matrix.)"""
            dadd = self._eps_diag(H, eps)
            prev = getattr(self, "_inv_prev", None)
            if not (prev is not None and tuple(prev.shape) == (2, H, H) and os.environ.get("PM_WARM_INVERSE", "1") == "1"):
                # ONE launch, one workgroup per matrix: the two sweeps run side by side on two CUs
                self._call("spd_inverse", "pm_spd_inverse_batch_f64", at(o), H, nHH, _ptr(dadd), H, None, at(o_inv), H, nHH,
                           at(o_inv + 2 * nHH), 2, st)
                prev = whole[o_inv:o_inv + 2 * nHH].view(2, H, H).clone()
            work = self._buf("spd_warm_work", (2 * int(_lib.load().pm_spd_inverse_warm_work_len(H)),))
            self._call("spd_inverse", "pm_inverse_warm_batch_f64", at(o), H, nHH, _ptr(dadd), H, _ptr(prev), nHH,
                       _ptr(work), at(o_inv), nHH, at(o_inv + 2 * nHH), at(o_inv + 2 * nHH + 4), 2, 2, st)
            self._inv_prev = whole[o_inv:o_inv + 2 * nHH].view(2, H, H)     # (`whole` is this step's own tensor: no copy)

        def pack_and_invert():
            """[sum_ss | sum_zz | sum_s | sum_sz | sum |y|^2] straight from the E-step kernel's buffer, then the inverses."""
            st = self._stream()
            self._call("pack_stats", "pm_gsc_pack_stats_f64", _ptr(raw[0]), H, _ptr(yy),
                       ctypes.c_void_p(packed.data_ptr() + 8 * o), st)
            invert(st)

        # On one rank the inverses need nothing but the E-step kernel's own sums: their chain of eight small launches (~60 us on
        # a mostly idle device) runs on a stream of its own BESIDE the contraction over the datapoints instead of behind it.
        inv_early = None
        if (raw is not None and my_N and whole.is_cuda and H <= 256 and getattr(comm, "size", 1) == 1 and self.overlap_moments
                and self.early_inverse and self.timer is None):
            s3 = getattr(self, "_inv_stream", None)
            if s3 is None:
                s3 = self._inv_stream = torch.cuda.Stream(device=self.device)
            fork3 = torch.cuda.Event()
            fork3.record()
            s3.wait_event(fork3)
            with torch.cuda.stream(s3):
                pack_and_invert()
                inv_early = torch.cuda.Event()
                inv_early.record(s3)
        if my_N:
            s = self._stream()
            lists = getattr(raw[0], "_pm_lists", None) if (raw is not None and big is not None and self._in_step) else None
            if lists is not None and lists[4] is big:
                # the same product in two parts (pm_gsc_estep_lists_f64): rows of xsz with a handful of entries above the
                # threshold from their lists (one stream over [Y | xs | xsz]), the others -- a fifth of the datapoints at
                # config 4: those no selected state explains spread their weight over every singleton -- gathered into the
                # MFMA GEMM by a device-side row list
                # (the outer products of the lists for the two H x H blocks + a stream over the Y columns only was measured:
                # same EM iteration -- scratch/gsc_list_pairs_r04.hip)
                side = None
                if self.overlap_moments and self.timer is None:
                    # the gathered GEMM (MFMA-bound, one of its workgroups fits beside a sparse-product workgroup on a CU) on a
                    # second stream: both add into `packed` with atomics, the main stream joins before anything reads it
                    side = getattr(self, "_side_stream", None)
                    if side is None:
                        side = self._side_stream = torch.cuda.Stream(device=self.device)
                    fork = torch.cuda.Event()
                    fork.record()
                if self.list_pairs and H % 64 == 0:
                    # the two H x H blocks of the listed datapoints as outer products of their lists (pm_gsc_list_pairs_f64):
                    # the sparse product then streams the D columns of Y only -- half the bytes at config 4
                    self._call("stats_sparse", "pm_wp_sparse_t_f64", _ptr(lists[0]), _ptr(lists[1]), _ptr(big), ldx,
                               _ptr(packed), H, my_N, H, D, s)
                    self._call("stats_pairs", "pm_gsc_list_pairs_f64", _ptr(lists[0]),
                               ctypes.c_void_p(lists[1].data_ptr() + 8 * 16 * my_N), _ptr(lists[1]), my_N, H,
                               ctypes.c_void_p(packed.data_ptr() + 8 * nWp), s)
                else:
                    self._call("stats_sparse", "pm_wp_sparse_t_f64", _ptr(lists[0]), _ptr(lists[1]), _ptr(big), ldx,
                               _ptr(packed), H, my_N, H, D + 2 * H, s)
                if side is not None:
                    side.wait_event(fork)
                    s = ctypes.c_void_p(side.cuda_stream)
                self._call("stats_gemm", "pm_gemm_tn_acc_rows_f64", _ptr(big), ldx, _ptr(xsz), ldx, _ptr(packed), H,
                           D + 2 * H, H, _ptr(lists[2]), _ptr(lists[3]), my_N, my_N, s)
                if side is not None:
                    join = torch.cuda.Event()
                    join.record(side)
                    torch.cuda.current_stream().wait_event(join)
            elif big is not None:      # [Y | xs | xsz]^T . xsz -> [Wp ; xs^T xsz ; xsz^T xsz]: the head of the packed buffer
                self._call("stats_gemm", "pm_gemm_tn_acc_f64", _ptr(big), ldx, _ptr(xsz), ldx, _ptr(packed), H,
                           D + 2 * H, H, my_N, s)
            else:
                self._call("stats_gemm", "pm_gemm_tn_acc_f64", _ptr(Y), D, _ptr(xsz), ldx, _ptr(packed), H, D, H, my_N, s)
            if big is not None:
                pass
            elif paired:    # [xs | xsz]^T . xsz -> the (2H, H) block [xs^T xsz ; xsz^T xsz] of the packed buffer
                self._call("moment_gemm", "pm_gemm_tn_acc_f64", _ptr(xs), ldx, _ptr(xsz), ldx,
                           ctypes.c_void_p(packed.data_ptr() + 8 * nWp), H, 2 * H, H, my_N, s)
            else:
                self._call("moment_gemm", "pm_gemm_tn_acc_f64", _ptr(xs), H, _ptr(xsz), H,
                           ctypes.c_void_p(packed.data_ptr() + 8 * nWp), H, H, H, my_N, s)
                self._call("moment_gemm", "pm_gemm_tn_acc_f64", _ptr(xsz), H, _ptr(xsz), H,
                           ctypes.c_void_p(packed.data_ptr() + 8 * (nWp + nHH)), H, H, H, my_N, s)
        if inv_early is not None:
            pass
        elif raw is not None:
            self._call("pack_stats", "pm_gsc_pack_stats_f64", _ptr(raw[0]), H, _ptr(yy),
                       ctypes.c_void_p(packed.data_ptr() + 8 * o), self._stream())
        else:
            packed[o:o + nHH] = sum_ss.reshape(-1)
            packed[o + nHH:o + 2 * nHH] = sum_zz.reshape(-1)
            sums = suff_stats.get('_sums')               # column sums from the E-step kernel, when it made suff_stats
            packed[o2:o2 + H] = sums[0] if sums is not None else xs.sum(dim=0)
            packed[o2 + H:o2 + 2 * H] = sums[1] if sums is not None else xsz.sum(dim=0)
            packed[o2 + 2 * H] = yy[0]
        comm.allreduce_device(packed)      # replaces gsc_et.py:608-610,620,668,671,713
        data_sq = self._data_second_moment(res) if 'sigma_sq' in self.to_learn else None
        st = self._stream()
        have_inv = packed.is_cuda and H <= 256
        if inv_early is not None:
            torch.cuda.current_stream(self.device).wait_event(inv_early)
        elif have_inv:
            invert(st)
        Wt_next = None
        self._seed = None
        if have_inv and 'W' in self.to_learn and self.sigma_sq_type == 'scalar' and self.speculate:
            # W_new^T = (sum xpt_szsz)^-T . Wp^T on the device too (gsc_et.py:625; the inverse sits there transposed): the
            # next step's scores GEMM can then start before the host has even seen this step's result
            Wt_next = whole[o_wt:o_wt + nWp].view(H, D)
            self._gemm_nt(whole[o_inv + nHH:o_inv + 2 * nHH].view(H, H), whole[:nWp].view(D, H), Wt_next, "solve_gemm")
        # The rest of the update is H- and H x H-sized (gsc_et.py:640-713): done on the device as well
        # (pm_gsc_mstep_finish_f64), the next E-step's tables exist there before the host has seen anything, and inside
        # an EM loop on a flat annealing schedule that E-step is launched right here (E_step adopts it iff it is called
        # with exactly the parameters this M-step returns).  The host still receives everything with the one download.
        fin = None
        scores_side = None
        if Wt_next is not None and self.speculate_estep and my_N and old_dev is not None:
            if self.overlap_scores and self.timer is None:
                # the scores GEMM needs W_new^T only: it starts here on the second stream while the main one still runs the
                # Gram product, the one-workgroup finish kernel and the download (0.04 ms of a mostly idle device at config 4)
                side = getattr(self, "_side_stream", None)
                if side is None:
                    side = self._side_stream = torch.cuda.Stream(device=self.device)
                solved = torch.cuda.Event()
                solved.record()
                side.wait_event(solved)
                with torch.cuda.stream(side):
                    A_side = self._gemm_nt(Y, Wt_next, self._buf("scores_spec", (my_N, H)), "scores_gemm")
                    scores_side = (A_side, torch.cuda.Event())
                    scores_side[1].record(side)
            G_next = self._gemm_nt(Wt_next, Wt_next, torch.empty((H, H), dtype=torch.float64, device=self.device),
                                   "gram_gemm")
            learn = sum(bit for bit, k in ((1, 'pi'), (2, 'mu'), (4, 'psi_sq'), (8, 'sigma_sq')) if k in self.to_learn)
            tdev = torch.empty(9 * H, dtype=torch.float64, device=self.device)
            self._call("mstep_finish", "pm_gsc_mstep_finish_f64", at(nWp), at(nWp + nHH), at(o), at(o + nHH), at(o_inv),
                       at(o2), at(o2 + H), at(o2 + 2 * H), _ptr(G_next), _ptr(old_dev),
                       ctypes.c_double(float(N)), D, H, learn, at(o_par), _ptr(tdev), st)
            fin = {"G": G_next, "psi": whole[o_par + 2 * H:o_par + 2 * H + nHH].view(H, H), "tdev": tdev, "out": None}
            if self.deterministic:
                # the next pass's quanta (and its M-step's) from the parameters as the device holds them: bounds as in
                # _det_quanta, evaluated by one small kernel behind the finish kernel
                fin["det"] = self._det_dev_quanta(res, G_next, fin["psi"], tdev)

        def after_copy():
            if fin is None:
                self._speculate(res, Wt_next)
                return
            # zero fills of the next pass / the next M-step: enqueued while the scores GEMM still runs beside this stream
            zeros = (torch.zeros(int(_lib.load().pm_gsc_stats_len(H)), dtype=torch.float64, device=self.device),
                     torch.zeros(1, dtype=torch.int32, device=self.device))
            self._whole_next = torch.zeros(n_whole, dtype=torch.float64, device=self.device)
            if scores_side is not None:
                A = scores_side[0]
                torch.cuda.current_stream(self.device).wait_event(scores_side[1])
            else:
                A = self._gemm_nt(Y, Wt_next, self._buf("scores_spec", (my_N, H)), "scores_gemm")
            self._seed = {"ykey": res["key"], "Wt": Wt_next, "G": fin["G"], "A": A, "W_host": None}
            if self._in_step and self._next_anneal is not None:
                fin["out"] = self._launch_estep(res, A, fin["G"], fin["psi"], res["ynorm2"], fin["tdev"], 0.0,
                                                self._next_anneal['T'], None, lists=True, zeros=zeros)
                if fin.get("det") is not None:
                    fin["out"][3]._pm_det = fin["det"]

        if packed.is_cuda:
            n_down = o_par + n_par if fin is not None else (o_par if Wt_next is not None else
                                                            (o_wt if have_inv else n_stat))
            host = self._download(whole[:n_down], then=after_copy if Wt_next is not None else None)
        else:
            host = packed.numpy()
        dev_params = host[o_par:o_par + n_par] if fin is not None else None
        # (host time is on the loop's critical path on a slow host -- 0.58 ms of a 1.14 ms step, round 5: W comes back as the
        # transposed VIEW of one contiguous (H, D) copy instead of a strided (D, H) one; the H x H sums are copied out of the
        # pinned buffer only on the host-fallback path that reads them)
        Wc = host[o_wt:o_wt + nWp].reshape(H, D).copy() if Wt_next is not None else None
        W_given = Wc.T if Wc is not None else None
        Wp = host[:nWp].reshape(D, H)
        xs_xsz = host[nWp:nWp + nHH].reshape(H, H)
        xsz_xsz = host[nWp + nHH:nWp + 2 * nHH].reshape(H, H)
        sum_yy = float(host[o2 + 2 * H])

        inverses = None
        if have_inv:
            tail = host[o_inv:o_inv + n_inv]
            piv, acc = tail[2 * nHH:2 * nHH + 4], tail[2 * nHH + 4:]
            # (a rejected start of the general matrix leaves only the inverse of its upper-mirrored stand-in: host then)
            good = np.isfinite(tail).all() and piv[0] > 0 and piv[2] > 0 and piv[0] / piv[1] > 1e-12 \
                and piv[2] / piv[3] > 1e-12 and acc[1] != 0.0
            if good:        # well-conditioned: use the device inverses; else LAPACK on the host as upstream
                inverses = (tail[nHH:2 * nHH].reshape(H, H).T, tail[:nHH].reshape(H, H))    # (zz^-1, (ss + eps I)^-1)
        if inverses is None:
            self._inv_prev = None             # never warm-start the next inverses from rejected ones
            # (counted: a psi_sq asymmetric enough for the symmetrised cold start to be rejected step after step would put
            # every step on the host fallback -- correct results, a silent performance cliff otherwise)
            self.inverse_fallbacks += 1
        if inverses is None or W_given is None or not np.isfinite(W_given).all():
            W_given, self._seed, dev_params = None, None, None       # host fallback: whatever was speculated is void
        W_snap = None
        if W_given is not None:
            # ONE private snapshot of what model_params['W'] will hold, shared by everything that later asks "is this still the
            # W I returned?": an in-place edit by the caller must be seen as a different W
            W_snap = Wc.copy().T
            if self._seed is not None:
                self._seed["W_host"] = W_snap
        if dev_params is not None and np.isfinite(dev_params).all() and dev_params[-1] > 0:
            # the device's own update: what the caller gets is exactly what the launched E-step has used
            model_params['W'] = W_given
            snap = {'W': W_snap}
            for k, lo, hi, shape in (('pi', 0, H, (H,)), ('mu', H, 2 * H, (H,)), ('psi_sq', 2 * H, 2 * H + nHH, (H, H))):
                if k in self.to_learn:
                    model_params[k] = dev_params[lo:hi].reshape(shape).copy()
                    snap[k] = dev_params[lo:hi].reshape(shape).copy()
                else:
                    snap[k] = np.array(model_params[k], dtype=np.float64, copy=True)
            if 'sigma_sq' in self.to_learn:
                model_params['sigma_sq'] = float(dev_params[-1])
            snap['sigma_sq'] = np.array(model_params['sigma_sq'], dtype=np.float64, copy=True)
            # (the parameters as the device holds them: the next M-step's `old` if they come back unchanged; the same private
            # snapshots serve the speculative E-step's check)
            self._dev_params = (whole[o_par:o_par + n_par], snap)
            if fin["out"] is not None:
                self._spec = {"res": res, "T": self._next_anneal['T'], "out": fin["out"], "params": snap}
            return model_params
        sum_xpt_ss = host[o:o + nHH].reshape(H, H).copy()
        sum_xpt_szsz = host[o + nHH:o + 2 * nHH].reshape(H, H).copy()
        sum_xpt_s, sum_xpt_sz = host[o2:o2 + H].copy(), host[o2 + H:o2 + 2 * H].copy()
        with small_blas():
            return self._update(model_params, N, Wp, xs_xsz, xsz_xsz, sum_xpt_s, sum_xpt_sz, sum_xpt_ss,
                                sum_xpt_szsz, sum_yy, data_sq, inverses,
                                None if W_given is None else np.ascontiguousarray(W_given))

    def _det_quanta(self, res, model_params):
        """Deterministic mode (scalar sigma_sq): bounds of the sums the E-step kernel and the moment GEMM accumulate.  Posterior
        means: |kappa| <= |mu| + min(|y| / min_h |W_h|, |Psi| |W_a| (|y| + |W_a| |mu_a|) / sigma^2)  (Lambda >= Psi^-1);
        second moments: kappa^2 + Lambda^-1 <= kappa^2 + max psi_hh."""
        if self.sigma_sq_type != 'scalar':
            raise _lib.HipError("GSC: deterministic mode is built for scalar sigma_sq")
        ymax, ynmax = self._det_data_bounds(res)
        W = np.asarray(model_params['W'], dtype=np.float64)
        mu, psi = np.asarray(model_params['mu'], dtype=np.float64), np.asarray(model_params['psi_sq'], dtype=np.float64)
        s2 = float(model_params['sigma_sq'])
        wn = np.sqrt((W * W).sum(axis=0))
        mumax, g = float(np.abs(mu).max()), self.gamma
        wa = np.sqrt(g) * float(wn.max())
        zb = mumax + min(ynmax / max(float(wn.min()), 1e-300),
                         float(np.abs(psi).sum(axis=1).max()) * wa * (ynmax + wa * np.sqrt(g) * mumax) / s2)
        n = float(res["Y"].shape[0])
        self._det_set("gsc", [n, n * zb, n * (zb * zb + float(np.abs(np.diag(psi)).max()))])
        self._det_set("gemm", [n * max(ymax, 1.0, zb) * zb])

    def _det_dev_quanta(self, res, G, psi, tables):
        """Deterministic mode inside an EM loop: the same bounds from parameters that are on the device only (the M-step's
        solution: ``G`` = W^T W, ``psi``, ``tables`` of pm_gsc_mstep_finish_f64), by pm_gsc_det_quanta_f64 -- installed for the
        E-step kernel, the contraction GEMM and the sparse product.  Returns the record ``_det_dev_install`` re-installs from."""
        made = {"res": res, "G": G, "psi": psi, "tables": tables,
                "q": torch.empty(16, dtype=torch.float64, device=self.device)}
        self._det_dev_install(made)
        return made

    def _det_dev_install(self, made):
        from ._device import _DET_QUANTA_SET as cache
        if all(cache.get(u) is made for u in ("gsc", "gemm", "wp_sparse")):
            return
        ymax, ynmax = self._det_data_bounds(made["res"])
        st = self._stream()
        _lib.call("pm_gsc_det_quanta_f64", _ptr(made["G"]), made["G"].stride(0), _ptr(made["psi"]), _ptr(made["tables"]), self.H,
                  self.gamma, ctypes.c_double(ymax), ctypes.c_double(ynmax), ctypes.c_double(float(made["res"]["Y"].shape[0])),
                  _ptr(made["q"]), st, det=True)
        for u in ("gsc", "gemm", "wp_sparse"):
            cache[u] = made

    def _eps_diag(self, H, eps):
        """[eps ... eps | 0 ... 0]: the diagonal terms of the batched inverse of (sum_ss + eps I, sum_zz)."""
        d = getattr(self, "_eps_diag_dev", None)
        if d is None or d.numel() != 2 * H:
            d = torch.zeros(2 * H, dtype=torch.float64, device=self.device)
            d[:H] = eps
            self._eps_diag_dev = d
        return d

    def _data_second_moment(self, res):
        """sum_n y_n^2 per dimension (diagonal) / sum_n y_n y_n^T (full) over ALL ranks -- constants of the
        data set (gsc_et.py:681, 694), computed once per resident shard."""
        if self.sigma_sq_type == 'scalar':
            return None
        key = "data_sq_" + self.sigma_sq_type
        if res.get(key) is None:
            Y = res["Y"]
            N, D = Y.shape
            if self.sigma_sq_type == 'diagonal':
                acc = torch.zeros(D, dtype=torch.float64, device=self.device)
                zero = torch.zeros(D, dtype=torch.float64, device=self.device)
                if N:
                    self._call("col_moments", "pm_col_moments_f64", _ptr(Y), Y.stride(0), N, D, _ptr(zero), _ptr(acc),
                               self._stream())
            else:
                acc = torch.zeros((D, D), dtype=torch.float64, device=self.device)
                if N:
                    self._call("data_gram", "pm_gemm_tn_acc_f64", _ptr(Y), Y.stride(0), _ptr(Y), Y.stride(0), _ptr(acc), D,
                               D, D, N, self._stream())
            self.comm.allreduce_device(acc)
            res[key] = acc.cpu().numpy()
        return res[key]

    def _update(self, model_params, N, Wp, xs_xsz, xsz_xsz, sum_xpt_s, sum_xpt_sz, sum_xpt_ss, sum_xpt_szsz, sum_yy,
                data_sq=None, inverses=None, W_given=None):
        """The H x H parameter algebra of gsc_et.py:624-716 on the host.  ``inverses``: (sum_xpt_szsz^-1,
        (sum_xpt_ss + eps I)^-1) when the device already produced them."""
        D, eps = self.D, 1e-5
        try:
            if W_given is not None:       # Wp . (sum xpt_szsz)^-1 as the device computed it (and already uses it)
                W_n = W_given
            else:
                W_n = np.dot(Wp, inverses[0] if inverses is not None else np.linalg.inv(sum_xpt_szsz))
        except np.linalg.LinAlgError:
            try:
                noise = np.random.normal(0, eps, self.H)
                W_n = np.dot(Wp, np.linalg.pinv(sum_xpt_szsz + np.outer(noise, noise)))
            except np.linalg.LinAlgError:
                W_n = model_params['W'] + (eps * np.random.normal(0, 1, [self.D, self.H]))

        if 'pi' in self.to_learn:
            pi_eps = 5e-5
            pi_new = sum_xpt_s / N
            pi_new[pi_new <= pi_eps] = pi_eps
            pi_new[pi_new >= (1 - pi_eps)] = 1 - pi_eps
            model_params['pi'] = pi_new
        if 'W' in self.to_learn:
            model_params['W'] = W_n
        if 'mu' in self.to_learn:
            model_params['mu'] = sum_xpt_sz * 1. / (sum_xpt_s + np.finfo(np.float64).eps)
        if 'psi_sq' in self.to_learn:
            mu = model_params['mu']
            psi_sq = np.outer(mu, mu) * sum_xpt_ss + sum_xpt_szsz - 2 * (mu[:, None] * xs_xsz)
            ss_inv = inverses[1] if inverses is not None else np.linalg.inv(sum_xpt_ss + eps * np.eye(self.H))
            model_params['psi_sq'] = (psi_sq * ss_inv) + (eps * np.eye(self.H))
        if 'sigma_sq' in self.to_learn:
            if self.sigma_sq_type == 'full':            # gsc_et.py:677-688
                model_params['sigma_sq'] = (data_sq - W_n @ xsz_xsz @ W_n.T) / N + (eps * np.eye(self.D))
            elif self.sigma_sq_type == 'diagonal':      # gsc_et.py:690-701
                model_params['sigma_sq'] = (data_sq - ((W_n @ xsz_xsz) * W_n).sum(axis=1)) / N + eps
            else:                                       # gsc_et.py:703-713
                WT_outer = np.dot(W_n.T, W_n)
                # trace(xsz_xsz . W^T W) without the H^3 product the reference forms for it (gsc_et.py:706): this line
                # sits on the host's critical path between the M-step's download and the next E-step launch
                my_sigma_sq = sum_yy - float(np.einsum('ij,ji->', xsz_xsz, WT_outer))
                model_params['sigma_sq'] = (my_sigma_sq / N / D) + eps
        return model_params
