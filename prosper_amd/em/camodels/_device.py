"""Device plumbing shared by the HIP-backed component-analysis models: resident data shards,
workspace reuse, asynchronous parameter upload, the one device->host copy per M-step, kernel
launch bookkeeping and the NumPy-compatible handles (``DeviceArray``) results are returned in.
PyTorch supplies memory, streams and process groups; every computation on the hot path is a call
into libprosper_hip.so (include/prosper_hip.h)."""
import ctypes
import os

import numpy as np

from . import CAModel
from ... import _lib
from ...utils import parallel

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

try:
    from threadpoolctl import ThreadpoolController as _ThreadpoolController
except Exception:  # pragma: no cover
    _ThreadpoolController = None

_blas_controller = None


class small_blas(object):
    """Host LAPACK on the H x H matrices of an M-step with a handful of threads: on a 256-core GPU
    host OpenBLAS otherwise wakes every core for a 128 x 128 inverse (measured 16 ms instead of 0.2).
    The library scan of threadpoolctl (0.6 ms) is done once per process."""

    def __init__(self, threads=4):
        global _blas_controller
        if _blas_controller is None and _ThreadpoolController is not None:
            w = np.ones((256, 256))
            np.dot(w, w)                             # OpenBLAS builds its thread pool on the first threaded call:
            np.linalg.inv(w + 256 * np.eye(256))     # do that before any limit is in force (else every call rebuilds it)
            _blas_controller = _ThreadpoolController()
        self._ctx = _blas_controller.limit(limits=threads, user_api="blas") if _blas_controller is not None else None

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


class DeviceArray(object):
    """NumPy-compatible handle of a tensor living in HBM.

    ``np.asarray(handle)`` (or any NumPy function) downloads it once; row selection with an
    index array stays on the device (``CAModel.select_partial_data``)."""

    def __init__(self, tensor, np_dtype=None):
        self.tensor = tensor
        self._np_dtype = np.dtype(np_dtype) if np_dtype is not None else None
        self._host = None
        self.lse = None          # log-evidence per row, attached to 'logpj' handles

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    @property
    def ndim(self):
        return self.tensor.dim()

    @property
    def dtype(self):
        if self._np_dtype is not None:
            return self._np_dtype
        return np.dtype(str(self.tensor.dtype).replace("torch.", ""))

    def __len__(self):
        return self.tensor.shape[0]

    def numpy(self):
        if self._host is None:
            host = self.tensor.detach().cpu().numpy()
            if self._np_dtype is not None and host.dtype != self._np_dtype:
                host = host.astype(self._np_dtype)
            self._host = host
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None and np.dtype(dtype) != a.dtype else a

    def __getitem__(self, idx):
        if isinstance(idx, np.ndarray) and idx.ndim == 1 and idx.dtype.kind in "iub":
            t = torch.from_numpy(np.ascontiguousarray(idx)).to(self.tensor.device)
            out = DeviceArray(self.tensor[t], self._np_dtype)
            if self.lse is not None:
                out.lse = self.lse[t]
            return out
        return self.numpy()[idx]

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, device=%s)" % (self.shape, self.dtype, self.tensor.device)


class LazyCandidates(DeviceArray):
    """``data['candidates']`` as handed out by the fast path of ``select_Hprimes``: the
    selection is deferred.  If ``E_step`` is called next with the same parameters (what
    ``CAModel.step`` / ``compute_lpj`` do) both stages run as ONE fused pass that overlaps with
    the scores GEMM; if anything looks at the handle first (``np.asarray``, indexing, ``M_step``
    with foreign log-joints) the candidates are computed on the spot.  Either way the values
    are those of bsc_et.py:98-115."""

    def __init__(self, model, ticket, shape):
        self._model = model
        self._ticket = ticket
        self._shape = tuple(shape)
        self._np_dtype = np.dtype(np.int64)
        self._host = None
        self.lse = None

    @property
    def tensor(self):
        if self._ticket["cand"] is None:
            self._model._materialize_candidates(self._ticket)
        return self._ticket["cand"]

    @property
    def pending(self):
        return self._ticket["cand"] is None

    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return 2

    def __len__(self):
        return self._shape[0]

    def __repr__(self):
        return "LazyCandidates(shape=%s, pending=%s)" % (self._shape, self.pending)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_DET_QUANTA_SET = {}      # unit -> the quanta last installed in libprosper_hip_det.so's symbols (process-global, like them)


class _AnnealAt(object):
    """Read-only view of a schedule at another position (``LinearAnnealing.__getitem__`` reads ``cur_pos``,
    annealing.py:90-107); ``pos`` None: the schedule as it stands (objects without a position, e.g. a plain mapping)."""

    def __init__(self, anneal, pos):
        self._a, self._pos = anneal, pos

    def __getitem__(self, name):
        a = self._a
        if self._pos is None:
            return a[name]
        old = a.cur_pos
        a.cur_pos = self._pos
        try:
            return a[name]
        finally:
            a.cur_pos = old


class KernelTimer(object):
    """HIP-event timing of individual kernel launches on the stream they are enqueued on
    (torch's current stream, which is the one handed to the C ABI).  bench.py attaches one
    to a model to obtain per-kernel average durations inside the timed region."""

    def __init__(self, only=None, stride=1):
        self.events = {}
        self.only = set(only) if only else None    # labels to time (None = all)
        self.stride = max(1, int(stride))          # time every stride-th launch of a label
        self._count = {}

    def launch(self, label, fn):
        n = self._count[label] = self._count.get(label, 0) + 1
        if (self.only is not None and label not in self.only) or (n - 1) % self.stride:
            fn()
            return
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        fn()
        end.record()
        self.events.setdefault(label, []).append((start, end))

    def summary(self):
        """label -> (launches, average milliseconds); synchronises."""
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}



class DeviceCAModel(CAModel):
    """CAModel whose select_Hprimes / E_step / M_step run on one MI355X."""

    def __init__(self, D, H, Hprime, gamma, to_learn=['W', 'pi', 'sigma'], comm=parallel.COMM_WORLD,
                 device=None):
        CAModel.__init__(self, D, H, Hprime, gamma, to_learn, comm)
        self._device = device
        self._data = {}          # resident data shard: key, Y, ynorm2
        self._par = {}           # per-step parameter products
        self._ws = {}            # workspaces keyed by name
        self.timer = None        # optional KernelTimer (bench.py)
        # True: every kernel runs from libprosper_hip_det.so (the same sources with -DPM_DETERMINISTIC: addends of the M-step's
        # atomics rounded to a common quantum first, so their sums do not depend on the order they land in -- pm_common.h) and
        # the host takes no shortcut whose operand order varies: two identical EM loops then agree bit for bit.  Opt-in: a few
        # per cent slower, and sums carry the rounding error of their largest entries on all of them.
        self.deterministic = False
        self._pin = {}           # pinned staging buffers for asynchronous parameter uploads
        self._pin_out = {}       # pinned buffers of the device->host copies (one per M-step), by slot
        self.speculate = os.environ.get('PM_SPECULATE', '1') == '1'   # next step's GEMMs behind the M-step download
        self._seed_rec = None    # what _seed_next left on the device for the next select_Hprimes
        self._mstep_res = None   # resident shard of the M-step in progress (for _seed_next)

    def _state_masks(self):
        """uint16 mask per multi-cause state: bit j <=> candidate position j is on."""
        SM = self.state_matrix.astype(np.int64)
        if not SM.size:
            return np.zeros(0, np.uint16)
        return (SM << np.arange(self.Hprime)[None, :]).sum(axis=1).astype(np.uint16)

    def _u16_dev(self, arr):
        """uint16 payloads travel as int16 tensors (same bytes)."""
        arr = np.ascontiguousarray(arr, dtype=np.uint16)
        if not arr.size:
            return torch.zeros(1, dtype=torch.int16, device=self.device)
        return torch.from_numpy(arr.view(np.int16).copy()).to(self.device)

    @property
    def device(self):
        if self._device is None:
            if torch is None or not torch.cuda.is_available():
                raise _lib.HipError("BSC_ET needs a HIP device: the hot path has no CPU fallback")
            self._device = torch.device("cuda", torch.cuda.current_device())
        return torch.device(self._device)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _call(self, label, entry, *args):
        """Enqueue one C-ABI entry point on the current stream (raises on a bad status)."""
        det = self.deterministic
        if self.timer is None:
            _lib.call(entry, *args, det=det)
        else:
            self.timer.launch(label, lambda: _lib.call(entry, *args, det=det))

    # ---- deterministic reductions (libprosper_hip_det.so) ------------------------------------------------------------
    @staticmethod
    def _magic(bound):
        """1.5 * 2^e with 2^(e-1) >= bound: adding and subtracting it rounds a value to a multiple of 2^(e-52) (PM_Q)."""
        b = float(bound)
        if not np.isfinite(b) or b <= 0.0:
            return 0.0
        return 1.5 * 2.0 ** (int(np.ceil(np.log2(b))) + 1)

    def _det_set(self, unit, bounds):
        """Install the quanta of one kernel family (include/prosper_hip.h: pm_det_set_quanta) ahead of its next launches:
        ``bounds[c]`` = what no partial sum of category c can exceed on this shard with these parameters."""
        vals = tuple([self._magic(b) for b in bounds] + [0.0] * (8 - len(bounds)))
        # (a quantum is a power of two: it moves only when its bound crosses one -- the six uploads per step of a steady EM loop
        # were six small copies on the stream's critical path for values already there.  The cache is per PROCESS, like the
        # symbols: another model's different bounds invalidate it.)
        cache = _DET_QUANTA_SET
        if cache.get(unit) == vals:
            return
        cache[unit] = vals
        M = (ctypes.c_double * 8)(*vals)
        keep = self.__dict__.setdefault("_det_keep", [])
        keep.append(M)
        del keep[:-16]
        _lib.call("pm_det_set_quanta", _lib.DET_UNITS[unit], M, self._stream(), det=True)

    def _det_data_bounds(self, res):
        """(max |y_nd|, max |y_n|) over the resident shard: constants of the data, once per shard."""
        b = res.get("det_bounds")
        if b is None:
            Y = res["Y"]
            b = res["det_bounds"] = ((float(Y.abs().max()), float(res["ynorm2"].max().sqrt())) if Y.shape[0] else (0.0, 0.0))
        return b

    def _anneal_point(self, anneal):
        """What of an annealing point decides the inputs of an E-step: temperature, prior annealing, truncation, partial
        data, and the parameter noise of ``noisify_params`` (prosper/em/__init__.py:63-107)."""
        return (anneal['T'], bool(anneal['anneal_prior']), anneal['Ncut_factor'], anneal['partial']) + \
            tuple(anneal[p + '_noise'] for p in sorted(self.noise_policy))

    def _predict_anneal(self, anneal):
        """The annealing point of the NEXT ``step`` -- what the M-step launches the next E-step with (_speculate_estep) -- or
        None when it is not known.  A schedule is a pure function of its position (annealing.py:90-107; ``next`` ignores the
        gain and always accepts, :116-130), so for a ``LinearAnnealing`` it is read at ``cur_pos + 1``; but only a
        predictor that was RIGHT about the current step is trusted (a caller that does not advance the schedule between
        steps gets the flat predictor: same point again), nothing is predicted past the schedule's end, and a next step
        with parameter noise or partial data has inputs nobody knows yet.  A wrong prediction costs a dropped pass, never a
        wrong result (E_step compares the scalars the pass was launched with)."""
        sig = self._anneal_point(anneal)
        prev_sig, prev_peek = getattr(self, "_sig_hist", (None, None))
        peek, end = None, False
        cur, steps = getattr(anneal, "cur_pos", None), getattr(anneal, "steps", None)
        if isinstance(cur, int) and isinstance(steps, int):
            if cur + 1 >= steps:
                end = True
            else:
                peek = _AnnealAt(anneal, cur + 1)
        nxt = None
        if not end:
            if peek is not None and prev_peek is not None and sig == prev_peek:
                nxt = peek
            elif sig == prev_sig:
                nxt = peek if (peek is not None and self._anneal_point(peek) == sig) else _AnnealAt(anneal, cur)
        self._sig_hist = (sig, self._anneal_point(peek) if peek is not None else None)
        self._flat_schedule = (sig == prev_sig)
        if nxt is not None:
            nsig = self._anneal_point(nxt)
            if nsig[3] not in (0, 1) or any(nsig[4:]):
                nxt = None
        return nxt

    def step(self, anneal, model_params, my_data):
        """CAModel.step (camodels/__init__.py:163-193); the E-step knows that the M-step follows with the same arguments."""
        was, self._in_step = getattr(self, "_in_step", False), True
        self._step_id = getattr(self, "_step_id", 0) + 1
        try:
            return CAModel.step(self, anneal, model_params, my_data)
        finally:
            self._in_step = was

    def _dsc_estep(self, anneal, stats_name, par, res, cand, tab, S, prior, P, Kt, pi_key):
        """DSC / TSC E-step launch.  Inside ``step`` with no data truncation ahead the sixteen-lane kernel also produces the
        M-step's row statistics (pm_dsc_estep_mstats_f64: E[s] rows and their non-zero lists, Wq, qdiag, value counts,
        scalars) from the exponentials its log-sum-exp evaluates anyway -- ``M_step`` then skips its pass over the
        log-joints.  Returns the DeviceArray of log-joints with ``.lse`` and, fused, ``.mstats``."""
        N = res["Y"].shape[0]
        H, D, Hp = self.H, self.D, self.Hprime
        lib = _lib.load()
        logpj = torch.empty((N, Kt), dtype=torch.float64, device=self.device)
        lse = torch.empty((N,), dtype=torch.float64, device=self.device)
        out = DeviceArray(logpj)
        out.lse = lse
        out.mstats = None
        if not N:
            return out
        st = self._stream()
        if self.deterministic:
            self._det_dsc_quanta(res, par, P, prior, Kt)
        fuse = (getattr(self, "_in_step", False) and getattr(self, "fuse_mstats", True) and anneal['Ncut_factor'] <= 0.0
                and bool(lib.pm_dsc_estep_mstats_supported(H, Hp, S, int(P.K), int(P.flags))))
        if fuse:
            stats = self._buf(stats_name, (lib.pm_dsc_stats_len(H, D),))
            stats.zero_()
            expect = self._buf("expect", (N, H))
            nz = None
            if getattr(self, "sparse_wp", True):
                nz = (self._buf("nz_idx", (N, 16), torch.int16), self._buf("nz_val", (N, 16)))
            self._call("estep_mstats", "pm_dsc_estep_mstats_f64", _ptr(par["A"]), H, _ptr(par["G"]), _ptr(res["ynorm2"]),
                       _ptr(cand), _ptr(tab), S, _ptr(prior), ctypes.byref(P), N, H, D, Hp, _ptr(logpj), Kt, _ptr(lse),
                       _ptr(expect), H, _ptr(stats), _ptr(nz[0]) if nz else None, _ptr(nz[1]) if nz else None, st)
            out.mstats = {"stats": stats, "expect": expect, "nz": nz, "res": res, "cand": cand,
                          "P": (float(P.ecoef), float(P.pscale), int(P.flags)), "pi": np.array(pi_key, dtype=np.float64, copy=True)}
        else:
            self._call("estep", "pm_dsc_estep_f64", _ptr(par["A"]), H, _ptr(par["G"]), _ptr(res["ynorm2"]), _ptr(cand),
                       _ptr(tab), S, _ptr(prior), ctypes.byref(P), N, H, Hp, _ptr(logpj), Kt, _ptr(lse), st)
        return out

    def _det_dsc_quanta(self, res, par, P, prior, Kt):
        """Deterministic mode, DSC / TSC: bounds of the statistics' partial sums (latent values v_k, |v| <= vmax) -> quanta of
        the row kernels, the sparse product and the dense GEMM behind its gate (pm_common.h, PM_Q).  Set ahead of the E-step,
        whose parameters the M-step of the same EM step shares."""
        ymax, ynmax = self._det_data_bounds(res)
        W = np.asarray(par["W"], dtype=np.float64)
        wn = float(np.sqrt((W * W).sum(axis=0)).max()) if W.size else 0.0
        vmax = float(max(abs(P.values[k]) for k in range(int(P.K))))
        emax = (ynmax + self.gamma * vmax * wn) ** 2
        lpmax = abs(P.pscale) * float(prior.abs().max()) + abs(P.ecoef) * emax + np.log(max(Kt, 2))
        n = float(res["Y"].shape[0])
        self._det_set("dsc", [n * max(1.0, vmax * vmax), n * emax, n * lpmax])
        self._det_set("wp_sparse", [n * vmax * ymax])
        self._det_set("gemm", [n * max(1.0, vmax) * ymax, n * ymax])

    def _dsc_fused_stats(self, logpj, res, cand, P, pi_key, lse_cut):
        """The statistics workspace the E-step pass has already filled for exactly this M-step, or None."""
        ms = getattr(logpj, "mstats", None) if isinstance(logpj, DeviceArray) else None
        if ms is None:
            return None
        logpj.mstats = None
        if (ms["res"] is res and ms["cand"] is cand and lse_cut == float("-inf")
                and ms["P"] == (float(P.ecoef), float(P.pscale), int(P.flags))
                and np.array_equal(ms["pi"], np.asarray(pi_key, dtype=np.float64))):
            return ms
        return None

    def _rows_and_wp(self, rows_args, lp_ld, expect, Y, stats, my_N, K, flags, Hp, S, fused=None, cut_dev=None):
        """DSC / TSC M-step: the per-datapoint pass (pm_dsc_mstep_rows[_nz]_f64) and Wp = E[s]^T Y.  Where the
        sixteen-lane kernel applies the pass also leaves the non-zero lists of E[s] and Wp is accumulated from them
        (pm_wp_sparse_f64); the dense product follows behind the device-side gate (last scalar of `stats`: rows whose
        list overflowed) and only does work then.  ``fused``: the record of an E-step pass that has already produced the
        row statistics (``_dsc_estep``): only the product is left."""
        H, D = self.H, self.D
        lib = _lib.load()
        st = self._stream()
        gate = ctypes.c_void_p(stats.data_ptr() + 8 * (lib.pm_dsc_stats_len(H, D) - 1))
        if fused is not None:
            nz = fused["nz"]
            if nz is not None:
                self._call("stats_sparse", "pm_wp_sparse_f64", _ptr(nz[0]), _ptr(nz[1]), _ptr(Y), Y.stride(0), _ptr(stats),
                           D, gate, my_N, H, D, st)
                self._call("stats_gemm", "pm_gemm_tn_acc_gated_f64", _ptr(expect), H, _ptr(Y), D, _ptr(stats), D, H, D,
                           my_N, gate, st)
            else:
                self._call("stats_gemm", "pm_gemm_tn_acc_f64", _ptr(expect), H, _ptr(Y), D, _ptr(stats), D, H, D, my_N, st)
            return
        sparse = (getattr(self, "sparse_wp", True) and Y.is_cuda and H <= 256
                  and bool(lib.pm_dsc_rows16_supported(H, Hp, S, K, flags)))
        if cut_dev is not None:
            # ``cut_dev``: the data-truncation cut as the radix select left it on the device (round 6: no host round trip
            # between the select and this pass -- on a slow host the device idled a quarter of the step there)
            rows_args = rows_args[:4] + (_ptr(cut_dev),) + rows_args[4:]
            nzb = (self._buf("nz_idx", (my_N, 16), torch.int16), self._buf("nz_val", (my_N, 16))) if sparse else (None, None)
            self._call("mstep_rows", "pm_dsc_mstep_rows_cutp_f64", *(rows_args + (_ptr(nzb[0]), _ptr(nzb[1]), st)))
            if sparse:
                self._call("stats_sparse", "pm_wp_sparse_f64", _ptr(nzb[0]), _ptr(nzb[1]), _ptr(Y), Y.stride(0), _ptr(stats),
                           D, gate, my_N, H, D, st)
                self._call("stats_gemm", "pm_gemm_tn_acc_gated_f64", _ptr(expect), H, _ptr(Y), D, _ptr(stats), D, H, D,
                           my_N, gate, st)
            else:
                self._call("stats_gemm", "pm_gemm_tn_acc_f64", _ptr(expect), H, _ptr(Y), D, _ptr(stats), D, H, D, my_N, st)
            return
        if sparse:
            nz_idx, nz_val = self._buf("nz_idx", (my_N, 16), torch.int16), self._buf("nz_val", (my_N, 16))
            self._call("mstep_rows", "pm_dsc_mstep_rows_nz_f64", *(rows_args + (_ptr(nz_idx), _ptr(nz_val), st)))
            self._call("stats_sparse", "pm_wp_sparse_f64", _ptr(nz_idx), _ptr(nz_val), _ptr(Y), Y.stride(0), _ptr(stats),
                       D, gate, my_N, H, D, st)
            self._call("stats_gemm", "pm_gemm_tn_acc_gated_f64", _ptr(expect), H, _ptr(Y), D, _ptr(stats), D, H, D,
                       my_N, gate, st)
        else:
            self._call("mstep_rows", "pm_dsc_mstep_rows_f64", *(rows_args + (st,)))
            self._call("stats_gemm", "pm_gemm_tn_acc_f64", _ptr(expect), H, _ptr(Y), D, _ptr(stats), D, H, D, my_N, st)

    def _buf(self, name, shape, dtype=None):
        """Reusable device workspace (no allocation inside the EM loop once warm)."""
        dtype = dtype or torch.float64
        t = self._ws.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._ws[name] = t
        return t

    @staticmethod
    def _probe(y):
        """Cheap fingerprint of a host array (sum over ~16 evenly spaced rows): catches most in-place edits of an
        array that is already resident; ``invalidate_data()`` is the contract for the rest."""
        n = y.shape[0]
        return float(y[::max(1, n // 16)].sum()) if n else 0.0

    def _resident(self, y):
        """Device copy of the data shard + |y_n|^2, uploaded once and kept in HBM.  A shard is recognised by the
        IDENTITY of the caller's array (the record holds a reference, so its address cannot be recycled for
        another array -- ``select_partial_data`` builds a fresh ``y[sel]`` every step); tensors also by their
        version counter, host arrays by a strided fingerprint."""
        src = y
        if isinstance(y, DeviceArray):
            y = y.tensor
        d = self._data
        if d and d.get("src") is src:
            if torch.is_tensor(y):
                if d["ver"] == (y._version, tuple(y.shape)):
                    return d
            elif isinstance(y, np.ndarray) and d["ver"] == (self._probe(y), y.shape):
                return d
        if torch.is_tensor(y):
            ver = (y._version, tuple(y.shape))
            Y = y.to(device=self.device, dtype=torch.float64).contiguous()
        else:
            y = np.asarray(y)
            ver = (self._probe(y), y.shape)
            Y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).to(self.device)
        N, D = Y.shape
        assert D == self.D
        yn = torch.empty(N, dtype=torch.float64, device=self.device)
        if N:
            self._call("row_sqnorm", "pm_row_sqnorm_f64", _ptr(Y), D, N, D, _ptr(yn), self._stream())
        self._data_gen = getattr(self, "_data_gen", 0) + 1
        self._data = {"key": ("shard", self._data_gen), "src": src, "ver": ver, "Y": Y, "ynorm2": yn}
        self._par = {}
        return self._data

    def _global_count(self, res, my_N):
        """N = sum of the shard sizes (``comm.allreduce(my_N)``, bsc_et.py:225): fetched once per resident
        data set -- on an nccl-only group every host-side allreduce is a blocking device round trip."""
        if res.get("N_global") is None:
            res["N_global"] = self.comm.allreduce(my_N)
        return res["N_global"]

    def _data_moments(self, data):
        """Per-dimension mean and variance of the (sharded) data on the device: two passes over the
        resident shard (pm_col_moments_f64) and two tiny all-reduces, instead of the reference's two
        host passes over N x D (``parallel.allmean``, camodels/__init__.py:209-213)."""
        res = self._resident(data['y'])
        Y = res["Y"]
        my_N, D = Y.shape
        N = self._global_count(res, my_N)
        s1 = torch.zeros(D, dtype=torch.float64, device=self.device)
        if my_N:
            self._call("col_moments", "pm_col_moments_f64", _ptr(Y), Y.stride(0), my_N, D, None, _ptr(s1), self._stream())
        self.comm.allreduce_device(s1)
        mean = s1 / N
        s2 = torch.zeros(D, dtype=torch.float64, device=self.device)
        if my_N:
            self._call("col_moments", "pm_col_moments_f64", _ptr(Y), Y.stride(0), my_N, D, _ptr(mean), _ptr(s2), self._stream())
        self.comm.allreduce_device(s2)
        host = self._download(torch.cat([mean, s2 / N]), slot="moments").copy()
        return host[:D], host[D:]

    def standard_init(self, data):
        """W = data mean + N(0, (sigma_init/4)^2) per column, sigma = mean per-dimension std, pi = 1/H
        (camodels/__init__.py:196-235); the two passes over the data run on the device."""
        W_mean, sigma_sq = self._data_moments(data)
        D = W_mean.shape[0]
        assert D == self.D
        sigma_init = np.sqrt(sigma_sq).sum() / D
        W_init = W_mean[:, None] + np.random.normal(scale=sigma_init / 4., size=[D, self.H])
        return {'W': W_init, 'pi': 1. / self.H, 'sigma': sigma_init}

    # ------------------------------------------------------------------ data generation on the device
    def generate_data(self, model_params, my_N, device=False, seed=None):
        """``device=False``: the reference's host generator, same NumPy RNG stream (camodels/__init__.py:104-122 and
        the models' own versions).  ``device=True``: the same distribution drawn on the GPU (Philox streams of a
        ``torch.Generator``; ``seed`` makes it reproducible) -- the reference's generators are Python loops over
        datapoints (minutes at the sizes of BASELINE configs 3 / 5); returns ``DeviceArray`` handles."""
        if not device:
            return self._generate_data_host(model_params, my_N)
        return self.generate_data_device(model_params, my_N, seed)

    def _generate_data_host(self, model_params, my_N):
        return CAModel.generate_data(self, model_params, my_N)

    def _gen(self, seed):
        g = torch.Generator(device=self.device)
        if seed is None:
            g.seed()
        else:
            g.manual_seed(int(seed))
        return g

    def _mix_linear(self, S, W_DH):
        """y = S . W^T for latent values S (N,H) f64 on the device and W (D,H): the f64 MFMA GEMM of the hot path."""
        W = torch.from_numpy(np.ascontiguousarray(np.asarray(W_DH, dtype=np.float64))).to(self.device)      # (D,H)
        y = torch.empty((S.shape[0], W.shape[0]), dtype=torch.float64, device=self.device)
        if S.shape[0]:
            self._gemm_nt(S.contiguous(), W, y, "generate_gemm")
        return y

    def _draw_latents(self, model_params, my_N, g):
        """Binary latents, P(s_h = 1) = pi (scalar or per latent) -> (N,H) bool on the device."""
        pi = torch.as_tensor(np.asarray(model_params['pi'], dtype=np.float64)).to(self.device)
        return torch.rand((my_N, self.H), generator=g, device=self.device, dtype=torch.float64) < pi

    def _superpose(self, model_params, s, g):
        """Noise-free data for latents ``s`` (N,H) on the device; the linear superposition s . W^T here, models with
        another combination rule override it."""
        return self._mix_linear(s.to(torch.float64), model_params['W'])

    def generate_data_device(self, model_params, my_N, seed=None, noise_on=True):
        g = self._gen(seed)
        s = self._draw_latents(model_params, my_N, g)
        y = self._superpose(model_params, s, g)
        if noise_on:
            y += float(model_params['sigma']) * torch.randn(y.shape, generator=g, device=self.device, dtype=torch.float64)
        return {'y': DeviceArray(y), 's': DeviceArray(s)}

    def invalidate_data(self):
        """Forget the resident shard (call after modifying ``my_data['y']`` in place)."""
        self._data = {}
        self._par = {}

    def _gemm_nt(self, A, B, out, label="gemm_nt"):
        M, K = A.shape
        N = B.shape[0]
        # small outputs (Gram matrices, the H x H x D solve products): the deterministic one-workgroup-per-tile kernel --
        # every rank holding the same operands gets the same bits, and no zero fill + K-slice atomics
        fn = "pm_gemm_nt_small_f64" if (M <= 512 and N <= 512) else "pm_gemm_nt_f64"
        # (a matrix of ONE row reports whatever stride its history left -- (D, 1).t().contiguous() keeps (1, 1): H = 1 -- and
        # any leading dimension >= its row length describes it)
        ld = lambda t, cols: t.stride(0) if t.shape[0] > 1 else max(int(t.stride(0)), int(cols))
        self._call(label, fn, _ptr(A), ld(A, K), _ptr(B), ld(B, K), _ptr(out), ld(out, N), M, N, K, self._stream())
        return out

    def _upload(self, name, host, keep=False):
        """Asynchronous host -> device copy through a rotating pair of pinned staging buffers
        (a pageable ``.to(device)`` would block the host until the stream drains and stall the
        EM loop at every step boundary).  ``keep``: also return the staging buffer's NumPy view,
        which stays intact until the second-next upload under the same name."""
        slot = self._pin.setdefault(name, {"i": 0, "bufs": [None, None], "evs": [None, None]})
        i = slot["i"] = slot["i"] ^ 1
        buf = slot["bufs"][i]
        if buf is None or buf.shape != host.shape:
            buf = slot["bufs"][i] = torch.empty(host.shape, dtype=torch.float64).pin_memory()
        elif slot["evs"][i] is not None:
            slot["evs"][i].synchronize()          # the copy that last used this buffer has completed
        view = buf.numpy()
        view[...] = host
        dev = torch.empty(host.shape, dtype=torch.float64, device=self.device)
        dev.copy_(buf, non_blocking=True)
        ev = slot["evs"][i] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return (dev, view) if keep else dev

    def _download_async(self, flat, slot):
        """Device -> pinned host copy on the copy stream, behind what the main stream holds NOW.  Returns (NumPy view,
        event to ``synchronize()`` on before reading it)."""
        n = flat.numel()
        buf = self._pin_out.get(slot)
        if buf is None or buf.numel() < n:
            buf = self._pin_out[slot] = torch.empty(n, dtype=torch.float64).pin_memory()
        cs = getattr(self, "_copy_stream", None)
        if cs is None:
            cs = self._copy_stream = torch.cuda.Stream(device=self.device)
        cs.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(cs):
            buf[:n].copy_(flat, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
        return buf[:n].numpy(), ev

    def _download(self, flat, slot="default", then=None):
        """Device -> pinned host copy + wait; returns a NumPy view valid until the next call
        with the same ``slot``.  ``then()`` runs after the copy is enqueued and before the wait:
        work it launches keeps the device busy while the host digests the result."""
        n = flat.numel()
        buf = self._pin_out.get(slot)
        if buf is None or buf.numel() < n:
            buf = self._pin_out[slot] = torch.empty(n, dtype=torch.float64).pin_memory()
        dst = buf[:n]
        main = torch.cuda.current_stream(self.device)
        if then is not None:
            # the copy goes to a stream of its own: what ``then`` enqueues on the main stream (the next step's
            # speculative GEMMs) starts at once instead of queueing behind ~50 us of copy and launch gaps
            cs = getattr(self, "_copy_stream", None)
            if cs is None:
                cs = self._copy_stream = torch.cuda.Stream(device=self.device)
            cs.wait_stream(main)
            with torch.cuda.stream(cs):
                dst.copy_(flat, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cs)
            then()
        else:
            dst.copy_(flat, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(main)
        ev.synchronize()
        return dst.numpy()

    # ---- EM-loop pipelining shared by the linear models whose M-step solves W^T on the device (DSC, TSC) ----
    def _seed_next(self, res, Wt):
        """Next step's Gram matrix and scores from ``Wt`` = W_new^T (H,D), which the M-step has just solved on the
        device -- enqueued behind the M-step's download so they run while the host unpacks it.  ``_take_seed``
        hands them to the next ``select_Hprimes`` if the caller feeds the returned W back unchanged."""
        Y = res["Y"]
        N, H = Y.shape[0], self.H
        A = self._buf("scores_spec", (N, H))
        if self._par.get("A") is A:
            self._par = {}                 # the cached parameters' scores live in the buffer overwritten now
        G = self._gemm_nt(Wt, Wt, torch.empty((H, H), dtype=torch.float64, device=self.device), "gram_gemm")
        if N:
            self._gemm_nt(Y, Wt, A, "scores_gemm")
        self._seed_rec = {"ykey": res["key"], "Wt": Wt, "G": G, "A": A, "W": None}

    def _scores_params(self, W, res):
        """DSC / TSC: device copy of W^T (H,D), the Gram matrix and the scores for the current W and data.  In an EM loop
        the last M-step has left all three on the device (``_seed_next``): W is compared with ITS snapshot first, and once
        per ``step`` -- select_Hprimes, E_step and M_step see the same array object there, and a 256 x 128 comparison costs
        27 us of host time that sits on the loop's critical path (three of them per step until round 4: the device idled
        ~0.1 ms per 0.7 ms iteration waiting for the E-step launch)."""
        W_in = W
        W = np.asarray(W, dtype=np.float64)
        par = self._par
        # The once-per-step shortcut keys on the IDENTITY of the caller's array, so only for an object that is already the
        # float64 ndarray the comparison would read (a converted temporary's id() can be recycled), and the record keeps a
        # reference to it (an id() is only unique among live objects).  Contract: W is not edited in place between
        # select_Hprimes, E_step and M_step of one ``step`` (CAModel.step never does).
        tag = (getattr(self, "_step_id", 0), id(W)) if (getattr(self, "_in_step", False) and W is W_in) else None
        if tag is not None and par.get("checked") == tag and par.get("checked_obj") is W and par.get("ykey") == res["key"]:
            return par
        if getattr(self, "_seed_rec", None) is not None:
            seeded = self._take_seed(W, res)
            if seeded is not None:           # W^T, Gram matrix and scores left on the device by the last M-step
                seeded["checked"], seeded["checked_obj"] = tag, (W if tag is not None else None)
                self._par = seeded
                return seeded
        if par.get("ykey") == res["key"] and par.get("W") is not None and par["W"].shape == W.shape \
                and np.array_equal(par["W"], W):
            par["checked"], par["checked_obj"] = tag, (W if tag is not None else None)
            return par
        Wt = self._upload("W", W).t().contiguous()
        G = self._gemm_nt(Wt, Wt, self._buf("gram", (self.H, self.H)), "gram_gemm")
        Y = res["Y"]
        A = self._buf("scores", (Y.shape[0], self.H))
        if Y.shape[0]:
            self._gemm_nt(Y, Wt, A, "scores_gemm")
        self._par = {"ykey": res["key"], "W": W.copy(), "Wt": Wt, "G": G, "A": A, "checked": tag,
                     "checked_obj": W if tag is not None else None}
        return self._par

    def _take_seed(self, W, res):
        """The seeded parameter record if ``W`` (D,H) is what the last M-step returned (compared with a private
        snapshot, so in-place edits by the caller are seen); the seed is consumed either way."""
        seed, self._seed_rec = getattr(self, "_seed_rec", None), None
        if seed is None or seed["W"] is None or seed["ykey"] != res["key"] or seed["W"].shape != W.shape \
                or not np.array_equal(seed["W"], W):
            return None
        return {"ykey": res["key"], "W": seed["W"], "Wt": seed["Wt"], "G": seed["G"], "A": seed["A"]}

    def _invert_normal_matrix(self, Wq_u, qdiag, status=None):
        """Wq = triu(Wq_u) + triu(Wq_u, 1)^T + diag(qdiag) and its inverse, enqueued on the CURRENT stream:
        ``(Wq, Winv, status)`` -- one-workgroup SPD inverse (csrc/spd_inverse.hip) instead of ~40 rocSOLVER launches.
        Only for device tensors with H <= 256.  ``status`` (3 doubles; ``status``: where to put them, e.g. a slice of the
        caller's download buffer) = [smallest pivot, largest pivot, accurate]: see ``_solve_normal_eq``.
        From the second call on the previous call's inverse warm-starts a Newton-Schulz refinement
        (pm_spd_inverse_warm_f64: ~40 us instead of the sweep's 0.3 ms when the matrix has moved little -- an EM loop;
        the device falls back to the sweep by itself otherwise and says so in status[2]).  ``PM_WARM_INVERSE=0`` disables it."""
        H = qdiag.shape[0]
        Wq = torch.empty((H, H), dtype=torch.float64, device=Wq_u.device)
        Winv = torch.empty((H, H), dtype=torch.float64, device=Wq_u.device)
        prev = getattr(self, "_winv_prev", None)
        warm = (prev is not None and tuple(prev.shape) == (H, H) and prev.device == Wq_u.device
                and os.environ.get("PM_WARM_INVERSE", "1") == "1")
        if warm:
            piv = status if status is not None else torch.empty(3, dtype=torch.float64, device=Wq_u.device)
            work = self._buf("spd_warm_work", (int(_lib.load().pm_spd_inverse_warm_work_len(H)),))
            # (pivots[2] <- the device's own verdict on the warm start: 1 / 2 = refinement accepted, 0 = the sweep ran)
            # The LONG form (scaled start, eight + one steps, ~60 us) where the start is likely to be far: the step after
            # one whose start was (``_warm_long``: _solve_accurate), and data-truncation steps, whose kept set can jump
            # (``_warm_force_long``, set by the model's M_step) -- the short form would hand those to the 0.3 ms sweep.
            # (Measured and dropped: the long form on truncation steps only while a far start is recent.  On the ramp the start
            # residual has norm ~0.08 in every direction: R_0^8 lands at 2-3e-8 in the Frobenius norm, just above the short
            # form's guard, and the step pays the sweep -- the schedule's mean went 2.43 -> 2.49 ms.)
            long = getattr(self, "_warm_long", False) or getattr(self, "_warm_force_long", False)
            self._call("spd_inverse", "pm_spd_inverse_warm_long_f64" if long else "pm_spd_inverse_warm_f64", _ptr(Wq_u), H,
                       _ptr(qdiag), H, _ptr(prev), H, _ptr(work), _ptr(Wq), _ptr(Winv), H, _ptr(piv), self._stream())
        else:
            if status is not None:
                piv = status
                piv[2:] = 1.0
            else:
                piv = torch.ones(3, dtype=torch.float64, device=Wq_u.device)
            self._call("spd_inverse", "pm_spd_inverse_f64", _ptr(Wq_u), H, _ptr(qdiag), H, _ptr(Wq), _ptr(Winv),
                       H, _ptr(piv), self._stream())
        self._winv_was_warm = warm
        self._winv_prev = Winv
        return Wq, Winv, piv

    def _one(self, device):
        one = getattr(self, "_one_dev", None)
        if one is None or one.device != device:
            one = self._one_dev = torch.ones(1, dtype=torch.float64, device=device)
        return one

    def _apply_inverse(self, Wq, Winv, rhs, refine=True, out=None):
        """X = Winv . rhs with one step of iterative refinement, X += Winv (rhs - Wq X) -- skipped (``refine=False``) behind
        the warm-started inverse, whose four Newton-Schulz steps already leave ||I - Wq Winv|| at rounding level.  Products
        run on the deterministic small-GEMM kernel (pm_gemm_nn_small_f64: fixed summation order, no zero fill, no
        K-slice atomics), so ranks that solve the same all-reduced system stay bitwise identical.  ``out``: where X goes."""
        H, D = rhs.shape
        st = self._stream()
        X = out if out is not None else torch.empty((H, D), dtype=torch.float64, device=rhs.device)
        nn = lambda A, B, C: self._call("solve_gemm", "pm_gemm_nn_small_f64", _ptr(A), A.stride(0), _ptr(B), B.stride(0),
                                        _ptr(C), C.stride(0), A.shape[0], B.shape[1], A.shape[1], st)
        if not refine:
            nn(Winv, rhs, X)                 # (Winv is symmetric)
            return X
        X0 = torch.empty((H, D), dtype=torch.float64, device=rhs.device)
        T = torch.empty((H, D), dtype=torch.float64, device=rhs.device)
        nn(Winv, rhs, X0)
        nn(Wq, X0, T)
        R = rhs - T
        nn(Winv, R, T)
        torch.add(X0, T, out=X)
        return X

    def _solve_normal_eq(self, Wq_u, qdiag, rhs, pre=None, out=None, status=None):
        """X = Wq^-1 . rhs, enqueued on the device, for the symmetric second-moment matrix
        Wq = triu(Wq_u) + triu(Wq_u, 1)^T + diag(qdiag) -- the models' ``np.linalg.lstsq(Wq, Wp)``
        (bsc_et.py:380, dsc_et.py:741).  Returns (X (H,D), status (3,) = [smallest, largest pivot of the elimination --
        smallest <= 0 marks a failed factorisation --, accurate], Wq (H,H)); the caller fetches the status with its one
        download, checks the pivots (``_solve_ok``: "singular" -> LAPACK's lstsq on the host) and hands status[2] to
        ``_solve_accurate``.  ``pre``: the result of ``_invert_normal_matrix`` when the caller has already run it;
        ``out`` / ``status``: device tensors to write X / the status into (slices of the caller's download buffer)."""
        H, D = rhs.shape
        self._last_solve = None
        if rhs.is_cuda and H <= 256:
            Wq, Winv, piv = pre if pre is not None else self._invert_normal_matrix(Wq_u, qdiag, status)
            warm = getattr(self, "_winv_was_warm", False)
            # (after a rejected warm start the next one is likely to be rejected too -- the first EM steps, a jump in the
            # annealing schedule --: refine right away then instead of finding out from the download and repeating the solve)
            refine = (not warm) or getattr(self, "_refine_next", False)
            X = self._apply_inverse(Wq, Winv, rhs, refine=refine, out=out)
            if not refine:
                # The refinement pass of the solve was skipped on the HOST's guess that the device would accept the warm
                # start.  The device's verdict rides behind the pivots; a caller that reads "sweep ran" there repeats the
                # solve with the refinement (_solve_accurate): the accuracy of W no longer depends on the call history.
                self._last_solve = (Wq, Winv, rhs)
            return X, piv, Wq
        if rhs.is_cuda:
            # H > 256: the one-workgroup inverse on 256-blocks + Schur complements (the library's own GEMMs; no rocSOLVER)
            Wq = torch.triu(Wq_u, 1)
            Wq = (Wq + Wq.t() + torch.diag(torch.diagonal(Wq_u) + qdiag)).contiguous()
            Winv, pmin, pmax = self._spd_inverse_blocked(Wq)
            piv = torch.stack([pmin, pmax, self._one(rhs.device)[0]])
            if status is not None:
                status.copy_(piv)
            return self._apply_inverse(Wq, Winv, rhs, out=out), piv, Wq
        # host tensors: the world_size-2 gloo tests feed CPU statistics through the same finalize code (never the
        # product path, whose statistics live on the device)
        Wq = torch.triu(Wq_u, 1)
        Wq = Wq + Wq.t() + torch.diag(torch.diagonal(Wq_u) + qdiag)
        Lc, info = torch.linalg.cholesky_ex(Wq)
        d = torch.diagonal(Lc) ** 2                       # squared Cholesky diagonal = the elimination's pivots
        X = torch.cholesky_solve(rhs, Lc).contiguous()    # garbage if the factorisation failed
        piv = torch.stack([torch.where(info.reshape(()) == 0, d.min(), -torch.ones((), dtype=d.dtype, device=d.device)),
                           d.max(), torch.ones((), dtype=d.dtype, device=d.device)])
        if out is not None:
            out.copy_(X)
            X = out
        if status is not None:
            status.copy_(piv)
        return X, piv, Wq

    def _solve_accurate(self, flag):
        """``None`` if the solution `_solve_normal_eq` returned is at full accuracy -- the refined cold solve, or the warm
        start the device accepted (``flag`` = status[2] from the download: 1) -- else (the device rejected the warm start and
        ran the sweep, but the host had skipped the refinement pass) the refined solution X (H,D) as a host array, computed
        now from the sweep's inverse: X0 + Winv (rhs - Wq X0), exactly what the cold path returns."""
        last, self._last_solve = getattr(self, "_last_solve", None), None
        self._refine_next = (flag == 0.0)
        self._warm_long = (flag != 1.0)       # (a far start this time: the scaled long form next time)
        if flag != 0.0 or last is None:
            return None
        Wq, Winv, rhs = last
        self._redo_dev = self._apply_inverse(Wq, Winv, rhs, refine=True)      # (kept: callers seed the next step from it)
        return self._redo_dev.cpu().numpy()

    def _spd_inverse_blocked(self, A):
        """Inverse of a symmetric positive definite device matrix of any size from pm_spd_inverse_f64 (n <= 256, one
        workgroup) by recursive 2 x 2 blocking:  with T = A11^-1 A12 and the Schur complement S = A22 - A12^T T,
            A^-1 = [[A11^-1 + T S^-1 T^T, -T S^-1], [-S^-1 T^T, S^-1]].
        Returns (inverse, smallest pivot, largest pivot) -- the pivots of the blocks ARE the pivots of the unblocked
        elimination (first those of A11, then those of S).  Products go through pm_gemm_tn_acc_f64 / pm_gemm_nt_f64."""
        n = A.shape[0]
        dev = A.device
        st = self._stream()
        if n <= 256:
            A = A.contiguous()
            inv = torch.empty((n, n), dtype=torch.float64, device=dev)
            piv = torch.empty(2, dtype=torch.float64, device=dev)
            self._call("spd_inverse", "pm_spd_inverse_f64", _ptr(A), n, None, n, None, _ptr(inv), n, _ptr(piv), st)
            return inv, piv[0], piv[1]
        n1 = min(256, n // 2 // 16 * 16) if n <= 512 else (n // 2 + 15) // 16 * 16
        n2 = n - n1
        A11, A12, A22 = A[:n1, :n1].contiguous(), A[:n1, n1:].contiguous(), A[n1:, n1:].contiguous()
        I11, p1min, p1max = self._spd_inverse_blocked(A11)
        T = torch.zeros((n1, n2), dtype=torch.float64, device=dev)
        self._call("solve_gemm", "pm_gemm_tn_acc_f64", _ptr(I11), n1, _ptr(A12), n2, _ptr(T), n2, n1, n2, n1, st)   # I11^T A12
        C = torch.zeros((n2, n2), dtype=torch.float64, device=dev)
        self._call("solve_gemm", "pm_gemm_tn_acc_f64", _ptr(A12), n2, _ptr(T), n2, _ptr(C), n2, n2, n2, n1, st)     # A12^T T
        S = A22 - C
        S = (0.5 * (S + S.t())).contiguous()
        IS, p2min, p2max = self._spd_inverse_blocked(S)
        B12 = torch.empty((n1, n2), dtype=torch.float64, device=dev)
        self._gemm_nt(T, IS, B12, "solve_gemm")                  # T S^-1 (S^-1 symmetric)
        TST = torch.empty((n1, n1), dtype=torch.float64, device=dev)
        self._gemm_nt(B12, T, TST, "solve_gemm")                 # T S^-1 T^T
        out = torch.empty((n, n), dtype=torch.float64, device=dev)
        B11 = I11 + TST
        out[:n1, :n1] = 0.5 * (B11 + B11.t())
        out[:n1, n1:] = -B12
        out[n1:, :n1] = -B12.t()
        out[n1:, n1:] = IS
        return out, torch.minimum(p1min, p2min), torch.maximum(p1max, p2max)

    @staticmethod
    def _solve_ok(piv_min, piv_max):
        """The pivots of ``_solve_normal_eq`` describe a usable solution (positive definite, ratio above 1e-11)."""
        ratio = piv_min / piv_max if piv_max != 0 else 0.0
        return piv_min > 0 and np.isfinite(ratio) and ratio > 1e-11

    def _device_candidates(self, cand, N):
        if isinstance(cand, DeviceArray):
            t = cand.tensor
        else:
            t = torch.from_numpy(np.ascontiguousarray(cand, dtype=np.int32)).to(self.device)
        if t.dtype != torch.int32:
            t = t.to(torch.int32)
        assert tuple(t.shape) == (N, self.Hprime)
        return t.contiguous()

    KTH_ROUNDS = ((52, 12), (40, 12), (28, 12), (16, 12), (4, 12), (0, 4))     # (shift, bits) of the radix digits

    def _kth_largest_global(self, lse, N_use):
        """sort(all log-evidences over all ranks)[-N_use] (bsc_et.py:252 via parallel.allsort) by distributed radix
        select (csrc/kth_select.hip): per round one histogram pass over the local shard, ONE all-reduce of 4096 bins
        (32 KB, instead of gathering 8 N bytes to every rank and sorting them there), a one-workgroup scan; six rounds
        decide all 64 bits of the order statistic, so the value is exactly the one a full sort returns.  Only the
        final double reaches the host.  (CPU tensors -- the gloo tests -- walk the same protocol with torch ops.)"""
        comm = self.comm
        lse = lse.contiguous()
        dev = lse.device
        n = int(lse.shape[0])
        if lse.is_cuda:
            return float(self._kth_select_dev(lse, N_use).item())
        state = torch.zeros(2, dtype=torch.int64, device=dev)
        state[1] = int(N_use)
        hist = torch.zeros(4096, dtype=torch.int64, device=dev)
        # host tensors (tests): the same rounds on the same 64-bit keys
        b = lse.view(torch.int64)
        key = torch.where(b < 0, ~b, b | torch.iinfo(torch.int64).min)          # order-preserving as UNSIGNED 64-bit
        ukey = key.numpy().view(np.uint64)
        prefix, k = np.uint64(0), int(N_use)
        for shift, bits in self.KTH_ROUNDS:
            top = shift + bits
            match = np.ones(n, dtype=bool) if top >= 64 else ((ukey >> np.uint64(top)) == (prefix >> np.uint64(top)))
            digit = ((ukey[match] >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
            hist = torch.from_numpy(np.bincount(digit, minlength=4096).astype(np.int64))
            comm.allreduce_device(hist)
            h = hist.numpy()
            above = 0
            for d in range((1 << bits) - 1, -1, -1):
                if above + int(h[d]) >= k:
                    break
                above += int(h[d])
            prefix |= np.uint64(d) << np.uint64(shift)
            k -= above
        bits64 = (prefix & np.uint64(0x7FFFFFFFFFFFFFFF)) if (prefix >> np.uint64(63)) else ~prefix
        return float(np.array([bits64], dtype=np.uint64).view(np.float64)[0])

    def _kth_select_dev(self, lse, N_use):
        """The radix select of ``_kth_largest_global`` on a device tensor, result LEFT ON THE DEVICE (one double): the
        deferred statistics of a data-truncation step (pm_bsc_defer_apply_f64) read the cut from there, so the step has no
        host round trip between its E-step pass and its M-step kernels."""
        comm = self.comm
        lse = lse.contiguous()
        dev = lse.device
        n = int(lse.shape[0])
        # one persistent buffer [states: 7 x (prefix, k) | pad | 6 histograms of 4096 bins], zeroed when it is allocated and
        # left zeroed by every select's final kernel; the rank travels as an argument of round 0: 6 + 1 launches, no fills
        ws = self.__dict__.get("_kth_buf")
        if ws is None or ws[0].device != dev or ws[1]:
            ws = self._kth_buf = [torch.zeros(16 + 6 * 4096, dtype=torch.int64, device=dev), False]
        ws[1] = True                   # (dirty until the final kernel is enqueued: an exception in between re-zeroes)
        buf = ws[0]
        states, hists = buf[:14], buf[16:].view(6, 4096)
        st = self._stream()
        prev = (0, 1)
        for r, (shift, bits) in enumerate(self.KTH_ROUNDS):
            if r and comm.size > 1:
                comm.allreduce_device(hists[r - 1])
            self._call("kth_round", "pm_kth_round_k_f64", _ptr(lse) if n else None, n, _ptr(states), _ptr(hists), r, prev[0],
                       prev[1], shift, bits, int(N_use) if r == 0 else -1, st)
            prev = (shift, bits)
        if comm.size > 1:
            comm.allreduce_device(hists[len(self.KTH_ROUNDS) - 1])
        out = torch.empty(1, dtype=torch.float64, device=dev)
        self._call("kth_final", "pm_kth_final_z_f64", _ptr(states), _ptr(hists), len(self.KTH_ROUNDS), prev[0], prev[1],
                   _ptr(out), 1, st)
        ws[1] = False
        return out

    # ------------------------------------------------------------------ inference ("next" row)
    def inference(self, anneal, model_params, test_data, topK=10, logprob=False, adaptive=True,
                  Hprime_max=None, gamma_max=None):
        """Top-K posterior states and marginals per datapoint (camodels/__init__.py:256-375).

        Same return dict as the reference: ``s`` (N,topK,H) int8, ``m`` (N,H), ``p`` (N,topK),
        ``gamma`` / ``Hprime`` (N,).  With ``adaptive`` the datapoints whose MAP state has exactly
        gamma active units are re-run with Hprime+1 / gamma+1 (the state table is regenerated) until
        none remain or the caps are reached.  The log-joints come from the HIP E-step; the
        normalisation, top-K and marginals run on the device too."""
        from . import generate_state_matrix
        assert 'y' in test_data, "Key 'y' in test_data dict not defined."
        model_params = self.check_params(model_params)
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
        res_p = torch.zeros((my_N, topK), dtype=torch.float64, device=dev)
        res_gamma = torch.zeros((my_N,), dtype=torch.float64, device=dev)
        res_Hprime = torch.zeros((my_N,), dtype=torch.float64, device=dev)

        cur_y = my_y
        which = torch.ones(my_N, dtype=torch.bool, device=dev)
        try:
            while bool(which.any()):
                ind_n = torch.nonzero(which).flatten()
                logpj, cand = self.compute_lpj(anneal, model_params, {'y': cur_y})
                lp = logpj.tensor if isinstance(logpj, DeviceArray) else torch.as_tensor(np.asarray(logpj)).to(dev)
                cd = cand.tensor if isinstance(cand, DeviceArray) else torch.as_tensor(np.asarray(cand)).to(dev)
                cd = cd.long()
                n_cur, K = lp.shape
                Hp = self.Hprime
                k_eff = min(topK, K)
                # top-K columns of the normalised posterior and the log-marginals: one HIP pass over the rows
                # (csrc/infer_kernels.hip).  Upstream quirk (:309-312): logprob=False reports exp(logpj - max), NOT the
                # normalised value -- the kernel returns both
                SMh = self.state_matrix.astype(np.int64)
                mk = (SMh << np.arange(SMh.shape[1])[None, :]).sum(axis=1).astype(np.uint16) if SMh.size else np.zeros(1, np.uint16)
                masks_d = torch.from_numpy(mk.view(np.int16).copy()).to(dev)
                lp = lp.contiguous() if lp.stride(1) != 1 else lp
                cd32 = cd.to(torch.int32).contiguous()
                top_idx32 = torch.empty((n_cur, k_eff), dtype=torch.int32, device=dev)
                top_val = torch.empty((n_cur, k_eff), dtype=torch.float64, device=dev)
                top_rel = torch.empty((n_cur, k_eff), dtype=torch.float64, device=dev)
                m_blk = torch.empty((n_cur, H), dtype=torch.float64, device=dev)
                self._call("infer_topk", "pm_infer_topk_f64", _ptr(lp), lp.stride(0), _ptr(cd32), _ptr(masks_d), n_cur, H, Hp,
                           self.no_states, k_eff, _ptr(top_idx32), _ptr(top_val), _ptr(top_rel), _ptr(m_blk), H, self._stream())
                if bool((top_idx32 < 0).any()):
                    # fewer than topK comparable log-joints in some row (NaN: a non-finite datapoint or parameter).  The
                    # reference's argsort would rank the NaNs somewhere and carry on; indexing with -1 here would silently
                    # report a wrapped state -- refuse instead
                    raise _lib.HipError("inference: non-finite log-joints (NaN) in %d datapoint(s)"
                                        % int((top_idx32 < 0).any(dim=1).sum()))
                top_idx = top_idx32.long()
                res_Hprime[ind_n] = float(self.Hprime)
                res_gamma[ind_n] = float(self.gamma)
                # top-K states as H-dimensional binary vectors
                SM = torch.from_numpy(self.state_matrix.astype(np.int8)).to(dev) if self.no_states else \
                    torch.zeros((1, Hp), dtype=torch.int8, device=dev)
                # upstream quirk (:313-319): a re-run datapoint's earlier entries are never cleared -- one-cause
                # states only SET their bit, the null state writes nothing, multi-cause states overwrite the
                # candidate positions; start from what is there
                s_blk = res_s[ind_n, :k_eff].clone()
                single = (top_idx >= 1) & (top_idx <= H)
                if bool(single.any()):
                    nn_, mm_ = torch.nonzero(single, as_tuple=True)
                    s_blk[nn_, mm_, top_idx[nn_, mm_] - 1] = 1
                multi = top_idx > H
                if bool(multi.any()):
                    nn_, mm_ = torch.nonzero(multi, as_tuple=True)
                    rows = SM[top_idx[nn_, mm_] - H - 1]                           # (M, Hp)
                    s_blk[nn_[:, None].expand(-1, Hp), mm_[:, None].expand(-1, Hp), cd[nn_]] = rows
                res_s[ind_n, :k_eff] = s_blk
                res_p[ind_n, :k_eff] = top_val if logprob else torch.exp(top_rel)
                res_m[ind_n] = m_blk        # log p(s_h = 1 | y): log-sum-exp over the states containing h (the kernel)
                if not adaptive:
                    break
                which = ((res_s[:, 0, :] != 0).sum(-1) == self.gamma)
                if not bool(which.any()):
                    break
                if (Hprime_max is not None and self.Hprime == Hprime_max) and \
                        (gamma_max is not None and self.gamma == gamma_max):
                    break
                cur_y = my_y[which.cpu().numpy()] if not torch.is_tensor(my_y) else my_y[which]
                n_left = int(which.sum())
                print("Rank %i: For %i data points MAP state has activity equal to gamma." % (comm.rank, n_left))
                if not ((self.Hprime == self.H) or (Hprime_max is not None and self.Hprime == Hprime_max)):
                    self.Hprime += 1
                if (self.gamma == self.H) or (gamma_max is not None and self.gamma == gamma_max):
                    continue
                self.gamma += 1
                print("Rank %i: Updating state matrix and running again." % comm.rank)
                (self.state_list, self.no_states, self.state_matrix,
                 self.state_abs) = generate_state_matrix(self.Hprime, self.gamma)
        finally:
            comm.Barrier()
            self.Hprime, self.gamma = Hprime_start, gamma_start
            (self.state_list, self.no_states, self.state_matrix,
             self.state_abs) = generate_state_matrix(self.Hprime, self.gamma)
        m_out = res_m if logprob else torch.exp(res_m)
        return {'s': res_s.cpu().numpy(), 'm': m_out.cpu().numpy(), 'p': res_p.cpu().numpy(),
                'gamma': res_gamma.cpu().numpy(), 'Hprime': res_Hprime.cpu().numpy()}
