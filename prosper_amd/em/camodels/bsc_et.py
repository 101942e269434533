"""Binary Sparse Coding on the MI355X: drop-in for prosper/em/camodels/bsc_et.py.

Same class name, constructor, ``select_Hprimes / E_step / M_step`` signatures, dict keys,
``dlog`` side effects and return layouts as the reference (``BSC_ET``, bsc_et.py:24-438);
the three hot methods enqueue hand-written HIP kernels through the C ABI of
libprosper_hip.so (include/prosper_hip.h) instead of looping over datapoints in NumPy:

  select_Hprimes  scores A = Y.W^T and Gram G = W.W^T (f64 MFMA GEMM), wavefront top-H'
  E_step          energies of all 1+H+S truncated states from A and G, log-sum-exp
  M_step          posterior weights, E[s], Wq scatter, sigma/pi/L statistics, then
                  Wp = E[s]^T.Y (f64 MFMA GEMM), ONE all-reduce of the packed statistics
                  (RCCL over xGMI; replaces the MPI calls at bsc_et.py:225-417), and the
                  H x H solve on the device.

Data stays resident in HBM between EM steps; ``candidates`` and ``logpj`` are returned as
``DeviceArray`` handles (NumPy-convertible, downloaded only when a caller looks at them).
There is no CPU fallback for the kernels: without the HIP library / a GPU this raises.
"""
import ctypes
import os
from math import pi as _PI

import numpy as np
from scipy.special import comb

from ._device import (DeviceCAModel, DeviceArray, LazyCandidates, KernelTimer, _ptr, small_blas)  # noqa: F401
from ... import _lib
from ...utils import parallel
from ...utils import tracing
from ...utils.datalog import dlog

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class BSC_ET(DeviceCAModel):
    """Binary Sparse Coding with Expectation Truncation; HIP kernels behind the
    reference's plugin surface.  ``device`` defaults to the current CUDA/HIP device."""

    def __init__(self, D, H, Hprime, gamma, to_learn=['W', 'pi', 'sigma'], comm=parallel.COMM_WORLD,
                 device=None):
        DeviceCAModel.__init__(self, D, H, Hprime, gamma, to_learn, comm, device)
        self._tables = None      # device copies of the state table
        self.use_rows16 = True   # 16-lanes-per-datapoint kernels when the shape allows (tests flip this)
        self._side = None        # side stream of the chunked GEMM / row-kernel pipeline
        self._a0 = None          # (par, data key, rows): whose first-chunk scores the scores_c0 buffer holds
        self._spec_ok = False    # the last M-step's seeded parameters were used as they were
        self.speculate = os.environ.get('PM_SPECULATE', '1') == '1'
        self._in_step = False
        self._spec_estep = None  # next step's E-step, launched by the M-step (_speculate_estep)
        self._next_anneal = None  # the NEXT step's annealing point as predicted in step() (_predict_anneal), or None
        self._flat_schedule = False
        self.spec_hits = 0
        self.speculate_estep = os.environ.get('PM_SPECULATE_ESTEP', '1') == '1'
        # plain attributes (the tests flip them to compare code paths; no environment switches):
        self.fuse_mstats = True      # M-step row statistics inside the fused E-step pass
        self.defer_stats = True      # data-truncation steps: the E-step pass leaves per-datapoint records, added after the cut
                                     # is known (pm_bsc_defer_apply_f64) -- no second pass over the log-joints
        self.sparse_wp = True        # Wp = E[s]^T Y from the non-zero lists of E[s] (pm_bsc_wp_sparse_f64); the dense product
                                     # still runs -- decided on the device -- when a list overflowed
        self.use_fused = True        # scores GEMM + select + E-step as ONE kernel (bsc_fused8.hip / bsc_fused.hip)
        self.overlap_streams = False  # two-kernel path: GEMM(c+1) beside the row kernel of chunk c
        self.chunk_rounds = 0        # two-kernel path: GEMM rounds per chunk (0 = whole shard)
        self.max_chunk_rows = 1 << 20

    # ------------------------------------------------------------------ plumbing
    def _state_tables(self):
        """Upload the truncated state table: 16-bit masks + CSR lists of the multi-cause
        states containing each pair of candidate positions (for E[s s^T])."""
        key = (self.Hprime, self.gamma, self.no_states, self.use_rows16)
        if self._tables is not None and self._tables["key"] == key:
            return self._tables
        _lib.load()
        Hp = self.Hprime
        SM = self.state_matrix.astype(np.int64)                      # (S, Hp)
        masks = (SM << np.arange(Hp)[None, :]).sum(axis=1).astype(np.uint16) if SM.size else np.zeros(0, np.uint16)
        ptr = [0]
        lst = []
        for i in range(Hp):
            for j in range(Hp):
                if SM.size:
                    lst.extend(np.where((SM[:, i] == 1) & (SM[:, j] == 1))[0].tolist())
                ptr.append(len(lst))
        # fast path tables: parent state (state minus its highest candidate position) and the
        # first state index of every size 2..gamma (states are ordered by size)
        mlist = masks.tolist()
        index_of = {m: i for i, m in enumerate(mlist)}
        parents = np.full(len(mlist), 0xFFFF, dtype=np.uint16)
        for i, m in enumerate(mlist):
            rest = m & ~(1 << (m.bit_length() - 1))
            if bin(rest).count("1") >= 2:
                parents[i] = index_of[rest]
        sizes = np.array([bin(m).count("1") for m in mlist], dtype=np.int64)
        size_off = [int(np.searchsorted(sizes, g, side="left")) for g in range(2, self.gamma + 1)] + [len(mlist)]
        size_off = (size_off + [len(mlist)])[:max(self.gamma, 1)]
        dev = self.device
        self._tables = {
            "key": key,
            "parents": torch.from_numpy(parents.view(np.int16).copy()).to(dev)
            if parents.size else torch.zeros(1, dtype=torch.int16, device=dev),
            "size_off": (ctypes.c_int32 * len(size_off))(*size_off),
            "fast": bool(_lib.load().pm_bsc_rows16_supported(self.H, Hp, len(mlist))) and self.use_rows16,
            # (the list-writing M-step pass needs 16 H more doubles of LDS: H = 256, H' = 16 does not fit)
            "fast_nz": bool(_lib.load().pm_bsc_rows16_nz_supported(self.H, Hp, len(mlist))) and self.use_rows16,
            # uint16 payloads travel as int16 tensors (same bytes)
            "masks": torch.from_numpy(np.ascontiguousarray(masks).view(np.int16).copy()).to(dev)
            if masks.size else torch.zeros(1, dtype=torch.int16, device=dev),
            "pair_ptr": torch.tensor(ptr, dtype=torch.int32, device=dev),
            "pair_states": torch.tensor(lst if lst else [0], dtype=torch.int16, device=dev),
            "pair_len": len(lst),
        }
        return self._tables

    def _same_W(self, par, W):
        """Is ``W`` (D,H) the matrix ``par`` was built from?  Compared against a private host snapshot
        (a pinned staging / download buffer), so in-place edits by the caller are seen."""
        Wh = par.get("Whost")
        if Wh is None or W.shape != (self.D, self.H):
            return False
        return np.array_equal(Wh, W.T if par["Whost_T"] else W)

    def _params_dev(self, W, res):
        """Device copy of W^T (H,D) and the Gram matrix G = W.W^T for the current W.  The M-step leaves
        its solution on the device (``_seed_params``), so inside an EM loop nothing is uploaded."""
        par = self._par
        if par.get("ykey") == res["key"] and ((self._in_step and par.get("checked") is W) or self._same_W(par, W)):
            par["checked"] = W       # (inside step() nothing can touch W between select_Hprimes and E_step)
            if par.pop("seeded", False):
                self._spec_ok = True         # the EM loop fed the M-step's W straight back: keep speculating
            return par
        self._spec_ok = False
        W = np.asarray(W, dtype=np.float64)
        if W.T.flags.c_contiguous:           # (D,H) view of an (H,D) array, e.g. what M_step returns
            Wt, Whost, flag = self._upload("W", W.T, keep=True) + (True,)
        else:
            Wd, Whost = self._upload("W", W, keep=True)
            Wt, flag = Wd.t().contiguous(), False                                  # transposed on device
        G = self._gemm_nt(Wt, Wt, self._buf("gram", (self.H, self.H)), "gram_gemm")
        self._par = {"ykey": res["key"], "Whost": Whost, "Whost_T": flag, "Wt": Wt, "G": G, "A": None}
        return self._par

    def _seed_params(self, res, Wt_dev, G, speculate):
        """After an M-step the next step's W^T and Gram matrix are already on the device.  Returns the
        parameter record to install once the host snapshot ``Whost`` has arrived.  ``speculate``: the
        previous seed was consumed unchanged (plain EM loop, no parameter noise), so the next step's
        scores GEMM is enqueued right away and runs while the host finishes this step."""
        par = {"ykey": res["key"], "Whost": None, "Whost_T": True, "Wt": Wt_dev, "G": G, "A": None, "seeded": True}
        if speculate:
            self._prefetch_scores(res, par)
        return par

    def install_parameters(self, data, Wt_dev, Wt_host):
        """Install a W^T (H,D) that already lives on the device -- the state an M-step leaves behind --
        for the shard ``data['y']``: the Gram matrix is recomputed and scores of earlier parameters are
        dropped.  ``Wt_host`` is the caller's host copy of the same matrix (checked against the W of later calls: it is
        kept by reference, so it must not share memory with an array the caller goes on to edit)."""
        res = self._resident(data['y'])
        G = self._gemm_nt(Wt_dev, Wt_dev, self._buf("gram", (self.H, self.H)), "gram_gemm")
        par = self._seed_params(res, Wt_dev, G, False)
        par["Whost"] = Wt_host
        par["seeded"] = False
        self._par, self._a0 = par, None

    def _scores(self, model_params, res):
        """A = Y.W^T (N,H) and G = W.W^T for the current W; reused by E_step when
        select_Hprimes just computed them for the same W and data."""
        par = self._params_dev(np.asarray(model_params['W']), res)
        if par["A"] is None:
            Y = res["Y"]
            N = Y.shape[0]
            A = self._buf("scores", (N, self.H))
            if N:
                self._gemm_nt(Y, par["Wt"], A, "scores_gemm")
            par["A"] = A
        return par

    # ---- scores GEMM + select + E-step in one kernel ---------------------------------------------
    def _fused(self):
        """The one-kernel E-step applies (H <= 256 and the row passes' LDS areas fit); the K dimension is zero-padded
        to a multiple of 8 where D is not one."""
        if not (self.use_fused and self._state_tables()["fast"]):
            return False
        D8 = (self.D + 7) // 8 * 8
        return bool(_lib.load().pm_bsc_fused_supported(self.H, D8, self.Hprime, self.no_states))

    def _k8(self, holder, name, t):
        """``t`` (rows, D) with its rows zero-padded to a multiple of 8 columns (cached in ``holder``)."""
        D8 = (self.D + 7) // 8 * 8
        if D8 == self.D and t.stride(0) % 2 == 0 and t.data_ptr() % 16 == 0:
            return t
        p = holder.get(name)
        if p is None or p[0] is not t:
            buf = torch.zeros((t.shape[0], D8), dtype=torch.float64, device=self.device)
            buf[:, :self.D] = t
            p = holder[name] = (t, buf)
        return p[1]

    def _tile8_whole_shard(self):
        """The 8-wavefront tile with its lean passes applies (config 2's shape class): one C call per shard."""
        lib = _lib.load()
        D8 = (self.D + 7) // 8 * 8
        return bool(lib.pm_bsc_fused8_supported(self.H, D8, self.Hprime, self.no_states)
                    and lib.pm_bsc_fused8_whole_shard(self.H, self.Hprime, self.gamma, self.no_states))

    def _dominant_rows(self, N):
        """Datapoints the launch labelled ``estep_fused`` covers (bench.py's roofline accounting)."""
        if self._tile8_whole_shard():
            D8 = (self.D + 7) // 8 * 8
            return int(_lib.load().pm_bsc_fused8_main_rows(N, D8)) or N
        return self._fused_rows(N)

    def _fused_rows(self, N):
        """Rows of the shard the one-kernel E-step takes: whole rounds of resident workgroups (2 per CU, 64 datapoints
        each).  A ragged last round would run a fraction of the chip for a whole tile time (tiles cannot be split over
        K: the row passes need complete scores), so those rows go through the two-kernel path, whose GEMM does split K
        -- unless they fill most of a round anyway, or the shard is smaller than one round."""
        if self._tile8_whole_shard():
            return N          # the 8-wavefront kernels take the whole shard (a TAIL launch for the ragged round)
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        rnd = 2 * cus * 64
        if N < rnd:
            return N
        main = N // rnd * rnd
        return N if (N - main) * 10 >= rnd * 7 else main

    def _fused_estep(self, res, par, mode, cand, P, wmu, ymu, logpj, lse, mstats=None, defer=False):
        """``mstats`` = (expect (N,H), stats): also produce the per-datapoint M-step statistics of the rows the fused
        kernel takes (E_step inside ``step``).  ``defer`` (a data-truncation step ahead): as per-datapoint records that
        ``M_step`` adds once the cut is known, instead of accumulating them."""
        tab = self._state_tables()
        Y8, W8 = self._k8(res, "Y8", res["Y"]), self._k8(par, "Wt8", par["Wt"])
        N, H, Hp, S = Y8.shape[0], self.H, self.Hprime, self.no_states
        ldl = logpj.stride(0) if logpj is not None else 0
        main = self._fused_rows(N)
        Pref = ctypes.byref(P) if P is not None else None
        cur = torch.cuda.current_stream(self.device)
        if main < N:      # the ragged last round: split-K scores GEMM + row kernel on those rows, ahead of the big launch
            # (on a side stream beside it they steal slots from whole rounds: 2.01 vs 1.89 ms per pass)
            st = ctypes.c_void_p(cur.cuda_stream)
            off = lambda t, w=1: ctypes.c_void_p(t.data_ptr() + main * w * t.element_size()) if t is not None else None
            A = self._rest_scores(res, par, main, st)
            self._call("select_estep_rest", "pm_bsc_select_estep_f64", _ptr(A), H, _ptr(par["G"]), off(res["ynorm2"]),
                       _ptr(wmu), off(ymu), _ptr(tab["masks"]), _ptr(tab["parents"]), tab["size_off"], S, self.gamma,
                       Pref, N - main, H, Hp, mode, off(cand, Hp), off(logpj, ldl), ldl, off(lse), st)
        D8 = Y8.shape[1]
        # the 8-wavefront tile (bsc_fused8.hip) where it applies; its M-statistics ride on the lean passes (H' = 8)
        tile8 = self._tile8_whole_shard()
        args = (_ptr(Y8), Y8.stride(0), _ptr(W8), W8.stride(0), _ptr(par["G"]),
                _ptr(res["ynorm2"]), _ptr(wmu), _ptr(ymu), _ptr(tab["masks"]), _ptr(tab["parents"]), tab["size_off"],
                S, self.gamma, Pref, main, Y8.shape[1], H, Hp, mode, _ptr(cand), _ptr(logpj), ldl, _ptr(lse),
                _ptr(mstats[0]) if mstats else None, H, _ptr(mstats[1]) if mstats else None, self.D)
        if tile8:
            # two launches, labelled apart (timers / traces): whole rounds of 64-row tiles, then the ragged remainder
            main_rows = int(_lib.load().pm_bsc_fused8_main_rows(main, Y8.shape[1]))
            entry = "pm_bsc_estep_fused8_f64"
            self._nz = None
            self._qd_done = bool(mstats)                                  # these passes leave qdiag complete (= mus)
            if mstats and self.sparse_wp and main == N:
                nz_max = 16                                            # PM_BSC_NZ_MAX
                nz = (self._buf("nz_idx", (N, nz_max), torch.int16), self._buf("nz_val", (N, nz_max)))
                args = args + (_ptr(nz[0]), _ptr(nz[1]))
                entry = "pm_bsc_estep_fused8_nz_f64"
                # ("listed": the pass stores the dense E[s] row of a datapoint only when its list overflowed)
                self._nz = {"idx": nz[0], "val": nz[1], "stats": mstats[1], "rows": N, "dense_rows": "overflowed"}
                if defer:
                    rec = self._buf("defer_rec", (N, 40))              # PM_BSC_DEFER_LD
                    args = args + (_ptr(rec),)
                    entry = "pm_bsc_estep_fused8_defer_f64"
                    self._nz["defer"] = rec
            if main_rows > 0:
                self._call("estep_fused", entry, *(args + (1, self._stream())))
            if main_rows < main:
                self._call("estep_fused_tail", entry, *(args + (2, self._stream())))
            return main
        self._call("estep_fused", "pm_bsc_estep_fused_f64", *(args + (self._stream(),)))
        return main

    # ---- scores GEMM, then the fused, chunked select + E-step (two-kernel path) -------------------
    def _round_rows(self):
        """Rows of one round of resident 128x128 GEMM tiles (2 workgroups per CU)."""
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        tiles_n = (self.H + 127) // 128
        return max(1, (2 * cus) // tiles_n) * 128

    def _chunk_rows(self, N=None):
        """Rows per pipeline chunk: the largest whole number of GEMM rounds that fits the shard (so
        the main GEMM launch has no ragged last round; the remainder is a second, split-K launch),
        capped at ``max_chunk_rows`` to bound the score buffer."""
        rr = self._round_rows()
        if self.chunk_rounds > 0:
            return rr * self.chunk_rounds
        N = self.max_chunk_rows if N is None else min(N, self.max_chunk_rows)
        return max(rr, N // rr * rr)

    def _whole_shard(self, N):
        """One scores buffer and one row-kernel launch for the shard (the default up to ``max_chunk_rows``)."""
        return self.chunk_rounds <= 0 and not self.overlap_streams and N <= self.max_chunk_rows + self._round_rows()

    def _ensure_scores(self, res, par):
        """Whole-shard mode: A = Y.W^T for every row of the shard in one (N, H) buffer.  The rows that fill whole
        rounds of resident GEMM tiles go out as one launch, the ragged remainder as a split-K launch."""
        Y = res["Y"]
        N, H = Y.shape[0], self.H
        A = self._buf("scores_all", (N, H))
        tag = (par, res["key"], "all", N)
        if self._a0 is not None and self._a0[0] is par and self._a0[1:] == tag[1:]:
            return A
        stream = torch.cuda.current_stream(self.device)
        # one call: pm_gemm_nt_f64 runs whole rounds of tiles and K-slices of the ragged last round in ONE launch
        self._scores_chunk(res, par, A, 0, N, N, 1, stream)
        self._a0 = tag
        return A

    def _launch_rows(self, N):
        """Datapoints one ``scores_gemm`` launch covers (bench.py's roofline accounting)."""
        return N if self._whole_shard(N) else min(N, self._chunk_rows(N))

    def _run_select_estep(self, res, par, mode, cand, P=None, wmu=None, ymu=None, logpj=None, lse=None, mstats=None,
                          defer=False):
        """Scores GEMM + fused select/E-step kernel.  Whole-shard mode: the scores of all rows, then ONE pass of
        the row kernel.  Chunked mode (``chunk_rounds`` / shards beyond ``max_chunk_rows``): a chunk is a whole
        number of rounds of resident GEMM tiles; its (chunk, H) score block is consumed by the row kernel
        straight away, so the scores buffer is bounded (two alternating buffers).  Measured: chunk size barely
        matters (one round per chunk 2.27 ms, whole shard 2.15 ms per pass at config 2).
        ``overlap``: GEMM of chunk c+1 on a side stream while the row kernel of chunk c runs."""
        if self._fused():
            return self._fused_estep(res, par, mode, cand, P, wmu, ymu, logpj, lse, mstats, defer)
        Y = res["Y"]
        N, H, Hp, S = Y.shape[0], self.H, self.Hprime, self.no_states
        tab = self._state_tables()
        ldl = logpj.stride(0) if logpj is not None else 0
        main = torch.cuda.current_stream(self.device)

        def rows_kernel(A, r0, r1):
            off = lambda t, w=1: ctypes.c_void_p(t.data_ptr() + r0 * w * t.element_size()) if t is not None else None
            self._call("select_estep", "pm_bsc_select_estep_f64", _ptr(A), H, _ptr(par["G"]), off(res["ynorm2"]),
                       _ptr(wmu), off(ymu), _ptr(tab["masks"]), _ptr(tab["parents"]), tab["size_off"], S,
                       self.gamma, ctypes.byref(P) if P is not None else None, r1 - r0, H, Hp, mode,
                       off(cand, Hp), off(logpj, ldl), ldl, off(lse), ctypes.c_void_p(main.cuda_stream))

        if self._whole_shard(N):
            rows_kernel(self._ensure_scores(res, par), 0, N)
            return

        rows = self._chunk_rows(N)
        nchunks = (N + rows - 1) // rows
        bufs = [self._buf("scores_c0", (rows, H)), self._buf("scores_c1", (rows, H))]
        overlap = self.overlap_streams and nchunks > 1
        if overlap:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            side = self._side
            side.wait_stream(main)
        else:
            side = main
        done = [None, None]
        self._a0 = None
        for c in range(nchunks):
            r0, r1 = c * rows, min(N, (c + 1) * rows)
            A = bufs[c & 1]
            with torch.cuda.stream(side):
                if overlap and done[c & 1] is not None:
                    side.wait_event(done[c & 1])     # the row kernel that read this buffer is finished
                self._scores_chunk(res, par, A, r0, r1, rows, nchunks, side)
                if overlap:
                    ready = torch.cuda.Event()
                    ready.record(side)
            if overlap:
                main.wait_event(ready)
            rows_kernel(A, r0, r1)
            if overlap:
                done[c & 1] = torch.cuda.Event()
                done[c & 1].record(main)

    def _scores_chunk(self, res, par, A, r0, r1, rows, nchunks, stream):
        Y = res["Y"]
        self._call("scores_gemm" if (r1 - r0 == rows or nchunks == 1) else "scores_gemm_tail", "pm_gemm_nt_f64",
                   _ptr(Y[r0:r1]), Y.stride(0), _ptr(par["Wt"]), par["Wt"].stride(0), _ptr(A), self.H, r1 - r0, self.H,
                   Y.shape[1], ctypes.c_void_p(stream.cuda_stream))

    def _prefetch_scores(self, res, par):
        """Enqueue the scores GEMMs as soon as W is known (select_Hprimes, or speculatively at the end of an
        M-step), so the device is busy while the host walks on to E_step."""
        N = res["Y"].shape[0]
        if N and self._fused():
            self._rest_scores(res, par, self._fused_rows(N), self._stream())     # the ragged last round's split-K GEMM
        elif N and self._whole_shard(N):
            self._ensure_scores(res, par)

    def _rest_scores(self, res, par, main, st):
        """Scores of the rows behind the fused kernel's whole rounds (needs W only: enqueued as soon as W is known,
        speculatively behind an M-step's download); cached on the parameter record."""
        N, H = res["Y"].shape[0], self.H
        if main >= N:
            return None
        got = par.get("A_rest")
        if got is not None and got[0] == main and got[2] == res["key"]:
            return got[1]
        A = self._buf("scores_rest", (N - main, H))
        Yr = res["Y"][main:]
        self._call("scores_gemm_rest", "pm_gemm_nt_f64", _ptr(Yr), Yr.stride(0), _ptr(par["Wt"]), par["Wt"].stride(0),
                   _ptr(A), H, N - main, H, Yr.shape[1], st)
        par["A_rest"] = (main, A, res["key"])
        return A

    def _materialize_candidates(self, ticket):
        """Selection on its own (someone looked at the lazy candidates before E_step ran)."""
        res, par = ticket["res"], ticket["par"]
        N = res["Y"].shape[0]
        cand = torch.empty((N, self.Hprime), dtype=torch.int32, device=self.device)
        if N:
            self._run_select_estep(res, par, 1, cand)
        ticket["cand"] = cand

    # ------------------------------------------------------------------ data generation
    @tracing.traced
    def generate_from_hidden(self, model_params, my_hdata):
        """y = s.W^T + N(0, sigma^2) (bsc_et.py:67-95; one normal((my_N, D)) draw).
        Does not obey gamma."""
        W = model_params['W'].T
        s = my_hdata['s']
        my_N = s.shape[0]
        y = np.asarray(s, dtype=np.float64) @ W
        y += np.random.normal(scale=model_params['sigma'], size=(my_N, W.shape[1]))
        return {'y': y, 's': s}

    # ------------------------------------------------------------------ hot path
    def step(self, anneal, model_params, my_data):
        """CAModel.step (camodels/__init__.py:163-193); E_step knows that M_step follows with the same arguments."""
        self._in_step = True
        self._par.pop("checked", None)        # the first look at W in every step is a full comparison (_same_W)
        self._next_anneal = self._predict_anneal(anneal)
        try:
            return DeviceCAModel.step(self, anneal, model_params, my_data)
        finally:
            self._in_step = False

    @tracing.traced
    def select_Hprimes(self, model_params, data):
        """Annotate ``data`` with ``data['candidates']`` (N, Hprime): per datapoint the
        Hprime latents with the largest <W_h,y>/|W_h|/|y|, ascending (bsc_et.py:98-115)."""
        res = self._resident(data['y'])
        N = res["Y"].shape[0]
        if self._state_tables()["fast"]:
            # deferred: E_step fuses selection with the log-joints in one pass (LazyCandidates)
            par = self._params_dev(np.asarray(model_params['W']), res)
            self._prefetch_scores(res, par)
            ticket = {"res": res, "par": par, "cand": None}
            data['candidates'] = LazyCandidates(self, ticket, (N, self.Hprime))
            return data
        par = self._scores(model_params, res)
        cand = torch.empty((N, self.Hprime), dtype=torch.int32, device=self.device)
        if N:
            G = par["G"]
            self._call("select", "pm_bsc_select_f64", _ptr(par["A"]), self.H, _ptr(G), self.H + 1, _ptr(res["ynorm2"]),
                      N, self.H, self.Hprime, _ptr(cand), self._stream())
        data['candidates'] = DeviceArray(cand, np.int64)
        return data

    def _estep_params(self, anneal, pies, sigma, mu):
        beta = 1. / anneal['T']
        pre1 = -1. / 2. / sigma / sigma
        return _lib.EStepParams(pil_bar=float(np.log(pies / (1. - pies))), ecoef=float(beta * pre1),
                                prior_scale=float(beta if anneal['anneal_prior'] else 1.0),
                                mu_sqnorm=float(np.dot(mu, mu)))

    def _mu_terms(self, par, res, mu):
        """W.mu and Y.mu when the data offset is non-zero (bsc_et.py:169 uses y - mu)."""
        if not np.any(mu):
            return None, None
        mu_d = torch.from_numpy(np.ascontiguousarray(mu, dtype=np.float64)).to(self.device).reshape(1, -1)
        wmu = self._gemm_nt(par["Wt"], mu_d, self._buf("wmu", (self.H, 1)))
        N = res["Y"].shape[0]
        ymu = self._buf("ymu", (N, 1))
        if N:
            self._gemm_nt(res["Y"], mu_d, ymu)
        return wmu, ymu

    def _estep_outputs(self, N):
        """Fresh result tensors (handed to the caller; the caching allocator recycles last step's).  Rows padded to whole
        128-byte lines: unaligned rows made every row store straddle two lines (WRITE_SIZE 1.8x the bytes); the caller
        sees the (N, K) view."""
        K = 1 + self.H + self.no_states
        Kpad = (K + 15) // 16 * 16
        logpj = torch.empty((N, Kpad), dtype=torch.float64, device=self.device)[:, :K]
        return logpj, torch.empty((N,), dtype=torch.float64, device=self.device)

    def _launch_estep(self, res, par, P, cand, wmu, ymu, want_ms):
        """Enqueue the fast-path E-step for parameter record ``par``: selection too when ``cand`` is None.  Returns the
        DeviceArray of log-joints with ``.lse``, ``.cand`` and (``want_ms``, fused kernel only) ``.mstats``."""
        N, D = res["Y"].shape
        H, Hp = self.H, self.Hprime
        logpj, lse = self._estep_outputs(N)
        mode = 2
        if cand is None:      # selection + log-joints in one pass
            cand = torch.empty((N, Hp), dtype=torch.int32, device=self.device)
            mode = 3
        defer = (want_ms == "defer")
        if defer and not (self.defer_stats and self.sparse_wp and N and self._fused() and self._tile8_whole_shard()
                          and self._fused_rows(N) == N):
            want_ms = False           # (no deferred form on this path: the M-step runs its own pass behind the cut)
        mstats, rows = None, 0
        # (deterministic mode: the quanta of the statistics derive from W's column norms -- a pass whose W^T exists only on the
        # device so far, the M-step's speculative launch, carries no statistics; the M-step's own pass forms them then)
        Wh = par.get("Whost")
        if N and want_ms and self.fuse_mstats and self._fused() and Hp <= 8 and 'mu' not in self.to_learn \
                and (not self.deterministic or Wh is not None or defer):
            if self.deterministic and not defer:
                self._det_quanta(res, Wh.T if par["Whost_T"] else Wh, np.full(1, np.sqrt(max(P.mu_sqnorm, 0.0))), P, N)
            n_stats = _lib.load().pm_bsc_stats_len(H, D)
            # two statistics workspaces alternate: the M-step that launches the NEXT E-step (speculation) still reads its
            # own Wp -- the right-hand side of the W solve, and of the host fallback -- from the other one
            self._stats_flip = 1 - getattr(self, "_stats_flip", 0)
            stats = self._buf("stats%d" % self._stats_flip, (n_stats,))
            stats.zero_()
            mstats = (self._buf("expect", (N, H)), stats)
        self._nz = None
        self._qd_done = False
        if N:
            rows = self._run_select_estep(res, par, mode, cand, P, wmu, ymu, logpj, lse, mstats, defer and mstats is not None)
        out = DeviceArray(logpj)
        out.lse = lse
        out.cand = cand
        if mstats is not None:
            out.mstats = {"expect": mstats[0], "stats": mstats[1], "rows": rows, "res": res, "cand": cand,
                          "P": (P.pil_bar, P.ecoef, P.prior_scale, P.mu_sqnorm), "nz": self._nz, "qd_done": self._qd_done}
        self._nz = None
        return out

    def _want_ms(self, anneal):
        """Which M-step statistics the E-step pass inside ``step`` carries: True -- accumulated in the pass (no data
        truncation ahead); "defer" -- left as per-datapoint records that M_step adds once the cut is known
        (``Ncut_factor > 0``: 49 of the 50 steps of the reference's schedules, bars-learning.py:77-80); False -- none."""
        if anneal['Ncut_factor'] <= 0.0 and not (self.deterministic and self.defer_stats and self.sparse_wp
                                                 and self._fused() and self._tile8_whole_shard()):
            return True
        # (deterministic mode takes the deferred form on EVERY step: the records need no quanta while the pass runs -- the
        # M-step's speculative launch of the next pass knows W^T on the device only, not the bounds the quanta derive from --
        # and the apply kernel runs when M_step has them; until round 6 every deterministic step paid the M-step's own
        # pass over the log-joints instead: + 0.4 ms at config 2)
        return "defer" if self.defer_stats else False

    def _speculate_estep(self, res, par, anneal, pies, sigma):
        """Called by the M-step as soon as its scalar statistics are on the host, while the device still solves for
        W: the NEXT step's E-step for (W_new on the device, pi_new, sigma_new, the same annealing point), enqueued
        behind the solve.  The device then walks from one EM step into the next without waiting for the host's
        turnaround (download, bookkeeping, the next step's Python); E_step adopts the pass iff it is called with exactly
        these parameters, else the pass is dropped (wasted device time, never a wrong result)."""
        if not (np.isfinite(pies) and np.isfinite(sigma) and 0.0 < pies < 1.0 and sigma > 0.0):
            return
        P = self._estep_params(anneal, pies, sigma, np.zeros(1))
        want_ms = self._want_ms(anneal)
        out = self._launch_estep(res, par, P, None, None, None, want_ms)
        self._spec_estep = {"par": par, "res": res, "want_ms": want_ms, "out": out,
                            "P": (P.pil_bar, P.ecoef, P.prior_scale, P.mu_sqnorm)}

    @tracing.traced
    def E_step(self, anneal, model_params, my_data):
        """Log-pseudo-joints of the truncated state set -> ``{'logpj': (N, 1+H+S)}``
        (bsc_et.py:119-192).  Inserts ``model_params['mu']`` when absent, as upstream."""
        res = self._resident(my_data['y'])
        N, D = res["Y"].shape
        H, Hp, S = self.H, self.Hprime, self.no_states
        try:
            mu = model_params['mu']
        except KeyError:
            mu = np.zeros(D)
            model_params['mu'] = mu
        tab = self._state_tables()
        W = np.asarray(model_params['W'])
        mu64 = np.asarray(mu, dtype=np.float64)
        P = self._estep_params(anneal, model_params['pi'], model_params['sigma'], mu64)
        tracing.tracepoint("E_step:iterating")
        cobj = my_data['candidates']
        if tab["fast"]:
            par = self._params_dev(W, res)
            fuse = (isinstance(cobj, LazyCandidates) and cobj.pending and cobj._model is self
                    and cobj._ticket["res"] is res and cobj._ticket["par"] is par)
            # inside CAModel.step with no data truncation ahead the M-step's per-datapoint statistics are produced by
            # the same pass (fused kernel): posterior weights from the exponentials the log-sum-exp evaluates anyway, no
            # second pass over the 665 MB of log-joints
            want_ms = self._in_step and self._want_ms(anneal)
            sp, self._spec_estep = self._spec_estep, None
            if (sp is not None and fuse and sp["par"] is par and sp["res"] is res and sp["want_ms"] == want_ms
                    and sp["P"] == (P.pil_bar, P.ecoef, P.prior_scale, P.mu_sqnorm) and not np.any(mu64)):
                out = sp["out"]          # the previous M-step has already launched exactly this pass (_speculate_estep)
                self.spec_hits += 1
            else:
                wmu, ymu = self._mu_terms(par, res, mu64)
                cand = None if fuse else self._device_candidates(cobj, N)
                out = self._launch_estep(res, par, P, cand, wmu, ymu, want_ms)
            if fuse:
                cobj._ticket["cand"] = out.cand
            return {'logpj': out}
        par = self._scores(model_params, res)
        cand = self._device_candidates(cobj, N)
        wmu, ymu = self._mu_terms(par, res, mu64)
        logpj, lse = self._estep_outputs(N)
        if N:
            self._call("estep", "pm_bsc_estep_f64", _ptr(par["A"]), H, _ptr(par["G"]), _ptr(res["ynorm2"]),
                      _ptr(wmu), _ptr(ymu), _ptr(cand), _ptr(tab["masks"]), S, ctypes.byref(P),
                      N, H, Hp, _ptr(logpj), logpj.stride(0), _ptr(lse), self._stream())
        out = DeviceArray(logpj)
        out.lse = lse
        return {'logpj': out}

    @tracing.traced
    def M_step(self, anneal, model_params, my_suff_stat, my_data):
        """New W, pi, sigma (, mu) from the posterior over the truncated states
        (bsc_et.py:195-438).  Logs ``N``, ``L`` (free energy) and ``N_use`` to dlog."""
        comm = self.comm
        H, Hp, D, gamma, S = self.H, self.Hprime, self.D, self.gamma, self.no_states
        W_DH = np.asarray(model_params['W'])
        pies = model_params['pi']
        sigma = model_params['sigma']
        mu = np.asarray(model_params['mu'], dtype=np.float64)

        if anneal['data_noise'] > 0:
            # upstream adds my_data['data_noise'] to y for the M-step only (bsc_et.py:228-230), a key nothing in
            # the reference ever sets (KeyError there); not carried over -- refuse instead of ignoring the schedule
            raise NotImplementedError("anneal['data_noise'] > 0 is not supported by the HIP M-step "
                                      "(reference bsc_et.py:228-230 reads an undefined my_data['data_noise'])")
        res = self._resident(my_data['y'])
        Y = res["Y"]
        my_N = Y.shape[0]
        tab = self._state_tables()
        cand = self._device_candidates(my_data['candidates'], my_N)
        K = 1 + H + S

        logpj = my_suff_stat['logpj']
        if isinstance(logpj, DeviceArray) and logpj.lse is not None:
            lp, lse = logpj.tensor, logpj.lse
        else:   # log-joints handed in from outside: upload, recompute the log-evidence
            lp = torch.from_numpy(np.ascontiguousarray(np.asarray(logpj), dtype=np.float64)).to(self.device)
            lse = torch.logsumexp(lp, dim=1)
        if lp.dim() != 2 or lp.stride(1) != 1:
            lp = lp.contiguous()
        ldl = lp.stride(0) if my_N else K
        lse = lse.contiguous()
        assert tuple(lp.shape) == (my_N, K)

        N = self._global_count(res, my_N)

        # factors of the pi update (bsc_et.py:237-244)
        A_pi_gamma = 0
        B_pi_gamma = 0
        for gamma_p in range(gamma + 1):
            t = comb(H, gamma_p) * (pies ** gamma_p) * ((1 - pies) ** (H - gamma_p))
            A_pi_gamma += t
            B_pi_gamma += gamma_p * t
        E_pi_gamma = pies * H * A_pi_gamma / B_pi_gamma

        # data truncation (bsc_et.py:247-258): keep the N_use datapoints with the largest evidence
        ncut = anneal['Ncut_factor'] > 0.0
        N_use = (int(N * (1 - (1 - A_pi_gamma) * anneal['Ncut_factor'])) or N) if ncut else N
        # (int(...) == 0 on a handful of datapoints: upstream's allsort(...)[-0] is the SMALLEST value -- everything is kept,
        # bsc_et.py:251-253)

        # per-datapoint statistics + Wp GEMM into the packed buffer
        tracing.tracepoint("M_step:iterating")
        _lib.load()
        n_stats = _lib.load().pm_bsc_stats_len(H, D)
        expect = self._buf("expect", (my_N, H))
        P = self._estep_params(anneal, pies, sigma, mu)
        st = self._stream()
        if self.deterministic and my_N:
            self._det_quanta(res, np.asarray(model_params['W'], dtype=np.float64), mu, P, N)
        # statistics the E-step pass has already produced for its first `done` rows (same shard, candidates, scalars)
        ms = getattr(logpj, "mstats", None) if isinstance(logpj, DeviceArray) else None
        if ms is not None and any(ms["stats"] is self._ws.get("stats%d" % k) for k in (0, 1)):
            stats = ms["stats"]                       # (the workspace that pass accumulated into: see _launch_estep)
        else:
            self._stats_flip = 1 - getattr(self, "_stats_flip", 0)
            # (with 'mu' learned the D column sums of the kept datapoints ride behind the statistics: one buffer, one all-reduce)
            stats = self._buf("stats%d" % self._stats_flip, (n_stats + (D if 'mu' in self.to_learn else 0),))[:n_stats]
        self._ws["stats"] = stats                     # (the workspace of THIS M-step, for tools and tests)
        done, nz = 0, None
        mine = (ms is not None and ms["res"] is res and ms["stats"] is stats and ms["expect"] is expect and ms["cand"] is cand
                and ms["P"] == (P.pil_bar, P.ecoef, P.prior_scale, P.mu_sqnorm) and tab["fast"])
        rec = ms["nz"].get("defer") if (mine and ms.get("nz")) else None
        lse_cut = float("-inf")
        if ncut and rec is None:
            tracing.tracepoint("M_step:truncating")
            lse_cut = self._kth_largest_global(lse, N_use)
            if lse_cut < -745.1332191019412:     # log(2^-1075): the reference's un-stabilised evidence sums are exactly
                lse_cut = float("-inf")          # 0 there, and `all_denoms >= 0` keeps every datapoint (bsc_et.py:253)
        if mine and rec is not None:
            # The E-step pass has left every datapoint's statistics as a record beside its non-zero list (a data-truncation
            # step: _want_ms): the cut is selected ON THE DEVICE and stays there, pm_bsc_defer_apply_f64 adds the records of
            # the datapoints above it and empties the lists of the others -- no host round trip, no second pass over the
            # 665 MB of log-joints (round 6; pm_bsc_mstep_rows16_nz_f64 before: 0.42 ms + a device idle for the cut).
            # (Measured and dropped: the kept datapoints' half of the apply kernel on a second stream beside the sparse
            # product -- 2.36-2.41 ms per step either way, scratch/em_ncut_wall.py: what one gains the other loses.)
            tracing.tracepoint("M_step:truncating")
            if ncut:
                cut_dev = self._kth_select_dev(lse, N_use)
            else:
                cut_dev = torch.full((1,), float("-inf"), dtype=torch.float64, device=lse.device)
            done = ms["rows"]
            nz = ms["nz"]
            logpj.mstats = None
            assert done == my_N and ms.get("qd_done")
            self._call("defer_apply", "pm_bsc_defer_apply_f64", _ptr(lse), _ptr(cut_dev), _ptr(cand), _ptr(rec),
                       _ptr(nz["idx"]), _ptr(nz["val"]), _ptr(expect), H, _ptr(stats), ctypes.byref(P), my_N, H, D, Hp, st)
        elif mine and lse_cut == float("-inf"):
            done = ms["rows"]
            nz = ms.get("nz")
            logpj.mstats = None
            lib = _lib.load()
            o_wq, o_qd, o_mus = (lib.pm_bsc_stats_offset_wq(H, D), lib.pm_bsc_stats_offset_qdiag(H, D),
                                 lib.pm_bsc_stats_offset_mus(H, D))
            # the 4-wavefront pass leaves mus = sum E[s] and the multi-cause block of Wq incl. its diagonal; qdiag (the
            # singletons' share of the diagonal, s_h^2 = s_h) is the difference.  The whole-shard passes leave the full
            # diagonal in qdiag themselves (one small launch less per step).
            if not ms.get("qd_done"):
                stats[o_qd:o_mus] = stats[o_mus:o_mus + H] - torch.diagonal(stats[o_wq:o_qd].view(H, H))
        else:
            stats.zero_()
        if done:
            if done < my_N:
                r = my_N - done
                off = lambda t, w=1: ctypes.c_void_p(t.data_ptr() + done * w * t.element_size())
                self._call("mstep_rows", "pm_bsc_mstep_rows16_f64", off(lp, ldl), ldl, off(lse), ctypes.c_double(lse_cut),
                           off(cand, Hp), _ptr(tab["masks"]), S, ctypes.byref(P), r, H, D, Hp, off(expect, H), H,
                           _ptr(stats), st)
        elif my_N and tab["fast_nz"] and self.sparse_wp and expect.is_cuda:
            # (the M-step's own pass -- after a data-truncation step, or on log-joints from outside: lists as well)
            nzb = (self._buf("nz_idx", (my_N, 16), torch.int16), self._buf("nz_val", (my_N, 16)))
            self._call("mstep_rows", "pm_bsc_mstep_rows16_nz_f64", _ptr(lp), ldl, _ptr(lse), ctypes.c_double(lse_cut),
                       _ptr(cand), _ptr(tab["masks"]), S, ctypes.byref(P), my_N, H, D, Hp, _ptr(expect), H,
                       _ptr(stats), _ptr(nzb[0]), _ptr(nzb[1]), st)
            done = my_N
            nz = {"idx": nzb[0], "val": nzb[1], "stats": stats, "rows": my_N}
        elif my_N and tab["fast"]:
            self._call("mstep_rows", "pm_bsc_mstep_rows16_f64", _ptr(lp), ldl, _ptr(lse), ctypes.c_double(lse_cut),
                       _ptr(cand), _ptr(tab["masks"]), S, ctypes.byref(P), my_N, H, D, Hp, _ptr(expect), H,
                       _ptr(stats), st)
        elif my_N:
            self._call("mstep_rows", "pm_bsc_mstep_rows_f64", _ptr(lp), ldl, _ptr(lse), ctypes.c_double(lse_cut), _ptr(cand),
                      _ptr(tab["masks"]), S, _ptr(tab["pair_ptr"]), _ptr(tab["pair_states"]), tab["pair_len"],
                      ctypes.byref(P), my_N, H, D, Hp, _ptr(expect), H, _ptr(stats), st)
        # (Everything but Wp is final here, and the H x H inverse needs only the second moments -- but putting it on a
        # high-priority side stream ahead of the statistics GEMM gains nothing: its 1024-thread workgroup needs a whole
        # CU's registers, the GEMM's workgroups refill every slot as it frees, so the inverse still starts when the GEMM
        # has drained; 4.88 vs 4.87 ms per EM iteration.  Round 4, with the warm start's small launches beside the SPARSE
        # product on one rank: 2.27 against 2.21 ms -- slower; the chain stays in stream order.)
        if my_N and done == my_N and nz is not None and nz["stats"] is stats and nz["rows"] == my_N:
            # the lists of this very pass: sparse product, and the dense one behind the device-side gate (scalars[3])
            gate = ctypes.c_void_p(stats.data_ptr() + 8 * (_lib.load().pm_bsc_stats_offset_scalars(H, D) + 3))
            self._expect_rows = nz.get("dense_rows", "all")      # (for tools / tests: which rows of _ws["expect"] the pass stored)
            if nz.get("dense_rows") == "overflowed":
                # (the E-step pass stored the dense rows of the overflowed datapoints only: if the gate is set, this
                # launch fills in the listed ones for the dense product instead of returning at once)
                self._call("stats_sparse", "pm_bsc_wp_sparse_expand_f64", _ptr(nz["idx"]), _ptr(nz["val"]), _ptr(Y),
                           Y.stride(0), _ptr(stats), _ptr(expect), H, my_N, H, D, st)
            else:
                self._call("stats_sparse", "pm_bsc_wp_sparse_f64", _ptr(nz["idx"]), _ptr(nz["val"]), _ptr(Y), Y.stride(0),
                           _ptr(stats), my_N, H, D, st)
            self._call("stats_gemm", "pm_gemm_tn_acc_gated_f64", _ptr(expect), H, _ptr(Y), D, _ptr(stats), D, H, D,
                       my_N, gate, st)
        elif my_N:
            self._expect_rows = "all"
            self._call("stats_gemm", "pm_gemm_tn_acc_f64", _ptr(expect), H, _ptr(Y), D, _ptr(stats), D, H, D, my_N, st)
        need_mu = 'mu' in self.to_learn
        if need_mu:       # my_data_sum over the kept datapoints (bsc_et.py:422-430): one pass over the shard
            packed = self._ws["stats%d" % self._stats_flip]
            assert packed.numel() == n_stats + D and packed.data_ptr() == stats.data_ptr()
            packed[n_stats:].zero_()
            if my_N:
                self._call("data_sum", "pm_col_sum_kept_f64", _ptr(Y), Y.stride(0), my_N, D, _ptr(lse),
                           ctypes.c_double(lse_cut), ctypes.c_void_p(packed.data_ptr() + 8 * n_stats), st)
        else:
            packed = stats

        self._warm_force_long = ncut      # (the kept set of a truncation step can jump: the inverse's scaled warm start)
        # the exchange of the step (replaces bsc_et.py:225,258,266,373,374,387,417,426,427)
        comm.allreduce_device(packed)
        return self._finalize(packed, model_params, A_pi_gamma, E_pi_gamma, res, None, anneal)

    def _det_quanta(self, res, W_DH, mu, P, N):
        """Deterministic mode: bounds no partial sum of the M-step's statistics can exceed on this shard with these
        parameters -> the quanta of the kernels about to run (pm_common.h, PM_Q).  Energies: |y - mu - sum_{h in s} W_h|^2
        <= (|y| + |mu| + gamma max|W_h|)^2; log-joints: prior + |ecoef| energy; sums of probabilities: N."""
        ymax, ynmax = self._det_data_bounds(res)
        wn = float(np.sqrt((W_DH * W_DH).sum(axis=0)).max()) if W_DH.size else 0.0
        mun, mumax = float(np.linalg.norm(mu)), float(np.abs(mu).max()) if np.size(mu) else 0.0
        emax = (ynmax + mun + self.gamma * wn) ** 2
        K = 1 + self.H + self.no_states
        lpmax = abs(P.prior_scale * P.pil_bar) * self.gamma + abs(P.ecoef) * emax + np.log(K)
        n = float(max(N, res["Y"].shape[0]))
        cats = [n, n * emax, n * lpmax]
        for unit in ("bsc_rows16", "bsc_kernels", "bsc_fused", "bsc_fused8"):
            self._det_set(unit, cats)
        self._det_set("wp_sparse", [n * (ymax + mumax)])
        self._det_set("gemm", [n * (ymax + mumax), n * (ymax + mumax)])

    def _scalar_updates(self, host, pies, sigma, E_pi_gamma):
        """pi and sigma updates (bsc_et.py:393-420) from the head of the downloaded statistics, ``host`` =
        [mus (H) | sum of squared residuals, sum of free energies, N_use, -]."""
        H, D = self.H, self.D
        mus_h = host[:H].copy()
        my_sigma, Fs, N_use = float(host[H]), float(host[H + 1]), int(round(host[H + 2]))
        pi_new = E_pi_gamma * float(mus_h.sum()) / H / N_use if 'pi' in self.to_learn else pies
        sigma_new = np.sqrt(my_sigma / D / N_use) if 'sigma' in self.to_learn else sigma
        return mus_h, my_sigma, Fs, N_use, pi_new, sigma_new

    def _finalize(self, packed, model_params, A_pi_gamma, E_pi_gamma, res=None, pre=None, anneal=None):
        """Parameter updates from the all-reduced statistics (bsc_et.py:264-267, 369-438).
        Everything is enqueued on the device first (Wq assembly, Cholesky solve, reductions) and
        fetched with ONE device->host copy, so an EM step synchronises with the GPU exactly once.
        Device-agnostic: ``packed`` may live in HBM (product path) or on the host (the
        world_size-2 gloo tests feed it CPU tensors)."""
        H, D = self.H, self.D
        W_DH = np.asarray(model_params['W'])
        pies, sigma = model_params['pi'], model_params['sigma']
        mu = np.asarray(model_params['mu'], dtype=np.float64)
        lib = _lib.load()
        n_stats = lib.pm_bsc_stats_len(H, D)
        o_wq, o_qd = lib.pm_bsc_stats_offset_wq(H, D), lib.pm_bsc_stats_offset_qdiag(H, D)
        o_mus, o_sc = lib.pm_bsc_stats_offset_mus(H, D), lib.pm_bsc_stats_offset_scalars(H, D)
        Wp = packed[:o_wq].view(H, D)
        Wq_u = packed[o_wq:o_qd].view(H, H)
        qdiag = packed[o_qd:o_mus]
        mus = packed[o_mus:o_sc]
        learn_W, learn_mu = 'W' in self.to_learn, 'mu' in self.to_learn

        head = packed[o_mus:o_sc + 4]             # [mus (H) | 4 scalars], contiguous in the packed buffer; summed on the host
        Wq = rhs = seed = None
        # A plain EM loop whose next annealing point is known (_predict_anneal: a flat schedule, or a LinearAnnealing that
        # moved as predicted last step): the scalar statistics travel to the host AHEAD of the W solve, so pi_new / sigma_new
        # are known while the device still inverts Wq, and the next step's E-step is enqueued right behind the solve
        # (_speculate_estep) -- the device never waits for the host between two EM steps.
        early = None
        if (packed.is_cuda and res is not None and anneal is not None and learn_W and not learn_mu and not np.any(mu)
                and self._in_step and self._next_anneal is not None and self._spec_ok and self.speculate
                and self.speculate_estep
                and self._state_tables()["fast"] and self._fused()):
            early = self._download_async(head, slot="mstep_early")
        # The download buffer [mus | scalars | status (3) + pad | X (H,D) | data sums (D)]: the inverse writes its status and
        # the solve its solution straight into it (no torch.cat of 2 MB), on the device.  A fresh tensor every step (the caching
        # allocator recycles the memory once nothing refers to it): parameter records of earlier steps -- a LazyCandidates
        # ticket someone still holds -- keep pointing at THEIR W^T.
        n_head = H + 4
        o_st, o_x = n_head, n_head + 4
        n_flat = o_x + (H * D if learn_W else 0) + (D if learn_mu else 0)
        flat = torch.empty(n_flat, dtype=torch.float64, device=packed.device)
        if early is None:
            flat[:n_head] = head
        if learn_W:
            tracing.tracepoint("M_step:update W")
            rhs = Wp
            if np.any(mu):   # Wp was accumulated against y, the reference uses y - mu
                rhs = Wp - torch.outer(mus, torch.from_numpy(mu).to(packed.device))
            rhs = rhs.contiguous()
            # (under speculation the next E-step accumulates into the OTHER statistics workspace: Wp stays intact for the
            # host-lstsq fallback and the repeated solve below)
            X, status, Wq = self._solve_normal_eq(Wq_u, qdiag, rhs, pre, out=flat[o_x:o_x + H * D].view(H, D),
                                                  status=flat[o_st:o_st + 3])
            if pre is not None:
                flat[o_st:o_st + 3] = status
            if packed.is_cuda and res is not None:   # next step's W^T is already here (its Gram matrix: `then` below)
                seed = X
        if learn_mu:
            flat[n_flat - D:] = packed[n_stats:]
        if flat.is_cuda:                                        # the one synchronisation of the EM step
            spec = []
            if learn_W:
                self._par = {}                                  # its host snapshot may live in the buffer reused now
                self._a0 = None
            def then():
                # (The download is on its way -- on the copy stream, a blit kernel of 2 MB that needs CUs: it must get going
                # BEFORE the next E-step's 1024-thread workgroups fill the chip, or it crawls between them for 0.3 ms and
                # costs that launch its whole-round tiling.  The Gram matrix and the workspace fill enqueued from here give
                # it that head start.)
                G = self._gemm_nt(seed, seed, torch.empty((H, H), dtype=torch.float64, device=seed.device), "gram_gemm")
                spec.append(self._seed_params(res, seed, G, self._spec_ok and self.speculate))
                if early is not None:
                    early[1].synchronize()
                    _, _, _, _, pi_e, sigma_e = self._scalar_updates(early[0], pies, sigma, E_pi_gamma)
                    self._speculate_estep(res, spec[0], self._next_anneal, pi_e, sigma_e)
            # (under speculation the head has already travelled: `body` then starts behind it)
            body = self._download(flat if early is None else flat[n_head:], slot="mstep", then=then if seed is not None else None)
            base = 0 if early is None else n_head
            host = body if early is None else early[0]
        else:
            host = body = flat.numpy()
            base = 0
        o_st, o_x = o_st - base, o_x - base                     # offsets into `body`

        mus_h, my_sigma, Fs, N_use, pi_new, sigma_new = self._scalar_updates(host, pies, sigma, E_pi_gamma)
        dlog.append('N', N_use)
        L = H * np.log(1 - pies) - 0.5 * D * np.log(2 * _PI * sigma ** 2) - np.log(A_pi_gamma)
        L += Fs / N_use
        dlog.append('L', L)

        pos = o_x
        if learn_W:
            ok = self._solve_ok(float(body[o_st]), float(body[o_st + 1]))
            redo = self._solve_accurate(float(body[o_st + 2])) if ok else None
            if redo is not None:
                # the device rejected the warm start of the inverse and the solve behind it had skipped its refinement
                # pass: W from the refined solve; whatever was seeded / speculated from the unrefined one is void
                self._a0 = None
                W_new = redo
                if seed is not None:      # the next step still finds its W^T and Gram matrix on the device
                    Xr = self._redo_dev
                    G = self._gemm_nt(Xr, Xr, torch.empty((H, H), dtype=torch.float64, device=Xr.device), "gram_gemm")
                    par = self._seed_params(res, Xr, G, False)
                    # (a PRIVATE snapshot: the caller gets a view of W_new and may edit it in place -- found by
                    # tests/test_call_sequences_gpu.py: the shared array hid such an edit from _same_W)
                    par["Whost"] = W_new.copy()
                    self._par = par
                self._spec_estep = None
            elif ok:
                Wt_host = body[o_x:o_x + H * D].reshape(H, D)
                W_new = Wt_host.copy()
                if seed is not None:
                    spec[0]["Whost"] = Wt_host
                    self._par = spec[0]
            else:   # numerically singular Wq: the reference's own LAPACK lstsq on the host
                self._a0 = None
                self._winv_prev = None        # never warm-start the next inverse from a rejected one
                with small_blas():
                    W_new = np.linalg.lstsq(Wq.cpu().numpy(), rhs.cpu().numpy(), rcond=None)[0]
            pos += H * D
        else:
            W_new = W_DH.T

        if 'pi' in self.to_learn:
            tracing.tracepoint("M_step:update pi")
        if 'sigma' in self.to_learn:
            tracing.tracepoint("M_step:update sigma")

        if learn_mu:
            tracing.tracepoint("M_step:update mu")
            # the reference divides by the rank-local kept count (bsc_et.py:428), which is only
            # meaningful on one rank; with several ranks the global count is used
            dsum = body[pos:pos + D]
            mu_new = dsum / N_use - np.inner(W_new.T / N_use, mus_h)
        else:
            mu_new = mu

        dlog.append('N_use', N_use)
        return {'W': W_new.T, 'pi': pi_new, 'sigma': sigma_new, 'mu': mu_new}
