/*
 * prosper_hip.h -- C ABI of libprosper_hip.so: the MI355X (gfx950) implementation of
 * prosper's truncated-EM hot path (select_Hprimes -> E_step -> M_step of the
 * component-analysis models in prosper/em/camodels/).
 *
 * The reference is pure Python/NumPy and has no FFI; each entry point below names the
 * reference code it replaces (file:line relative to the reference root).  A maintainer
 * binds these with ctypes from the model's select_Hprimes / E_step / M_step methods
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (e.g. torch.Tensor.data_ptr()) unless its name ends
 *     in _host; the caller owns the memory and keeps it alive until the stream has drained
 *   - matrices are dense row-major; "ld" arguments are row strides in elements
 *   - dims are int64_t; `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - calls only enqueue work: results are valid after the stream is synchronised
 *   - return value: 0 = ok, <0 = PM_E* bad argument, >0 = hipError_t of a failed launch
 *   - no hidden allocation, no global mutable state, no host<->device copies
 *   - float64 throughout: the reference computes in IEEE double
 */
#ifndef PROSPER_HIP_H
#define PROSPER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_EINVAL (-1)      /* null pointer / non-positive dimension            */
#define PM_ERANGE (-2)      /* dimension outside the supported range (see below) */

#define PM_MAX_H 1024       /* latents per datapoint row handled by one wavefront */
#define PM_MAX_HPRIME 16    /* candidates; state masks are 16-bit                 */

/* ABI version (major*1000 + minor); bumped when a signature changes. */
int pm_version(void);

/* ---------------------------------------------------------------------------------------
 * Deterministic reductions (libprosper_hip_det.so: the same sources built with -DPM_DETERMINISTIC)
 * ---------------------------------------------------------------------------------------
 * The M-step statistics are sums over datapoints formed with f64 atomics; the order the addends land in changes from run to
 * run and with it the last bits of the sums (the reference at a fixed number of ranks is deterministic).  In the deterministic
 * build every addend is rounded to a multiple of a quantum q_c = 2^-52 M_c before it is added, M_c a power of two no partial
 * sum of its category can exceed: all additions are then exact and the result is independent of their order -- the same
 * bits in every run.  pm_det_build(): 1 in that build, 0 in the default one (whose kernels contain none of this).
 * pm_det_set_quanta(unit, M8, stream): the eight magic constants 1.5 M_c of one kernel family (host OR device memory; copied before the
 * family's next launch on `stream`; PM_ERANGE-like -2 in the default build).  Categories per family:
 *   PM_DET_BSC_FUSED8 / PM_DET_DSC   0 Wq, mus, counts (sums of probabilities)   1 sum of q e   2 sum of log-evidences
 *   PM_DET_WP_SPARSE                 0 Wp (sums of E[s] y)
 *   PM_DET_GSC                       0 xpt_s / xpt_ss sums   1 xpt_sz sums   2 xpt_szsz sums
 *   PM_DET_GEMM                      0 the K-slices of pm_gemm_tn_acc_f64
 *   PM_DET_MCA                       0 Wq   1 Wp   2 pi   3 sum of q e   4 sums of log-evidences */
#define PM_DET_BSC_FUSED8 0
#define PM_DET_WP_SPARSE 1
#define PM_DET_GSC 2
#define PM_DET_GEMM 3
#define PM_DET_MCA 4
#define PM_DET_DSC 5
#define PM_DET_BSC_ROWS16 6   /* the other BSC kernel files: categories as PM_DET_BSC_FUSED8 */
#define PM_DET_BSC_FUSED 7
#define PM_DET_BSC_KERNELS 8
int pm_det_build(void);
int pm_det_set_quanta(int unit, const double *M8, void *stream);
/* GSC, deterministic build, inside an EM loop: the quanta of the next E-step (PM_DET_GSC) and of its M-step's contraction
 * (PM_DET_GEMM and PM_DET_WP_SPARSE), derived AND installed by one kernel from parameters that are on the device only -- gram
 * (H,H: its diagonal = |W_h|^2), psi_sq (H,H), tables as pm_gsc_mstep_finish_f64 leaves them (mu in row 6, 1/sigma_sq in
 * tables[8 H]); ymax = max |y_nd|, ynmax = max |y_n|, n = datapoints of the shard.  quanta16 (device): a copy of what was
 * installed, [8 of PM_DET_GSC | 8 of the contraction].  The bounds are those of GSC._det_quanta (gsc_et.py here; the
 * reference has no such mode).  -2 in the default build. */
int pm_gsc_det_quanta_f64(const double *gram, int64_t ldg, const double *psi_sq, const double *tables, int64_t H,
                          int64_t gamma, double ymax, double ynmax, double n, double *quanta16, void *stream);
/* A device-side row list that was built with an atomic counter (`rows[0 .. *count)`, distinct values in [0, N)) sorted
 * ascending, in place: what makes a gathered contraction over it (pm_gemm_tn_acc_rows_f64) independent of the order its
 * producers finished in.  `flags`: 8-byte aligned, 8 * ceil(N / 8) bytes that are ZERO on entry and zero again on return
 * (every listed row sets its byte; a workgroup per 8192 rows compacts them) + 4 * ceil(N / 8192) bytes of scratch behind them. */
int pm_sort_row_list_i32(int32_t *rows, const int32_t *count, int64_t N, unsigned char *flags, void *stream);
const char *pm_error_string(int code);

/* ---------------------------------------------------------------------------------------
 * Dense building blocks (hand-written f64 MFMA kernels, v_mfma_f64_16x16x4_f64)
 * ------------------------------------------------------------------------------------- */

/* C[M,N] = A[M,K] . B[N,K]^T   (both operands K-contiguous).
 * Scores a = W.y for every datapoint (replaces np.inner(W, y) per datapoint,
 * bsc_et.py:111, and the (W-y)**2 sums of bsc_et.py:177 via the Gram identity) with
 * A = Y, B = W^T-as-rows (H,D); also G = W.W^T (replaces np.inner(W,W), bsc_et.py:111). */
int pm_gemm_nt_f64(const double *A, int64_t lda, const double *B, int64_t ldb,
                   double *C, int64_t ldc, int64_t M, int64_t N, int64_t K, void *stream);

/* The same product for SMALL outputs (M, N <= 1024), deterministic: one workgroup per 16 x 16 tile, the K range split over
 * its eight wavefronts and summed in a fixed order -- no atomics, so every rank holding the same operands gets the same
 * bits (pm_gemm_nt_f64 splits such shapes over K with f64 atomics).  A == B: the Gram matrix W W^T of the models' energy
 * algebra (bsc_et.py:177-183 in Gram form), stored symmetric bit for bit. */
int pm_gemm_nt_small_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                         int64_t M, int64_t N, int64_t K, void *stream);
/* C = A . B (A: M x K, B: K x N, row-major; M <= 1024), same scheme: the W solve X = Wq^-1 . Wp behind the device inverse
 * (np.linalg.lstsq(Wq, Wp) at bsc_et.py:380, dsc_et.py:741) in one launch -- no zero fill, no K-slice atomics. */
int pm_gemm_nn_small_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                         int64_t M, int64_t N, int64_t K, void *stream);

/* C[M,N] += A[K,M]^T . B[K,N]   (reduction over the leading/row index, split over
 * workgroups, f64 atomics into C; C must be initialised by the caller).
 * Wp = E[s]^T . Y  -- replaces the per-datapoint np.outer accumulation of
 * bsc_et.py:339-363 (my_Wp). */
int pm_gemm_tn_acc_f64(const double *A, int64_t lda, const double *B, int64_t ldb,
                       double *C, int64_t ldc, int64_t M, int64_t N, int64_t K, void *stream);
/* The same product, decided on the device: the kernels return at once unless *gate (a device double) is non-zero.  The
 * dense half of the pair (pm_bsc_wp_sparse_f64, this) that is enqueued without the host knowing which one applies. */
int pm_gemm_tn_acc_gated_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                             int64_t M, int64_t N, int64_t K, const double *gate, void *stream);
/* C[M,N] += sum over the rows r = rows[0 .. *count) of A[r,:M]^T . B[r,:N]: the same product over a device-side list of
 * rows with a device-side length (count <= max_rows) -- the dense datapoints of GSC's moment contraction
 * (gsc_et.py:592-625; pm_gsc_estep_lists_f64 leaves the list).  `zero_row`: a row of zeros in A and in B (the ragged end of
 * the list reads it).  M % 128 == 0, N % 128 == 0, 16-byte aligned operands, else PM_ERANGE. */
int pm_gemm_tn_acc_rows_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t M,
                            int64_t N, const int32_t *rows, const int32_t *count, int64_t max_rows, int64_t zero_row,
                            void *stream);

/* out[n] = sum_d Y[n,d]^2 -- np.inner(y, y) of bsc_et.py:111 and (y**2).sum() of :172;
 * computed once per resident data shard. */
int pm_row_sqnorm_f64(const double *Y, int64_t ldy, int64_t N, int64_t D, double *out, void *stream);

/* sums[d] += sum_n Y[n,d] (center == NULL) or sum_n (Y[n,d] - center[d])^2: the two collective means of
 * CAModel.standard_init (camodels/__init__.py:209-217, dsc_et.py:892-899) over a resident shard.
 * `sums` (D) is accumulated into (caller zeroes it). */
int pm_col_moments_f64(const double *Y, int64_t ldy, int64_t N, int64_t D, const double *center,
                       double *sums, void *stream);

/* sums[d] += sum over the datapoints n with lse[n] >= cut of Y[n,d]: my_data_sum of BSC_ET.M_step when 'mu' is learned
 * (bsc_et.py:422-430) over the datapoints the truncation keeps (:247-258; cut = -inf keeps all).  `sums` is accumulated into. */
int pm_col_sum_kept_f64(const double *Y, int64_t ldy, int64_t N, int64_t D, const double *lse, double cut, double *sums,
                        void *stream);

/* out[n] = sum_d w[d] Y[n,d]^2 -- y^T Sigma^-1 y of GSC with a diagonal noise covariance
 * (gsc_et.py:418-419, 476, 780-781). */
int pm_row_wsqnorm_f64(const double *Y, int64_t ldy, int64_t N, int64_t D, const double *w, double *out,
                       void *stream);

/* inv = (U + U^T - diag(U) + diag(diag_add))^-1 for a symmetric positive definite n x n matrix given
 * by its upper triangle `upper` (n <= 256, one workgroup, Gauss-Jordan in registers, no pivoting).
 * `full` (optional) receives the assembled matrix, `pivots` (optional, 2 doubles) the smallest and
 * largest pivot: a non-positive or vanishing smallest pivot means "numerically singular".
 * Solves the H x H system of the M-step, np.linalg.lstsq(Wq, Wp) at bsc_et.py:380, as
 * W_new = inv . Wp (one pm_gemm_tn_acc_f64). */
int pm_spd_inverse_f64(const double *upper, int64_t ldu, const double *diag_add, int64_t n, double *full,
                       double *inv, int64_t ldo, double *pivots, void *stream);

/* The same inverse, warm-started: `prev_inv` (n x n, leading dimension n) is the inverse of a nearby matrix -- the
 * previous EM step's second moments.  Four Newton-Schulz steps X <- X + X (I - A X) on the matrix cores refine it into
 * `inv` (symmetrised); the Gauss-Jordan sweep is launched behind them and returns at once when the refinement converged:
 * the LAST residual the iteration formed, R_3 = (I - A prev_inv)^8, is below 1e-8 in the Frobenius norm, so the result's
 * is below 1e-16 up to rounding (round 6; until then the START residual had to be below 0.1 in the Frobenius norm, which
 * rejected the few-per-cent-in-every-direction residuals of an annealing ramp although they converge).  Then pivots =
 * {min_i 1 / inv_ii, max_i A_ii}: a lower bound of the smallest and an upper bound of the largest pivot of the sweep.  Else
 * the sweep overwrites `inv` with the exact inverse as pm_spd_inverse_f64 would.  The decision is taken on the device from
 * sums formed in a fixed order and left for the caller in pivots[2] (`pivots`: THREE doubles here): 1.0 = the refinement
 * stands and ||I - A prev_inv||_F < 1, 2.0 = it stands from a start further away, 0.0 = the sweep ran (`inv` is then exact
 * to cond(A) eps only: a caller that skipped the iterative refinement of its solve repeats it with one).  `work`:
 * pm_spd_inverse_warm_work_len(n) doubles; `full` (required, leading dimension ldo = n) receives the assembled matrix.
 * Same role as pm_spd_inverse_f64 (np.linalg.lstsq(Wq, Wp), bsc_et.py:380).
 * pm_spd_inverse_warm_long_f64: the same from a start that may be FAR -- a data-truncation step (bsc_et.py:247-258) whose
 * kept set jumped, a temperature step: the start is scaled by 1 / ||A prev_inv||_inf (both matrices are symmetric positive
 * definite, so the scaled residual's spectrum lies in [0, 1) and the iteration converges from any such start) and eight +
 * one steps run: accepted for ||A prev_inv||_inf / lambda_min(A prev_inv) up to ~14, ~60 us instead of the sweep's 0.3 ms. */
int64_t pm_spd_inverse_warm_work_len(int64_t n);
int pm_spd_inverse_warm_f64(const double *upper, int64_t ldu, const double *diag_add, int64_t n, const double *prev_inv,
                            int64_t ldp, double *work, double *full, double *inv, int64_t ldo, double *pivots,
                            void *stream);
int pm_spd_inverse_warm_long_f64(const double *upper, int64_t ldu, const double *diag_add, int64_t n,
                                 const double *prev_inv, int64_t ldp, double *work, double *full, double *inv, int64_t ldo,
                                 double *pivots, void *stream);

/* `batch` warm-started inverses at once (every kernel of pm_spd_inverse_warm_f64 gets a batch dimension): matrix b is
 * read at upper + b*stride_in (diag_add + b*n), its previous inverse at prev_inv + b*stride_prev (dense n x n), its
 * result written at inv + b*stride_out (dense n x n), pivots 2 doubles per matrix; `work`: batch *
 * pm_spd_inverse_warm_work_len(n) doubles (matrix b's accepted flag -- see above -- in the last word of its share).  GSC's M-step: (sum xpt_szsz)^-1 and (sum xpt_ss + eps I)^-1. */
int pm_spd_inverse_warm_batch_f64(const double *upper, int64_t ldu, int64_t stride_in, const double *diag_add, int64_t n,
                                  const double *prev_inv, int64_t stride_prev, double *work, double *inv,
                                  int64_t stride_out, double *pivots, int64_t batch, void *stream);

/* The same launches with GENERAL (non-symmetric) matrices among the batch: bit b of `general_mask` says matrix b is given in
 * full (row-major, both triangles; diag_add still added) and is refined by the left-sided Newton-Schulz iteration
 * X <- X + (I - X A^T) X from prev_inv + b*stride_prev -- its result is the inverse of the TRANSPOSE, (A^T)^-1 = (A^-1)^T,
 * not symmetrised.  GSC from its second EM step on: psi_sq leaves the M-step non-symmetric (gsc_et.py:660-675), with it
 * Lambda^-1 and sum xpt_szsz, which gsc_et.py:625 inverts as it is; W_new^T = (A^-1)^T Wp^T is then one pm_gemm_nt_f64.
 * `accepted` (batch doubles): 1.0 = the refinement of matrix b stands, 0.0 = its start was too far off and the guarded
 * sweep ran -- which for a general matrix inverts only its upper-mirrored stand-in: the caller then inverts on the host,
 * as the reference does.  A cold start is pm_spd_inverse_batch_f64 on the same buffer (upper-mirrored: a start within
 * ~1e-6) followed by this entry. */
int pm_inverse_warm_batch_f64(const double *mats, int64_t ldu, int64_t stride_in, const double *diag_add, int64_t n,
                              const double *prev_inv, int64_t stride_prev, double *work, double *inv, int64_t stride_out,
                              double *pivots, double *accepted, int64_t batch, uint32_t general_mask, void *stream);

/* `batch` independent inverses in one launch, one workgroup each (they run on different CUs at once): matrix b is
 * read at upper + b*stride_in, written at inv (and full, if given) + b*stride_out; diag_add (optional) holds n
 * doubles and pivots 2 doubles per matrix.  GSC's M-step needs (sum xpt_szsz)^-1 and (sum xpt_ss + eps I)^-1
 * (gsc_et.py:625, 673). */
int pm_spd_inverse_batch_f64(const double *upper, int64_t ldu, int64_t stride_in, const double *diag_add, int64_t n,
                             double *full, double *inv, int64_t ldo, int64_t stride_out, double *pivots,
                             int64_t batch, void *stream);

/* ---------------------------------------------------------------------------------------
 * Distributed k-th largest (the data-truncation cut): parallel.allsort(all_denoms)[-N_use], prosper/utils/parallel.py:87-110
 * with its consumers bsc_et.py:252, mca_et.py:252, dsc_et.py:832.  Radix select over an order-preserving 64-bit key
 * of the doubles, one digit per round: every rank histograms the digit at bit `shift` (`bits` wide, <= 12) of its
 * values that agree with the digits decided so far (state[0]); the caller sum-all-reduces the 4096 bins (RCCL) and
 * pm_kth_scan picks the bin that holds the k-th largest: state[0] |= digit << shift, state[1] = rank inside that bin;
 * `hist` is cleared.  Rounds (shift, bits) = (52,12) (40,12) (28,12) (16,12) (4,12) (0,4) decide all 64 bits;
 * pm_kth_value_f64 converts state[0] back: exactly the value a full sort returns.
 *   state (2 x uint64): [0] digits decided so far (0 before the first round), [1] k, 1-based from the largest.
 *   hist  (4096 x uint64), zeroed by the caller before the first round.
 * ------------------------------------------------------------------------------------- */
int pm_kth_hist_f64(const double *x, int64_t n, const uint64_t *state, int shift, int bits, uint64_t *hist, void *stream);
int pm_kth_scan(uint64_t *hist, uint64_t *state, int shift, int bits, void *stream);
int pm_kth_value_f64(const uint64_t *state, double *out, void *stream);
/* The same select in 1 + 6 + 1 launches instead of 13 (round 6: what a data-truncation step pays once nothing else in it
 * waits for the host): `hists` holds one histogram of 4096 bins PER ROUND (6 x 4096 uint64, zeroed by the caller -- one
 * fill), `states` seven (prefix, k) pairs, slot 0 = (0, k) from the caller.  pm_kth_round_f64(round r) first repeats, in
 * every workgroup, the scan of round r - 1 (its histogram -- sum-all-reduced by the caller between the two calls -- and
 * states[r - 1]; `shift_prev` / `bits_prev`: that round's digit; workgroup 0 stores the result as states[r]), then
 * histograms the digit [shift, shift + bits) of the values that agree with the prefix into hists[r].  pm_kth_final_f64
 * scans the last round and converts: out[0] = the k-th largest, exactly the value a full sort returns.  The value stays
 * on the device (pm_bsc_defer_apply_f64 reads it there).  n = 0 (an empty shard) is fine. */
int pm_kth_round_f64(const double *x, int64_t n, uint64_t *states, uint64_t *hists, int round, int shift_prev, int bits_prev,
                     int shift, int bits, void *stream);
int pm_kth_final_f64(uint64_t *states, const uint64_t *hists, int rounds, int shift_prev, int bits_prev, double *out,
                     void *stream);
/* ... without the fill in front: pm_kth_round_k_f64 (round 0 only) takes the rank k0 as an argument instead of from states[1],
 * pm_kth_final_z_f64 with rezero != 0 leaves the `rounds` histograms zeroed for the next select on the same buffer (which the
 * caller zeroes once, when it allocates it). */
int pm_kth_round_k_f64(const double *x, int64_t n, uint64_t *states, uint64_t *hists, int round, int shift_prev, int bits_prev,
                       int shift, int bits, int64_t k0, void *stream);
int pm_kth_final_z_f64(uint64_t *states, uint64_t *hists, int rounds, int shift_prev, int bits_prev, double *out, int rezero,
                       void *stream);

/* ---------------------------------------------------------------------------------------
 * Binary Sparse Coding (prosper/em/camodels/bsc_et.py)
 * ------------------------------------------------------------------------------------- */

/* select_Hprimes, bsc_et.py:98-115.
 *   scores  (N,H)  a[n,h] = <W_h, y_n>            (from pm_gemm_nt_f64)
 *   wnorm2  (H)    |W_h|^2  (diag of G; element stride `wnorm2_stride`, pass H+1 to read diag(G) in place)
 *   ynorm2  (N)    |y_n|^2
 *   cand    (N,Hprime) int32 out: indices of the Hprime largest a/|W_h|/|y|, ascending
 *           (last = best), i.e. np.argsort(sim)[-Hprime:]; ties broken towards the larger index. */
int pm_bsc_select_f64(const double *scores, int64_t lds, const double *wnorm2, int64_t wnorm2_stride,
                      const double *ynorm2, int64_t N, int64_t H, int64_t Hprime,
                      int32_t *cand, void *stream);

/* Scalars of one E-step (bsc_et.py:151-154,187-190). */
typedef struct pm_bsc_estep_params {
    double pil_bar;      /* log(pi / (1 - pi))                                   */
    double ecoef;        /* beta * pre1 = -(1/T) / (2 sigma^2): multiplies energies */
    double prior_scale;  /* 1 (anneal_prior false) or beta (anneal_prior true)    */
    double mu_sqnorm;    /* |mu|^2 (0 when mu == 0)                              */
} pm_bsc_estep_params;

/* E_step, bsc_et.py:119-192: log-pseudo-joints of the truncated state set
 *   [null ; singletons h = 0..H-1 ; multi-cause states in generate_state_matrix order].
 *   gram    (H,H)  G = W.W^T
 *   wmu     (H)    W.mu, or NULL when mu == 0
 *   ymu     (N)    Y.mu, or NULL when mu == 0
 *   cand    (N,Hprime) int32
 *   state_masks (S) uint16: bit j set <=> state uses candidate position j
 *                  (rows of camodels/__init__.py:21-47's state_matrix)
 *   logpj   (N, 1+H+S) out;  lse (N) out = log sum_k exp(logpj[n,k]) (stable), or NULL. */
int pm_bsc_estep_f64(const double *scores, int64_t lds, const double *gram, const double *ynorm2,
                     const double *wmu, const double *ymu, const int32_t *cand,
                     const uint16_t *state_masks, int64_t S,
                     const pm_bsc_estep_params *params_host,
                     int64_t N, int64_t H, int64_t Hprime,
                     double *logpj, int64_t ldl, double *lse, void *stream);

/* Layout of the packed M-step statistics buffer (float64, one RCCL all-reduce per EM step):
 *   [ Wp (H*D) | Wq (H*H) | qdiag (H) | mus (H) | scalars (PM_BSC_NSCALARS) ]
 * scalars: [0] sum_n sum_k q_nk e_nk (sigma statistic, bsc_et.py:395-415)
 *          [1] sum over kept n of log sum_k exp(logpj) (Fs, bsc_et.py:265)
 *          [2] number of kept datapoints (my_N after truncation, bsc_et.py:257)
 *          [3] number of datapoints whose E[s] row had more than PM_BSC_NZ_MAX non-zeros while the E-step pass wrote
 *              non-zero lists (pm_bsc_estep_fused8_nz_f64); gates pm_bsc_wp_sparse_f64 / pm_gemm_tn_acc_gated_f64 */
#define PM_BSC_NSCALARS 4
int64_t pm_bsc_stats_len(int64_t H, int64_t D);
int64_t pm_bsc_stats_offset_wq(int64_t H, int64_t D);
int64_t pm_bsc_stats_offset_qdiag(int64_t H, int64_t D);
int64_t pm_bsc_stats_offset_mus(int64_t H, int64_t D);
int64_t pm_bsc_stats_offset_scalars(int64_t H, int64_t D);

/* Per-datapoint part of M_step, bsc_et.py:271-272,334-366,395-415: posterior weights
 * q = exp(logpj - lse) of every kept datapoint (lse[n] >= lse_cut; pass -inf to keep all),
 * expectations E[s] (N,H) out (zero rows for cut datapoints), Wq block scatter, and the
 * scalar statistics.  `pair_ptr` (Hprime*Hprime+1) / `pair_states` are the CSR lists of
 * multi-cause states containing candidate positions (i,j) (derived from state_masks on
 * the host; pair_len = pair_ptr[Hprime*Hprime] entries).  Only the upper triangle of Wq is
 * accumulated (row <= col); the caller mirrors it and adds diag(qdiag).  Accumulates into `stats` (caller zeroes it once per EM step). */
int pm_bsc_mstep_rows_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                          const int32_t *cand, const uint16_t *state_masks, int64_t S,
                          const int32_t *pair_ptr, const uint16_t *pair_states, int64_t pair_len,
                          const pm_bsc_estep_params *params_host,
                          int64_t N, int64_t H, int64_t D, int64_t Hprime,
                          double *expect, int64_t lde, double *stats, void *stream);

/* ---------------------------------------------------------------------------------------
 * BSC fast path: 16 lanes per datapoint (four datapoints per wavefront, DPP row reductions),
 * incremental multi-cause energies, negligible posterior terms skipped.  Same results as the
 * entry points above; available when pm_bsc_rows16_supported(H, Hprime, S) != 0 (H <= 512 and
 * the per-datapoint state areas fit LDS).
 * ------------------------------------------------------------------------------------- */
int pm_bsc_rows16_supported(int64_t H, int64_t Hprime, int64_t S);
/* ... and whether the list-writing form of the M-step row pass (pm_bsc_mstep_rows16_nz_f64: it keeps one more score row
 * per datapoint slot in LDS) fits as well; when it does not, callers run pm_bsc_mstep_rows16_f64 and the dense
 * pm_gemm_tn_acc_f64 (same statistics, bsc_et.py:334-366). */
int pm_bsc_rows16_nz_supported(int64_t H, int64_t Hprime, int64_t S);

/* select_Hprimes (bsc_et.py:98-115) and/or E_step (bsc_et.py:119-192) in one pass over the
 * scores: mode bit 0 = select (write `cand`; otherwise `cand` is an input), bit 1 = E-step
 * (write `logpj`, `lse`).  `state_parents[s]` = index of the state obtained from state s by
 * dropping its highest candidate position (0xFFFF when that leaves a singleton);
 * `size_offsets_host[g-2]` = index of the first state with g causes, g = 2..gamma, and
 * size_offsets_host[gamma-1] = S (states are ordered by size, camodels/__init__.py:32-35).
 * Selection ties: similarities equal in their leading 42 mantissa bits rank by latent index.
 * Further selection modes (other models reuse the selection pass): bit 2 = the Hprime SMALLEST
 * first (mca_et.py:107), bit 3 = rank `scores` as they are (no normalisation), bit 4 = rank the
 * squared distance |W_h|^2 - 2 scores[n,h] with |W_h|^2 from the Gram diagonal (mmca_et.py:119-120). */
int pm_bsc_select_estep_f64(const double *scores, int64_t lds, const double *gram, const double *ynorm2,
                            const double *wmu, const double *ymu, const uint16_t *state_masks,
                            const uint16_t *state_parents, const int32_t *size_offsets_host, int64_t S,
                            int64_t gamma, const pm_bsc_estep_params *params_host, int64_t N, int64_t H,
                            int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl,
                            double *lse, void *stream);

/* Scores GEMM + select_Hprimes + E_step in ONE kernel (bsc_et.py:98-115 and :119-192): what pm_gemm_nt_f64 followed by
 * pm_bsc_select_estep_f64 computes, without the (N,H) scores buffer -- a workgroup owns 64 datapoints x all H latents
 * and runs the row pass out of its MFMA accumulators.  Y (N,D) data, Wt (H,D) = W^T; the other arguments, the
 * `mode` bits 0 and 1 and the outputs are those of pm_bsc_select_estep_f64 (BSC's own ranking only).  Needs
 * pm_bsc_fused_supported(H, D, Hprime, S): H <= 256, D a multiple of 8 (callers zero-pad the K dimension), 16-byte
 * aligned rows (ldy, ldw even).
 * M-step statistics in the same pass (stats != NULL; Hprime <= 8, mode & 2, lse given): for EVERY datapoint -- i.e. when
 * no data truncation follows (bsc_et.py:247-258 skipped) -- what pm_bsc_mstep_rows16_f64 computes from logpj:
 * E[s] rows into expect (N,H), and into the packed buffer `stats` of a model with D_stats observed dimensions (layout
 * above): the multi-cause part of Wq's upper triangle INCLUDING its diagonal, mus = sum_n E[s], the scalars.  The
 * qdiag block is not written: qdiag = mus - diag(Wq block) (s_h^2 = s_h).  `stats` is accumulated into. */
int pm_bsc_fused_supported(int64_t H, int64_t D, int64_t Hprime, int64_t S);
/* Workgroups of that kernel one CU holds at once for this shape (2 by design: one on the matrix pipe, one in its row
 * passes); diagnostic, <= 0 when unsupported. */
int pm_bsc_fused_occupancy(int64_t H, int64_t D, int64_t Hprime, int64_t S);
int pm_bsc_estep_fused_f64(const double *Y, int64_t ldy, const double *Wt, int64_t ldw, const double *gram,
                           const double *ynorm2, const double *wmu, const double *ymu,
                           const uint16_t *state_masks, const uint16_t *state_parents,
                           const int32_t *size_offsets_host, int64_t S, int64_t gamma,
                           const pm_bsc_estep_params *params_host, int64_t N, int64_t D, int64_t H,
                           int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl, double *lse,
                           double *expect, int64_t lde, double *stats, int64_t D_stats, void *stream);

/* The same pass for config 2's shape class as a 16-wavefront kernel (csrc/bsc_fused8.hip): a workgroup of 1024 threads
 * owns 128 datapoints x all latents, one per CU; a wavefront accumulates 16 datapoints x 128 latents (<= 128 registers:
 * four wavefronts per SIMD), the row passes run one half-wavefront per datapoint.  Same arguments, outputs and reference
 * lines (bsc_et.py:98-115, :119-192) as pm_bsc_estep_fused_f64.  Needs pm_bsc_fused8_supported(H, D, Hprime, S) --
 * 128 < H <= 256, Hprime = 5 .. 8 (round 6; 8 until then), S that of gamma = 3 or 4 (84 / 154 at Hprime = 8), D a multiple of 8 -- and pm_bsc_fused8_whole_shard(H, Hprime, gamma, S): the
 * complete state set of sizes 2 .. gamma, gamma 3 or 4.  The call takes a WHOLE shard: whole rounds of resident
 * workgroups run 128-row tiles, a ragged remainder of up to two rounds of 16-row workgroups that split K four ways runs in
 * a second launch (no scores buffer, no separate row kernel), and `stats` / `expect` (the M-step statistics of
 * pm_bsc_estep_fused_f64) are accepted.  pm_bsc_fused8_main_rows(N, D): the leading rows the main launch takes.
 * `part`: 0 both launches, 1 only the main one, 2 only the remainder (callers that time or trace them separately). */
int pm_bsc_fused8_supported(int64_t H, int64_t D, int64_t Hprime, int64_t S);
int pm_bsc_fused8_whole_shard(int64_t H, int64_t Hprime, int64_t gamma, int64_t S);
int64_t pm_bsc_fused8_main_rows(int64_t N, int64_t D);
int pm_bsc_estep_fused8_f64(const double *Y, int64_t ldy, const double *Wt, int64_t ldw, const double *gram,
                            const double *ynorm2, const double *wmu, const double *ymu,
                            const uint16_t *state_masks, const uint16_t *state_parents,
                            const int32_t *size_offsets_host, int64_t S, int64_t gamma,
                            const pm_bsc_estep_params *params_host, int64_t N, int64_t D, int64_t H,
                            int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl, double *lse,
                            double *expect, int64_t lde, double *stats, int64_t D_stats, int part, void *stream);

/* pm_bsc_estep_fused8_f64 that also leaves every E[s] row as a list of its non-zeros (statistics passes only: `stats`
 * given): nz_idx (N x PM_BSC_NZ_MAX uint16: latent indices, unused slots 0xFFFF) and nz_val (N x PM_BSC_NZ_MAX).  Past the
 * annealing phase a posterior of the truncated state set puts weight on a handful of latents (3.7 of 256 per datapoint
 * on config 2), and the M-step's Wp = E[s]^T Y (bsc_et.py:339-363) needs only those rows of the product.  A row with more
 * than PM_BSC_NZ_MAX non-zeros is counted in scalars[3], slot 0 of its list reads PM_BSC_NZ_OVERFLOW, and ITS dense row is
 * stored in `expect`; the dense row of a datapoint whose list is complete is NOT stored (round 5: the list is the row -- N x H
 * doubles of write traffic per step less; pm_bsc_expand_lists_gated_f64 rebuilds those rows when the dense product runs).
 * Statistics of the whole-shard passes (pm_bsc_fused8_whole_shard; both entries): the diagonal of the second moments is
 * left in `qdiag` in full (= mus: E[s_h^2] = E[s_h]) and the diagonal of the Wq block stays zero -- the assembled matrix
 * upper + upper^T - diag(upper) + diag(qdiag) is the same, no fix-up pass is needed. */
#define PM_BSC_NZ_MAX 16
#define PM_BSC_NZ_OVERFLOW 0xFFFEu      /* nz_idx[n][0] of a datapoint whose list overflowed (its dense row is in `expect`) */
int pm_bsc_estep_fused8_nz_f64(const double *Y, int64_t ldy, const double *Wt, int64_t ldw, const double *gram,
                               const double *ynorm2, const double *wmu, const double *ymu,
                               const uint16_t *state_masks, const uint16_t *state_parents,
                               const int32_t *size_offsets_host, int64_t S, int64_t gamma,
                               const pm_bsc_estep_params *params_host, int64_t N, int64_t D, int64_t H,
                               int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl, double *lse,
                               double *expect, int64_t lde, double *stats, int64_t D_stats, uint16_t *nz_idx,
                               double *nz_val, int part, void *stream);
/* DEFERRED statistics for a data-truncation step (bsc_et.py:247-258 keeps the N_use datapoints with the largest evidence;
 * which ones is known only after the E-step of every rank).  pm_bsc_estep_fused8_defer_f64 is pm_bsc_estep_fused8_nz_f64
 * that, given `records` (N x PM_BSC_DEFER_LD doubles), accumulates NOTHING into `stats` but the overflow count scalars[3]:
 * per datapoint it leaves the non-zero list of E[s] (as above; the dense row when the list overflowed) and the record
 * [36 entries k (k + 1) / 2 + i (i <= k candidate positions) of the multi-cause states' second moments, diagonal entries 0 |
 * ecoef sum_k q_k e_k | pad].  `records` NULL: exactly pm_bsc_estep_fused8_nz_f64.
 * pm_bsc_defer_apply_f64 then adds the records of the datapoints with lse[n] >= *cut (a DEVICE double: the radix select's
 * result never visits the host; values below log(2^-1075) mean "keep all", bsc_et.py:253) into `stats` -- Wq pair block,
 * mus = qdiag, sum q e, sum lse, kept count, i.e. what the in-pass statistics of the kept datapoints would have been
 * (bsc_et.py:334-366, 395-415) -- and empties the lists (zeroes the dense rows) of the others, so that
 * pm_bsc_wp_sparse_expand_f64 / pm_gemm_tn_acc_gated_f64 behind it see kept datapoints only.  H <= 256, Hprime = 5 .. 8. */
#define PM_BSC_DEFER_LD 40
int pm_bsc_estep_fused8_defer_f64(const double *Y, int64_t ldy, const double *Wt, int64_t ldw, const double *gram,
                                  const double *ynorm2, const double *wmu, const double *ymu,
                                  const uint16_t *state_masks, const uint16_t *state_parents,
                                  const int32_t *size_offsets_host, int64_t S, int64_t gamma,
                                  const pm_bsc_estep_params *params_host, int64_t N, int64_t D, int64_t H,
                                  int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl, double *lse,
                                  double *expect, int64_t lde, double *stats, int64_t D_stats, uint16_t *nz_idx,
                                  double *nz_val, double *records, int part, void *stream);
int pm_bsc_defer_apply_f64(const double *lse, const double *cut, const int32_t *cand, const double *records,
                           uint16_t *nz_idx, const double *nz_val, double *expect, int64_t lde, double *stats,
                           const pm_bsc_estep_params *params_host, int64_t N, int64_t H, int64_t D, int64_t Hprime,
                           void *stream);
/* Wp (the leading H x D block of `stats`) += sum_n sum_t nz_val[n,t] . Y[n,:] into row nz_idx[n,t]  -- my_Wp of
 * bsc_et.py:339-363 from the non-zero lists; returns at once on the device when scalars[3] of `stats` is non-zero (some
 * list overflowed: pm_gemm_tn_acc_gated_f64 on `expect` does the work then).  H <= 256. */
int pm_bsc_wp_sparse_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *stats,
                         int64_t N, int64_t H, int64_t D, void *stream);
/* The dense rows pm_bsc_estep_fused8_nz_f64 did not store, for the dense product that runs when some list overflowed:
 * expect[n, :] = 0, then expect[n, nz_idx[n,t]] = nz_val[n,t], for every datapoint whose list is complete (slot 0 is not
 * PM_BSC_NZ_OVERFLOW).  Decided on the device like the products themselves: returns at once unless *gate (scalars[3]) is
 * non-zero.  Enqueue it between pm_bsc_wp_sparse_f64 and pm_gemm_tn_acc_gated_f64 (or use pm_bsc_wp_sparse_expand_f64). */
int pm_bsc_expand_lists_gated_f64(const uint16_t *nz_idx, const double *nz_val, double *expect, int64_t lde,
                                  int64_t N, int64_t H, const double *gate, void *stream);
/* pm_bsc_wp_sparse_f64 and pm_bsc_expand_lists_gated_f64 in ONE launch (what the M-step enqueues): the sparse product, or --
 * scalars[3] of `stats` non-zero -- the completion of `expect` for the dense product behind it. */
int pm_bsc_wp_sparse_expand_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *stats,
                                double *expect, int64_t lde, int64_t N, int64_t H, int64_t D, void *stream);
/* The same accumulation for any model whose statistics start with Wp = E[s]^T Y (DSC / TSC, dsc_et.py:703-735): Wp (H x D,
 * leading dimension ldw) and the device-side gate (a double: non-zero = skip) are given explicitly. */
int pm_wp_sparse_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *Wp, int64_t ldw,
                     const double *gate, int64_t N, int64_t H, int64_t D, void *stream);
/* The transposed output, no gate: C (D x H, leading dimension ldc) += Y^T . V with the rows of V (N x H) given as lists --
 * the listed rows of GSC's contraction [Y | xpt_s | xpt_sz]^T . xpt_sz (gsc_et.py:592-625); rows with an empty list
 * contribute nothing here (pm_gemm_tn_acc_rows_f64 takes them). */
int pm_wp_sparse_t_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *C, int64_t ldc,
                       int64_t N, int64_t H, int64_t D, void *stream);

/* Fast-path twin of pm_bsc_mstep_rows_f64 (same outputs, same `stats` layout). */
int pm_bsc_mstep_rows16_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                            const int32_t *cand, const uint16_t *state_masks, int64_t S,
                            const pm_bsc_estep_params *params_host, int64_t N, int64_t H, int64_t D,
                            int64_t Hprime, double *expect, int64_t lde, double *stats, void *stream);
/* ... that also leaves the non-zeros of every E[s] row as a list (format of pm_bsc_estep_fused8_nz_f64; rows with more
 * than PM_BSC_NZ_MAX non-zeros count in scalars[3]).  The M-step after a data-truncation step. */
int pm_bsc_mstep_rows16_nz_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                               const int32_t *cand, const uint16_t *state_masks, int64_t S,
                               const pm_bsc_estep_params *params_host, int64_t N, int64_t H, int64_t D,
                               int64_t Hprime, double *expect, int64_t lde, double *stats, uint16_t *nz_idx,
                               double *nz_val, void *stream);

/* ---------------------------------------------------------------------------------------
 * Maximal Causes Analysis (prosper/em/camodels/mca_et.py)
 * ------------------------------------------------------------------------------------- */

/* R[n,h] = sum_d max(W[h,d] - Y[n,d], 0) = sum_d |max(W_hd, y_d) - y_d|: the candidate-selection
 * score of mca_et.py:104-106 (W (H,D) row-major).  Candidates = the Hprime SMALLEST per row,
 * ascending: pm_bsc_select_estep_f64 with mode = 1|4|8 (select, smallest-first, raw scores). */
int pm_mca_select_scores_f64(const double *Y, int64_t ldy, const double *W, int64_t ldw, double *R,
                             int64_t ldr, int64_t N, int64_t H, int64_t D, void *stream);

/* Scalars of one MCA step (mca_et.py:142-149, 213-221) or MMCA step (mmca_et.py:156-168, 251-262). */
typedef struct pm_mca_params {
    double pil_bar;   /* log(pi / (1 - pi))                                 */
    double pre1;      /* -1 / (2 sigma^2)                                   */
    double beta;      /* 1 / T (applied to the log-joints in the M-step)    */
    double inv_rho;   /* 1 / rho, rho = 1 / (1 - 1 / max(T, 1.05)) (MMCA: T bound 1.2, rho in [1, 35]) */
    double signed_w;  /* 0: MCA (W > 0).  1: MMCA, signed W (prosper/em/camodels/mmca_et.py): Wrho holds
                         sign(W)|W|^rho, Wrm1 holds |W|^(rho-1); Wbar_sd = sign(t)|t|^(1/rho), t = sum Wrho;
                         the M-step factor is min(1, (|W_jd| / |Wbar_sd|)^(rho-1)) (mmca_et.py:321-324) */
} pm_mca_params;

/* E_step, mca_et.py:114-179.  scores = Y.W^T (pm_gemm_nt_f64), wnorm2 = |W_h|^2 (contiguous),
 * Wrho = W^rho (H,D).  Multi-cause states use Wbar_sd = (sum_{j in s} Wrho[c_j,d])^(1/rho).
 * Outputs logpj (N,1+H+S), lse1 = log sum_k exp(logpj), lseb = log sum_k exp(beta*logpj). */
int pm_mca_estep_f64(const double *scores, int64_t lds, const double *wnorm2, const double *ynorm2,
                     const double *Y, int64_t ldy, const double *Wrho, const int32_t *cand,
                     const uint16_t *state_masks, int64_t S, const pm_mca_params *params_host, int64_t N,
                     int64_t H, int64_t D, int64_t Hprime, double *logpj, int64_t ldl, double *lse1,
                     double *lseb, void *stream);

/* Packed MCA statistics (float64): [ G1 = Q1^T.Y (H*D) | Wp_multi (H*D) | Wq_multi (H*D) |
 * q1sum (H) | scalars: sum E|s| (pi), sum_nk q e (sigma), sum lse1 (Q), kept count ].
 * Wp = G1 * W^2 + Wp_multi, Wq = q1sum (x) 1 * W^2 + Wq_multi (mca_et.py:293-321).
 * Those 3*H*D + H + 4 entries are the result (and what a multi-GPU job all-reduces); pm_mca_stats_len also
 * counts a scratch tail of 7 * 2*H*D entries: the kernels accumulate [Wp_multi | Wq_multi] once per XCD (one L2
 * each) and fold the copies into the documented slot, clearing the tail, before they return. */
#define PM_MCA_NSCALARS 4
int64_t pm_mca_stats_len(int64_t H, int64_t D);

/* Per-datapoint part of M_step, mca_et.py:236-327: q = exp(beta*logpj - lseb) for datapoints
 * with lseb[n] >= lse_cut; writes the singleton weights q1 (N,H) (zero rows for cut datapoints;
 * G1 = q1^T.Y is then one pm_gemm_tn_acc_f64 into stats[0..H*D)), scatters the multi-cause
 * terms Aid (mca_et.py:309) into Wp_multi / Wq_multi, accumulates the scalars.  W_new = Wp/Wq is
 * an element-wise ratio, so only weights that underflow to 0 (as in the reference) are dropped.
 * Wrm1 = W^(rho-1) (H,D).  `stats` is zeroed by the caller once per EM step.  Hprime <= 12; any D
 * (the observed dimensions are walked in slabs of 512, one launch each). */
int pm_mca_mstep_rows_f64(const double *logpj, int64_t ldl, const double *lse1, const double *lseb,
                          double lse_cut, const double *Y, int64_t ldy, const double *Wrho,
                          const double *Wrm1, const int32_t *cand, const uint16_t *state_masks, int64_t S,
                          const pm_mca_params *params_host, int64_t N, int64_t H, int64_t D,
                          int64_t Hprime, double *q1, int64_t ldq, double *stats, void *stream);

/* E_step and the per-datapoint part of M_step in one pass (no data truncation: every datapoint is
 * kept): outputs of pm_mca_estep_f64 plus q1 / stats of pm_mca_mstep_rows_f64, with every multi-cause
 * power evaluated once instead of twice (mca_et.py:114-179 + 236-327, mmca_et.py:126-199 + 274-347).
 * D <= 512, Hprime <= 12.  `stats` is accumulated into (caller zeroes it). */
int pm_mca_estep_mstats_f64(const double *scores, int64_t lds, const double *wnorm2, const double *ynorm2,
                            const double *Y, int64_t ldy, const double *Wrho, const double *Wrm1,
                            const int32_t *cand, const uint16_t *state_masks, int64_t S,
                            const pm_mca_params *params_host, int64_t N, int64_t H, int64_t D,
                            int64_t Hprime, double *logpj, int64_t ldl, double *lse1, double *lseb,
                            double *q1, int64_t ldq, double *stats, void *stream);
/* DEFERRED statistics for a data-truncation step (mca_et.py:248-262 keeps the N_use datapoints with the largest
 * log-denominator of the annealed weights; which ones is known only after every rank's E-step).
 * pm_mca_estep_mstats_defer_f64 is pm_mca_estep_mstats_f64 that, given `defer_rec` (N x Hprime x D doubles) and
 * `defer_sc` (N x 4), accumulates NOTHING into `stats`: per datapoint it leaves the Aid block of mca_et.py:309 (row j = its
 * j-th candidate: what it would have added to Wq, and times y to Wp) in defer_rec, [sum q |s|, sum q e, log sum exp(logpj), -]
 * in defer_sc, and the singleton posteriors in q1 as always.  Both NULL: exactly pm_mca_estep_mstats_f64.
 * pm_mca_defer_apply_f64 then adds the records of the datapoints with lseb[n] >= *cut (a DEVICE double: the radix
 * select's result never visits the host) into `stats` -- what the in-pass statistics of the kept datapoints would have been
 * (mca_et.py:274-327) -- and zeroes the q1 rows of the others, so that the G1 = q1^T Y product behind it
 * (pm_gemm_tn_acc_f64) sees kept datapoints only.  H, D <= 512, Hprime <= 12. */
int pm_mca_estep_mstats_defer_f64(const double *scores, int64_t lds, const double *wnorm2, const double *ynorm2,
                                  const double *Y, int64_t ldy, const double *Wrho, const double *Wrm1,
                                  const int32_t *cand, const uint16_t *state_masks, int64_t S,
                                  const pm_mca_params *params_host, int64_t N, int64_t H, int64_t D, int64_t Hprime,
                                  double *logpj, int64_t ldl, double *lse1, double *lseb, double *q1, int64_t ldq,
                                  double *stats, double *defer_rec, double *defer_sc, void *stream);
int64_t pm_mca_defer_apply_work_len(int64_t H, int64_t D);      /* doubles of `work` (per-group partial sums, no atomics) */
int pm_mca_defer_apply_f64(const double *lseb, const double *cut, const double *Y, int64_t ldy, const int32_t *cand,
                           const double *records, const double *scalars, double *q1, int64_t ldq, double *stats,
                           double *work, int64_t N, int64_t H, int64_t D, int64_t Hprime, void *stream);
/* The per-step tables of the MCA / MMCA kernels from W^T (H x D, already clamped by check_params): tabs = [ W^T | sign(W) |W|^rho |
 * |W|^(rho-1) ] (three H x D planes; mca_et.py:218-227, mmca_et.py:250-260 compute them with NumPy on the host) and
 * wnorm2[h] = |W_h|^2.  The caller keeps the reference's assertions (finite logarithms, W^rho > 1e-86) on its host copy. */
int pm_mca_tables_f64(const double *wt, int64_t H, int64_t D, double rho, double *tabs, double *wnorm2, void *stream);
/* The element-wise W update of MCA_ET.M_step (mca_et.py:333-348) from the all-reduced `stats`
 * [G1 (H,D) | Wp_m (H,D) | Wq_m (H,D) | q1sum (H) | ...] and the current W^T (H,D): wt_new = (G1 W^2 + Wp_m) / (q1sum W^2 + Wq_m),
 * 0 / tiny where the denominator is below the smallest normal double; wt_clamped (or NULL) = max(wt_new, w_tol), what
 * check_params (mca_et.py:44-55) makes of it at the top of the next step. */
int pm_mca_w_update_f64(const double *stats, const double *wt, int64_t H, int64_t D, double w_tol, double *wt_new,
                        double *wt_clamped, void *stream);

/* ---------------------------------------------------------------------------------------
 * Discrete Sparse Coding (prosper/em/camodels/dsc_et.py, DSC_ET): K-ary latents
 * ------------------------------------------------------------------------------------- */
#define PM_DSC_MAX_K 8      /* latent values incl. the zero value */

/* Scalars of one DSC step (dsc_et.py:376-388, 533-558). */
typedef struct pm_dsc_params {
    int32_t K;                       /* number of latent values                                   */
    int32_t K0;                      /* index of the value 0 (dsc_et.py:153)                      */
    double values[PM_DSC_MAX_K];     /* the latent values `states`                                */
    double logpi[PM_DSC_MAX_K];      /* log pi_k (selection only)                                 */
    double pre1;                     /* -1 / (2 sigma^2) (selection only)                         */
    double ecoef;                    /* beta * pre1: logpj = ecoef * e + pscale * prior           */
    double pscale;                   /* beta if anneal['anneal_prior'] else 1 (dsc_et.py:579-584)  */
    int32_t flags;                   /* PM_DSC_TABLE_ONLY | PM_DSC_LAST_POSITION (Ternary Sparse Coding) */
    int32_t reserved;
} pm_dsc_params;

/* select_Hprimes of DSC (dsc_et.py:347-410; `params_host` given) or TSC (tsc_et.py:142-213; `params_host` NULL) in ONE pass
 * over the scores: the ranking values of pm_dsc_select_scores_f64 / pm_tsc_select_scores_f64 are formed in registers and
 * ranked there (same values, same tie rules as the two-launch form); TSC candidates come out as latents (state % H).
 * Where pm_xsc_select_supported(H, Hprime, tsc) (TSC: H in {16, 32, 64, 128, 256}), else PM_ERANGE. */
int pm_xsc_select_supported(int64_t H, int64_t Hprime, int tsc);
int pm_xsc_select_f64(const double *scores, int64_t lds, const double *gram, const pm_dsc_params *params_host, int64_t N,
                      int64_t H, int64_t Hprime, int32_t *cand, void *stream);

/* Ternary Sparse Coding (prosper/em/camodels/tsc_et.py) runs on the DSC kernels with two flags:
 * PM_DSC_TABLE_ONLY     logpj has one column per row of the state table and nothing else (tsc_et.py:340-349:
 *                       the table holds the null and one-cause states too; no global singleton block);
 * PM_DSC_LAST_POSITION  a latent that occurs twice among a datapoint's candidates contributes to Wp / Wq
 *                       only through its LAST position, as NumPy's fancy-index `+=` does (tsc_et.py:471-475). */
#define PM_DSC_TABLE_ONLY 1
#define PM_DSC_LAST_POSITION 2

/* select_Hprimes of TSC, tsc_et.py:142-213: R (N, 2H) = minus the squared distance of every one-cause
 * state up to |y|^2 -- R[n,h] = -(G_hh + 2 scores[n,h]) (value -1), R[n,H+h] = -(G_hh - 2 scores[n,h])
 * (value +1).  The candidates are the latents (index mod H) of the Hprime largest entries, best last:
 * pm_bsc_select_estep_f64(mode = 1|8) on R with 2H columns. */
int pm_tsc_select_scores_f64(const double *scores, int64_t lds, const double *gram, int64_t N, int64_t H,
                             double *R, int64_t ldr, void *stream);

/* select_Hprimes, dsc_et.py:347-410: R[n,h] = -max_{k != K0} (pre1 (v_k^2 G_hh - 2 v_k scores[n,h]) + log pi_k),
 * the negated best singleton log-joint of latent h up to per-datapoint constants.  The candidates are
 * the Hprime smallest R, best first: pm_bsc_select_estep_f64(mode = 1|4|8) on R. */
int pm_dsc_select_scores_f64(const double *scores, int64_t lds, const double *gram,
                             const pm_dsc_params *params_host, int64_t N, int64_t H, double *R, int64_t ldr,
                             void *stream);

/* E_step, dsc_et.py:492-585.  state_idx (S,Hprime) uint8: index into `values` of every entry of the
 * multi-cause state matrix (itertools.product order, dsc_et.py:56-63); prior = pre_F (1 + (K-1)H + S)
 * (dsc_et.py:539-558).  Columns of logpj: [null | singletons by non-zero value then latent | states].
 * Also writes lse = log sum_k exp(logpj). */
int pm_dsc_estep_f64(const double *scores, int64_t lds, const double *gram, const double *ynorm2,
                     const int32_t *cand, const uint8_t *state_idx, int64_t S, const double *prior,
                     const pm_dsc_params *params_host, int64_t N, int64_t H, int64_t Hprime, double *logpj,
                     int64_t ldl, double *lse, void *stream);

/* Packed DSC statistics (float64): [ Wp = E[s]^T.Y (H*D, filled by pm_gemm_tn_acc_f64) |
 * Wq upper triangle, multi-cause part (H*H) | Wq diagonal, singleton part (H) |
 * expected counts of every non-zero value (PM_DSC_MAX_K, entry K0 unused) |
 * sum_nk q e, sum lse, kept datapoints, rows whose non-zero list overflowed (pm_dsc_mstep_rows_nz_f64) ]. */
int64_t pm_dsc_stats_len(int64_t H, int64_t D);

/* Per-datapoint part of M_step, dsc_et.py:660-735, for datapoints with lse[n] > lse_cut (strict,
 * dsc_et.py:832): writes E[s] rows into expect (N,H) (zero rows for cut datapoints) and accumulates
 * the statistics above into `stats` (caller zeroes it once per EM step). */
int pm_dsc_mstep_rows_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                          const int32_t *cand, const uint8_t *state_idx, int64_t S, const double *prior,
                          const pm_dsc_params *params_host, int64_t N, int64_t H, int64_t D, int64_t Hprime,
                          double *expect, int64_t lde, double *stats, void *stream);
/* The same pass that also leaves every E[s] row as a list of its non-zeros (format of pm_bsc_estep_fused8_nz_f64:
 * N x PM_BSC_NZ_MAX indices / values, unused index slots 0xFFFF; a longer row counts in the last scalar of `stats`), for
 * pm_wp_sparse_f64.  Only where pm_dsc_rows16_supported(...) holds (the sixteen-lanes-per-datapoint kernel). */
int pm_dsc_rows16_supported(int64_t H, int64_t Hprime, int64_t S, int64_t K, int flags);
int pm_dsc_mstep_rows_nz_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                          const int32_t *cand, const uint8_t *state_idx, int64_t S, const double *prior,
                          const pm_dsc_params *params_host, int64_t N, int64_t H, int64_t D, int64_t Hprime,
                          double *expect, int64_t lde, double *stats, uint16_t *nz_idx, double *nz_val,
                             void *stream);
/* The same pass with the data-truncation cut read on the DEVICE (round 6): `cut_dev` (one double, e.g. what pm_kth_final_f64
 * left, adjusted as dsc_et.py:832 / tsc_et.py:433-440 ask) replaces `lse_cut` when it is not NULL -- the M-step of a
 * truncation step then has no host round trip between the radix select and its row pass.  nz_idx / nz_val may be NULL
 * (then exactly pm_dsc_mstep_rows_f64). */
int pm_dsc_mstep_rows_cutp_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut, const double *cut_dev,
                               const int32_t *cand, const uint8_t *state_idx, int64_t S, const double *prior,
                               const pm_dsc_params *params_host, int64_t N, int64_t H, int64_t D, int64_t Hprime,
                               double *expect, int64_t lde, double *stats, uint16_t *nz_idx, double *nz_val, void *stream);

/* pm_dsc_estep_f64 that ALSO produces the M-step's row statistics -- what pm_dsc_mstep_rows_nz_f64 computes from the stored
 * log-joints (dsc_et.py:587-774: E[s] rows, their non-zero lists, the candidates' second moments -> Wq, qdiag, the value
 * counts, the sigma / likelihood scalars) -- from the exponentials its log-sum-exp evaluates anyway: no second pass over the
 * log-joints.  For the E-step of an EM iteration with no data truncation ahead (every datapoint kept); `stats` is
 * accumulated into (caller zeroes it), `expect` / `nz_*` as in pm_dsc_mstep_rows_nz_f64.  Sixteen lanes per datapoint:
 * where pm_dsc_estep_mstats_supported(H, Hprime, S, K, flags) holds (else PM_ERANGE: run the two passes). */
int pm_dsc_estep_mstats_supported(int64_t H, int64_t Hprime, int64_t S, int64_t K, int flags);
int pm_dsc_estep_mstats_f64(const double *scores, int64_t lds, const double *gram, const double *ynorm2, const int32_t *cand,
                            const uint8_t *state_idx, int64_t S, const double *prior, const pm_dsc_params *params_host,
                            int64_t N, int64_t H, int64_t D, int64_t Hprime, double *logpj, int64_t ldl, double *lse,
                            double *expect, int64_t lde, double *stats, uint16_t *nz_idx, double *nz_val, void *stream);

/* ---------------------------------------------------------------------------------------
 * Gaussian (spike-and-slab) Sparse Coding, scalar noise (prosper/em/camodels/gsc_et.py, GSC)
 * ------------------------------------------------------------------------------------- */

/* Shapes the GSC kernel covers: H <= 512, gamma <= 8 (g x g systems solved in registers; instantiated for 2, 3, 4, 6, 8). */
int pm_gsc_supported(int64_t H, int64_t Hprime, int64_t gamma);

/* stats (float64): [ sum_n xpt_ss, STRICT upper triangle, multi-cause part (H*H; its diagonal is sum_n xpt_s and is not formed here) |
 *                    sum_n xpt_szsz, both triangles as they are, multi-cause part (H*H) |
 *                    sum_n xpt_s (H) | sum_n xpt_sz (H) | diagonal of sum_n xpt_szsz that is not in the block above (H): the
 *                    singletons', and -- where the kernel keeps its column sums in LDS -- the multi-cause states' too |
 *                    scratch: seven more copies of the first 2*H*H entries ]
 * diag(sum xpt_ss) = sum_n xpt_s (s_h^2 = s_h).  The kernel accumulates the (H,H) blocks per XCD (one L2 each)
 * and folds the copies into the first 2*H*H entries (clearing the scratch) before it returns: the caller zeroes the
 * whole buffer once; the documented entries accumulate across calls as before. */
int64_t pm_gsc_stats_len(int64_t H);

/* The statistics pm_gsc_estep_f64 leaves in `stats` -> [sum xpt_ss (H,H) | sum xpt_szsz (H,H) | sum xpt_s (H) | sum xpt_sz (H)
 * | sum |y|^2] as the M-step all-reduces them (gsc_et.py:603-610, 662-671): xpt_ss mirrored from its upper triangle with
 * sum xpt_s on the diagonal, xpt_szsz as accumulated (it is not symmetric once psi_sq is not) plus the singletons' diagonal.
 * `sum_ynorm2`: device pointer to sum_n |y_n|^2 of the shard.  One launch. */
int pm_gsc_pack_stats_f64(const double *stats, int64_t H, const double *sum_ynorm2, double *out, void *stream);

/* The H- and H x H-sized tail of GSC's M-step for scalar sigma_sq (gsc_et.py:640-713) and the tables of the next
 * E-step, on the device (one workgroup): pi clip, mu, psi_sq (with (sum_ss + eps I)^-1 from pm_spd_inverse_batch_f64),
 * sigma_sq = (sum |y|^2 - trace(xsz_xsz . gram)) / N / D + eps, gram = W_new^T W_new.  `old` / `params`:
 * [pi (H) | mu (H) | psi_sq (H*H) | sigma_sq (1)]; `learn` bits 1 pi, 2 mu, 4 psi_sq, 8 sigma_sq (others copied from `old`).
 * `tables`: 9 x H doubles -- the eight rows pm_gsc_estep_f64 takes and a ninth holding 1 / sigma_sq, which that
 * function reads when it is called with sigma_sq == 0. */
int pm_gsc_mstep_finish_f64(const double *xs_xsz, const double *xsz_xsz, const double *sum_ss, const double *sum_zz,
                            const double *ss_inv, const double *sum_s, const double *sum_sz, const double *sum_yy,
                            const double *gram, const double *old, double N, int64_t D, int64_t H, int learn,
                            double *params, double *tables, void *stream);

/* select_Hprimes + E_step of GSC in one pass (gsc_et.py:721-809, 401-580, 260-398):
 *   scores (N,H) = Y.W;  gram = W^T.W (H,H);  psi_sq (H,H);
 *   tables (8*H): per-latent constants [c0 | c1 | gm | il | kl | ilam | mu | lpi] with
 *     lam = G_hh/s2 + 1/psi_hh, c0 = -(log psi_hh + log lam) - mu^2 G_hh/s2, c1 = 2 mu/s2,
 *     gm = G_hh mu, il = 1/(lam s2^2), kl = 1/(lam s2), ilam = 1/lam, lpi = log(pi/(1-pi))
 *   do_select != 0: candidates = the Hprime best component scores, sorted by latent index,
 *     written to `cand`; otherwise `cand` (sorted) is an input.
 *   sigma_sq == 0: `tables` has a ninth row whose first entry is 1/sigma_sq (left on the device by
 *     pm_gsc_mstep_finish_f64).
 * Outputs: xpt_s, xpt_sz (N,H) and the sums over datapoints accumulated into `stats`
 * (zeroed by the caller).  The (N,H,H) moments of the reference are never materialised. */
int pm_gsc_estep_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                     const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                     int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                     int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx,
                     double *stats, void *stream);

/* The same pass, also splitting the rows of xpt_sz for the M-step's contraction over the datapoints
 * [Y | xpt_s | xpt_sz]^T . xpt_sz (gsc_et.py:592-625: my_Wp, the two H x H moment products): a row with at most
 * PM_BSC_NZ_MAX entries above the threshold tables[8 H + 1] leaves them as a list (nz_idx / nz_val, N x PM_BSC_NZ_MAX,
 * format of pm_bsc_estep_fused8_nz_f64; for pm_wp_sparse_t_f64), any other row an empty list and its index in
 * dense_rows[0 .. *dense_count) (for pm_gemm_tn_acc_rows_f64; *dense_count = 0 at launch; order as the workgroups finish).
 * nz_val holds TWO planes of N x PM_BSC_NZ_MAX doubles: xpt_sz at the listed entries, then xpt_s at them (pm_gsc_list_pairs_f64).
 * The threshold is written by pm_gsc_mstep_finish_f64 (2^-57 / N of the smallest |column sum| of xpt_sz over all ranks: what
 * the lists drop is below the rounding of the sums); 0 keeps every row dense.  Where pm_gsc_lists_supported(H, Hprime,
 * gamma, D), else PM_ERANGE. */
int pm_gsc_lists_supported(int64_t H, int64_t Hprime, int64_t gamma, int64_t D);
int pm_gsc_estep_lists_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                           const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                           int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                           int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx, double *stats,
                           uint16_t *nz_idx, double *nz_val, int32_t *dense_rows, int32_t *dense_count, void *stream);

/* xs^T xsz and xsz^T xsz of GSC's M-step (gsc_et.py:603-610) over the LISTED datapoints from the lists alone (outer products of
 * a datapoint's listed entries): out[0 .. H*H) += sum_n xs_n xsz_n^T, out[H*H .. 2 H*H) += sum_n xsz_n xsz_n^T over the entries
 * nz_idx lists.  nz_val_s / nz_val: xpt_s / xpt_sz at those entries -- pm_gsc_estep_lists_f64 leaves them as the second and the
 * first plane of its `nz_val` (which therefore holds 2 x N x PM_BSC_NZ_MAX doubles).  The sparse product then streams the D
 * columns of Y only.  H in {64, 128, 192, 256}, else PM_ERANGE. */
int pm_gsc_list_pairs_f64(const uint16_t *nz_idx, const double *nz_val_s, const double *nz_val, int64_t N, int64_t H,
                          double *out, void *stream);

/* The same pass, also writing every state's log-joint -- what GSC.compute_lpj returns (gsc_et.py:811-944): no
 * annealing, prior odds included -- to logpj (N, ldl >= 1 + H + S): [null state ; singletons h = 0..H-1 ; multi-cause
 * states in state_masks order], rows in datapoint order (the reference sorts its cluster order back, :942-944). */
int pm_gsc_estep_lpj_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                         const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                         int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                         int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx,
                         double *stats, double *logpj, int64_t ldl, void *stream);

/* ... and every datapoint's un-normalised sums over the multi-cause states -- what GSC.compute_posterior_hprime returns for a
 * data cluster (gsc_et.py:260-398) -- blocks (N, ldb >= 2 Hprime^2 + 2 Hprime + 1):
 * [sum_s p_s [i, k in s] | sum_s p_s (kappa kappa^T + Lambda^-1)_ik | sum_s p_s [i in s] | sum_s p_s kappa_i | sum_s p_s] over the
 * candidate positions i, k in the order of `cand`, p_s = exp(beta lp_s) with the reference's clamp to `tiny`. */
int pm_gsc_estep_lpj_blocks_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                                const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                                int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                                int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx, double *stats,
                                double *logpj, int64_t ldl, double *blocks, int64_t ldb, void *stream);

/* GSC.component_scores (gsc_et.py:752-809): singleton log-posteriors without the prior, clamped as the reference
 * clamps them (NaN / below -DBL_MAX -> -DBL_MAX, +-inf -> 0), out (N, ldo >= H).  `scores`, `ynorm2`, `tables`
 * (rows c0, c1, gm, il are read), `sigma_sq` (> 0) as for pm_gsc_estep_f64. */
int pm_gsc_component_scores_f64(const double *scores, int64_t lds, const double *ynorm2, const double *tables,
                                double sigma_sq, int64_t N, int64_t H, double *out, int64_t ldo, void *stream);

/* ---------------------------------------------------------------------------------------
 * CAModel.inference (prosper/em/camodels/__init__.py:256-375), after compute_lpj
 * ------------------------------------------------------------------------------------- */

/* Per datapoint, from its K = 1 + H + S log-joints (logpj (N, ldl), columns [null ; singletons ; multi-cause states over
 * `cand` (N, Hprime) in state_masks order]): the topK columns of the normalised posterior, descending, ties larger column
 * first (argsort()[..., ::-1], :302) -> top_idx (N, topK) int32, their normalised log-probabilities top_lpc and their
 * log-joints relative to the row maximum top_rel (what the reference reports for logprob=False, :309-312), and the
 * log-marginals log p(s_h = 1 | y) -> marg (N, ldm >= H) (:321-327). */
int pm_infer_topk_f64(const double *logpj, int64_t ldl, const int32_t *cand, const uint16_t *state_masks, int64_t N,
                      int64_t H, int64_t Hprime, int64_t S, int64_t topK, int32_t *top_idx, double *top_lpc,
                      double *top_rel, double *marg, int64_t ldm, void *stream);
/* The same for logpj rows with `single_cols` >= H one-cause columns in front of the multi-cause ones -- DSC's K-ary latents
 * (dsc_et.py:927-1059): [null ; (K - 1) H one-cause states, the first non-zero value's block first ; S multi-cause states].
 * `state_masks` then marks the positions at which a state's latent takes the value 1 (the reference's marginal,
 * dsc_et.py:1010-1016: the first value's one-cause state + those multi-cause states); top-K runs over all columns. */
int pm_infer_topk_cols_f64(const double *logpj, int64_t ldl, const int32_t *cand, const uint16_t *state_masks, int64_t N,
                           int64_t H, int64_t Hprime, int64_t S, int64_t single_cols, int64_t topK, int32_t *top_idx,
                           double *top_lpc, double *top_rel, double *marg, int64_t ldm, void *stream);

/* TSC_ET.inference after compute_lpj (tsc_et.py:546-680): logpj (N, ldl >= S) holds ONE log-joint per row of the ternary state
 * table, state_vals (S, Hprime) int8 the latent values -1 / 0 / +1 per candidate position, cand (N, Hprime) may repeat a latent.
 * rank != 0: top_idx (N, topK) = the best states descending by (value, column), and tie[n] = 1 when two of row n's topK + 1
 * best are exactly equal -- the order NumPy's argsort()[::-1] (:626) gives exact ties is not a function of (value, column),
 * so the caller ranks those rows with NumPy and calls again with rank == 0, top_idx given.  Either way: top_lpc / top_post
 * (N, topK) normalised log-probabilities / probabilities; m_out / am_out (N, H; am_out may be NULL) the signed / absolute
 * marginals sum_s p_s v and sum_s p_s |v| written at the candidates in position order (a repeated candidate's last position
 * wins, :640-655); s_out (N, topK, H) int8 the top states written likewise.  Entries of other latents are left as they are. */
int pm_infer_topk_signed_f64(const double *logpj, int64_t ldl, const int32_t *cand, const int8_t *state_vals, int64_t N,
                             int64_t H, int64_t Hprime, int64_t S, int64_t topK, int rank, int32_t *top_idx, double *top_lpc,
                             double *top_post, int32_t *tie, int8_t *s_out, double *m_out, double *am_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PROSPER_HIP_H */
