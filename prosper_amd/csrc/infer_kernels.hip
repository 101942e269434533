// CAModel.inference (prosper/em/camodels/__init__.py:256-375), the per-datapoint part after compute_lpj: the top-K states of
// the normalised posterior and the marginals p(s_h = 1 | y) over the truncated state set.
//
// One wavefront per datapoint, lanes strided over the K = 1 + H + S log-joints of its row (a few KB: the passes below
// re-read it from L1 / L2):
//   pass 1   row maximum and log-sum-exp                                            (:297-301)
//   top-K    round r takes the largest entry strictly below round r - 1's winner in the order (value, index) -- the
//            reference sorts with argsort()[..., ::-1] (:302), so equal values come out larger index first; no entry is
//            marked or copied
//   marginals  log p(s_h = 1 | y) = logpjc[1 + h] for a latent outside the candidates, else the log-sum-exp of its
//            singleton and of the multi-cause states that contain its candidate position                   (:321-327)
// The top-K state vectors (res['s']) are assembled by the caller from the returned column indices: that part is index
// bookkeeping with the reference's quirks (entries of earlier adaptive rounds are never cleared, :313-319).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

__global__ __launch_bounds__(256) void infer_topk_kernel(const double *__restrict__ logpj, int64_t ldl,
                                                          const int32_t *__restrict__ cand,
                                                          const uint16_t *__restrict__ masks, int64_t N, int H, int Hp,
                                                          int S, int single_cols, int topK, int32_t *__restrict__ top_idx,
                                                          double *__restrict__ top_lpc, double *__restrict__ top_rel,
                                                          double *__restrict__ marg, int64_t ldm) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    // columns: [null ; single_cols one-cause states (binary latents: H; K-ary: (K - 1) H, the first value's block first) ;
    // S multi-cause states]
    const int K = 1 + single_cols + S, moff = 1 + single_cols;
    for (int64_t n = wave0; n < N; n += nwaves) {
        const double *row = logpj + n * ldl;
        // ---- maximum, log-sum-exp
        double mx = -INFINITY;
        for (int k = lane; k < K; k += 64) mx = fmax(mx, row[k]);
        mx = pm_wave_max(mx);
        double sum = 0.0;
        for (int k = lane; k < K; k += 64) sum += exp(row[k] - mx);
        sum = pm_wave_sum(sum);
        const double lse = mx + log(sum);
        // ---- top-K, descending by (value, index)
        double pv = INFINITY;
        int pi = 0x7FFFFFFF;
        for (int r = 0; r < topK; ++r) {
            double bv = -INFINITY;
            int bi = -1;
            for (int k = lane; k < K; k += 64) {
                const double v = row[k];
                const bool below = (v < pv) || (v == pv && k < pi);        // not yet taken
                const bool better = (v > bv) || (v == bv && k > bi);
                if (below && better && v == v) {
                    bv = v;
                    bi = k;
                }
            }
            pm_wave_argmax(bv, bi);
            if (lane == 0) {
                top_idx[n * topK + r] = bi;                               // -1: fewer than topK (finite or -inf) entries
                top_lpc[n * topK + r] = bi >= 0 ? bv - lse : -INFINITY;
                top_rel[n * topK + r] = bi >= 0 ? bv - mx : -INFINITY;
            }
            pv = bv;
            pi = bi;
            if (bi < 0) pv = -INFINITY, pi = -1;
        }
        // ---- marginals
        double *mrow = marg + n * ldm;
        for (int h = lane; h < H; h += 64) mrow[h] = row[1 + h] - lse;
        // (the stores above have completed before the candidates' entries are overwritten below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int j = 0; j < Hp; ++j) {
            const int c = cand[n * Hp + j];
            const double single = row[1 + c];
            double m = -INFINITY;
            for (int s = lane; s < S; s += 64)
                if ((masks[s] >> j) & 1) m = fmax(m, row[moff + s]);
            m = fmax(pm_wave_max(m), single);
            double acc = 0.0;
            for (int s = lane; s < S; s += 64)
                if ((masks[s] >> j) & 1) acc += exp(row[moff + s] - m);
            acc = pm_wave_sum(acc) + exp(single - m);
            if (lane == 0) mrow[c] = (m == -INFINITY) ? -INFINITY : (m + log(acc)) - lse;
        }
    }
}


// TSC_ET.inference (prosper/em/camodels/tsc_et.py:546-680) after compute_lpj: one log-joint per row of the ternary state table
// (no null / one-cause prefix), latent values -1 / 0 / +1 per candidate POSITION (state_vals, S x Hp), candidates with
// possible repeats.  One wavefront per datapoint:
//   top-K states of the normalised posterior, descending by (value, column) -- and a FLAG when two of the K + 1 best are exactly
//   equal: states that differ only in which position of a repeated candidate carries the value tie exactly, and the order
//   NumPy's argsort()[::-1] (:626, an introsort) leaves them in is not a function of (value, column); the caller re-ranks the
//   flagged rows with NumPy itself and calls again with rank == 0 (top_idx given);
//   signed and absolute marginals per position, sum_s p_s v_sj and sum_s p_s |v_sj| (:640-655);
//   the writes into s (N, topK, H) / m / am (N, H) in position order, so that a repeated candidate's LAST position wins, as
//   upstream; entries of latents outside the candidates are left as they are (upstream never clears a re-run datapoint's).
__global__ __launch_bounds__(256) void infer_topk_signed_kernel(const double *__restrict__ logpj, int64_t ldl,
                                                                 const int32_t *__restrict__ cand,
                                                                 const int8_t *__restrict__ vals, int64_t N, int H, int Hp,
                                                                 int S, int topK, int rank, int32_t *__restrict__ top_idx,
                                                                 double *__restrict__ top_lpc, double *__restrict__ top_post,
                                                                 int32_t *__restrict__ tie, int8_t *__restrict__ s_out,
                                                                 double *__restrict__ m_out, double *__restrict__ am_out,
                                                                 int write_am) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t n = wave0; n < N; n += nwaves) {
        const double *row = logpj + n * ldl;
        double mx = -INFINITY;
        for (int k = lane; k < S; k += 64) mx = fmax(mx, row[k]);
        mx = pm_wave_max(mx);
        double sum = 0.0;
        for (int k = lane; k < S; k += 64) sum += exp(row[k] - mx);
        sum = pm_wave_sum(sum);
        const double lse = mx + log(sum);
        if (rank) {
            double pv = INFINITY;
            int pi = 0x7FFFFFFF, tied = 0;
            for (int r = 0; r <= topK && r < S; ++r) {          // (one round past topK: a tie across the cut counts too)
                double bv = -INFINITY;
                int bi = -1;
                for (int k = lane; k < S; k += 64) {
                    const double v = row[k];
                    const bool below = (v < pv) || (v == pv && k < pi);
                    const bool better = (v > bv) || (v == bv && k > bi);
                    if (below && better && v == v) {
                        bv = v;
                        bi = k;
                    }
                }
                pm_wave_argmax(bv, bi);
                // (a tie in what upstream RANKS -- the normalised lpc = logpj - lse, tsc_et.py:620-626: two distinct log-joints can
                // round to the same lpc, and argsort's order among equals is not a function of (value, index): those rows go to
                // NumPy as the exactly tied ones do; round-5 advisor finding)
                if (r > 0 && bi >= 0 && (bv == pv || (bv - lse) == (pv - lse))) tied = 1;
                if (r < topK && lane == 0) top_idx[n * topK + r] = bi;
                pv = bv;
                pi = bi;
                if (bi < 0) pv = -INFINITY, pi = -1;
            }
            if (lane == 0) tie[n] = tied;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (top_idx written above is read below)
        for (int r = lane; r < topK; r += 64) {
            const int k = top_idx[n * topK + r];
            const double v = (k >= 0 && k < S) ? row[k] : -INFINITY;
            top_lpc[n * topK + r] = v - lse;
            top_post[n * topK + r] = exp(v - lse);
        }
        // marginals per position, then the writes in position order (lane 0: the order is the contract)
        for (int j = 0; j < Hp; ++j) {
            double a = 0.0, b = 0.0;
            for (int k = lane; k < S; k += 64) {
                const double pk = exp(row[k] - lse);
                const int v = vals[(int64_t)k * Hp + j];
                a += pk * (double)v;
                b += pk * (double)(v < 0 ? -v : v);
            }
            a = pm_wave_sum(a);
            b = pm_wave_sum(b);
            if (lane == 0) {
                const int c = cand[n * Hp + j];
                m_out[n * H + c] = a;
                if (write_am) am_out[n * H + c] = b;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // position j + 1 may hit the same latent
        }
        for (int r = lane; r < topK; r += 64) {
            const int k = top_idx[n * topK + r];
            if (k < 0 || k >= S) continue;
            int8_t *srow = s_out + (n * topK + r) * H;
            for (int j = 0; j < Hp; ++j) srow[cand[n * Hp + j]] = vals[(int64_t)k * Hp + j];   // (a lane's own stores: in order)
        }
    }
}

}  // namespace

extern "C" int pm_infer_topk_signed_f64(const double *logpj, int64_t ldl, const int32_t *cand, const int8_t *state_vals,
                                        int64_t N, int64_t H, int64_t Hprime, int64_t S, int64_t topK, int rank,
                                        int32_t *top_idx, double *top_lpc, double *top_post, int32_t *tie, int8_t *s_out,
                                        double *m_out, double *am_out, void *stream) {
    if (N == 0) return PM_OK;
    if (!logpj || !cand || !state_vals || !top_idx || !top_lpc || !top_post || !s_out || !m_out || N < 0 || H <= 0 ||
        Hprime <= 0 || S <= 0 || topK <= 0 || ldl < S || (rank && !tie))
        return PM_EINVAL;
    if (Hprime > PM_MAX_HPRIME || topK > S) return PM_ERANGE;
    int64_t blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(infer_topk_signed_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), logpj,
                       ldl, cand, state_vals, N, (int)H, (int)Hprime, (int)S, (int)topK, rank, top_idx, top_lpc, top_post, tie,
                       s_out, m_out, am_out ? am_out : m_out, am_out ? 1 : 0);
    return (int)hipGetLastError();
}

extern "C" int pm_infer_topk_f64(const double *logpj, int64_t ldl, const int32_t *cand, const uint16_t *state_masks,
                                 int64_t N, int64_t H, int64_t Hprime, int64_t S, int64_t topK, int32_t *top_idx,
                                 double *top_lpc, double *top_rel, double *marg, int64_t ldm, void *stream) {
    return pm_infer_topk_cols_f64(logpj, ldl, cand, state_masks, N, H, Hprime, S, H, topK, top_idx, top_lpc, top_rel, marg,
                                  ldm, stream);
}

extern "C" int pm_infer_topk_cols_f64(const double *logpj, int64_t ldl, const int32_t *cand, const uint16_t *state_masks,
                                      int64_t N, int64_t H, int64_t Hprime, int64_t S, int64_t single_cols, int64_t topK,
                                      int32_t *top_idx, double *top_lpc, double *top_rel, double *marg, int64_t ldm,
                                      void *stream) {
    if (N == 0) return PM_OK;
    if (!logpj || !cand || !top_idx || !top_lpc || !top_rel || !marg || N < 0 || H <= 0 || Hprime <= 0 || S < 0 ||
        single_cols < H || topK <= 0 || ldl < 1 + single_cols + S || ldm < H || (S > 0 && !state_masks))
        return PM_EINVAL;
    if (Hprime > PM_MAX_HPRIME || topK > 1 + single_cols + S || single_cols > INT32_MAX / 2) return PM_ERANGE;
    int64_t blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(infer_topk_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), logpj, ldl,
                       cand, state_masks, N, (int)H, (int)Hprime, (int)S, (int)single_cols, (int)topK, top_idx, top_lpc,
                       top_rel, marg, ldm);
    return (int)hipGetLastError();
}
