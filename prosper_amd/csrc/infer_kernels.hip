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

}  // namespace

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
