// Binary Sparse Coding: candidate selection, truncated E-step and the per-datapoint part of the
// M-step as gfx950 kernels.  One 64-lane wavefront owns one datapoint at a time (workgroup =
// 4 wavefronts, grid-stride over datapoints); a datapoint's H scores / K log-joints are spread
// over the lanes (lane-strided => every row access is a coalesced 512-B segment), reductions
// (top-H', max, log-sum-exp) run across the wavefront, and the state table plus the per-datapoint
// candidate block of the Gram matrix sit in LDS.
//
// Algebra (verified against the reference through oracle/bsc_oracle.py):
//   a_h = <W_h, y>  (scores GEMM),  G = W.W^T,
//   e_0 = |y|^2,  e_h = G_hh - 2 a_h + |y|^2,
//   e_s = |y|^2 - 2 sum_{j in s} a_{c_j} + sum_{j,j' in s} G_{c_j c_j'}
//   logpj = prior_scale * pil_bar * |s| + ecoef * e          (bsc_et.py:187-190)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

constexpr int WAVES = 4;  // wavefronts per workgroup

// LDS traffic between lanes of ONE wavefront: order the accesses, no workgroup barrier needed.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// select_Hprimes (bsc_et.py:98-115)
// ---------------------------------------------------------------------------------------------
template <int VPL>  // latents per lane: H <= 64 * VPL
__global__ __launch_bounds__(256) void bsc_select_kernel(const double *__restrict__ scores, int64_t lds,
                                                          const double *__restrict__ wnorm2, int64_t wstride,
                                                          const double *__restrict__ ynorm2, int64_t N, int H, int Hp,
                                                          int32_t *__restrict__ cand) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WAVES;

    double sw[VPL];
    unsigned invalid = 0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int h = lane + 64 * i;
        sw[i] = (h < H) ? sqrt(wnorm2[(int64_t)h * wstride]) : 1.0;
        if (h >= H) invalid |= 1u << i;
    }

    for (int64_t n = wave0; n < N; n += nwaves) {
        const double sy = sqrt(ynorm2[n]);
        const double *row = scores + n * lds;
        double v[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = lane + 64 * i;
            double x = -INFINITY;
            if (h < H) {
                x = row[h] / sw[i] / sy;   // same operation order as the reference's sim
                if (x != x) x = -INFINITY;  // NaN (zero-norm row) ranks lowest
            }
            v[i] = x;
        }
        unsigned taken = invalid;
        for (int r = 0; r < Hp; ++r) {
            double bv = -INFINITY;
            int bi = -1;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int h = lane + 64 * i;
                const bool free_slot = !((taken >> i) & 1u);
                if (free_slot && (v[i] > bv || (v[i] == bv && h > bi))) {
                    bv = v[i];
                    bi = h;
                }
            }
            pm_wave_argmax(bv, bi);
            if ((bi & 63) == lane) taken |= 1u << (bi >> 6);
            if (lane == 0) cand[n * Hp + (Hp - 1 - r)] = bi;  // ascending: best candidate last
        }
    }
}

// ---------------------------------------------------------------------------------------------
// E_step (bsc_et.py:119-192)
// ---------------------------------------------------------------------------------------------
struct LseAcc {  // per-lane online log-sum-exp
    double m, s;
    __device__ void push(double f) {
        if (f > m) {
            s = s * exp(m - f) + 1.0;
            m = f;
        } else {
            s += exp(f - m);
        }
    }
};

__global__ __launch_bounds__(256) void bsc_estep_kernel(const double *__restrict__ scores, int64_t lds,
                                                         const double *__restrict__ gram,
                                                         const double *__restrict__ ynorm2,
                                                         const double *__restrict__ wmu,
                                                         const double *__restrict__ ymu,
                                                         const int32_t *__restrict__ cand,
                                                         const uint16_t *__restrict__ masks, int S,
                                                         pm_bsc_estep_params P, int64_t N, int H, int Hp,
                                                         double *__restrict__ logpj, int64_t ldl,
                                                         double *__restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ w2 (H) | per wave: ac (16) gc (256) | masks (S) ]
    double *s_w2 = reinterpret_cast<double *>(smem);
    double *s_wave = s_w2 + H;
    uint16_t *s_masks = reinterpret_cast<uint16_t *>(s_wave + WAVES * (16 + 256));

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    for (int h = tid; h < H; h += 256) s_w2[h] = gram[(int64_t)h * H + h] + (wmu ? 2.0 * wmu[h] : 0.0);
    for (int s = tid; s < S; s += 256) s_masks[s] = masks[s];
    __syncthreads();

    double *s_ac = s_wave + wave * (16 + 256);
    double *s_gc = s_ac + 16;
    const double ppil = P.prior_scale * P.pil_bar;
    const int64_t wave0 = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t nwaves = (int64_t)gridDim.x * WAVES;

    for (int64_t n = wave0; n < N; n += nwaves) {
        const double *row = scores + n * lds;
        const int32_t *cn = cand + n * Hp;
        double yn = ynorm2[n];
        if (ymu) yn = yn - 2.0 * ymu[n] + P.mu_sqnorm;

        // candidate scores and the Hp x Hp candidate block of G -> LDS
        if (lane < Hp) {
            const int c = cn[lane];
            s_ac[lane] = row[c] - (wmu ? wmu[c] : 0.0);
        }
        for (int p = lane; p < Hp * Hp; p += 64) {
            const int i = p / Hp, j = p - i * Hp;
            s_gc[p] = gram[(int64_t)cn[i] * H + cn[j]];
        }
        wave_lds_sync();

        double *out = logpj + n * ldl;
        LseAcc acc{-INFINITY, 0.0};

        if (lane == 0) {  // null state
            const double f = P.ecoef * yn;
            out[0] = f;
            acc.push(f);
        }
        for (int h = lane; h < H; h += 64) {  // singleton states
            const double e = s_w2[h] - 2.0 * row[h] + yn;
            const double f = ppil + P.ecoef * e;
            out[1 + h] = f;
            acc.push(f);
        }
        for (int s = lane; s < S; s += 64) {  // multi-cause states over the candidates
            const unsigned mask = s_masks[s];
            double lin = 0.0, quad = 0.0;
            unsigned mi = mask;
            while (mi) {
                const int i = __builtin_ctz(mi);
                mi &= mi - 1;
                lin += s_ac[i];
                quad += s_gc[i * Hp + i];
                unsigned mj = mi;  // j > i: symmetric, counted twice
                double off = 0.0;
                while (mj) {
                    const int j = __builtin_ctz(mj);
                    mj &= mj - 1;
                    off += s_gc[i * Hp + j];
                }
                quad += 2.0 * off;
            }
            const double e = yn - 2.0 * lin + quad;
            const double f = ppil * (double)__builtin_popcount(mask) + P.ecoef * e;
            out[1 + H + s] = f;
            acc.push(f);
        }

        if (lse) {
            const double M = pm_wave_max(acc.m);
            const double part = (acc.s > 0.0) ? acc.s * exp(acc.m - M) : 0.0;
            const double tot = pm_wave_sum(part);
            if (lane == 0) lse[n] = M + log(tot);
        }
        wave_lds_sync();  // s_ac / s_gc are rewritten for the next datapoint
    }
}

// ---------------------------------------------------------------------------------------------
// M_step, per-datapoint part (bsc_et.py:271-272, 334-366, 395-415)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bsc_mstep_rows_kernel(const double *__restrict__ logpj, int64_t ldl,
                                                              const double *__restrict__ lse, double lse_cut,
                                                              const int32_t *__restrict__ cand,
                                                              const uint16_t *__restrict__ masks, int S,
                                                              const int32_t *__restrict__ pair_ptr,
                                                              const uint16_t *__restrict__ pair_states,
                                                              pm_bsc_estep_params P, int64_t N, int H, int D, int Hp,
                                                              double *__restrict__ expect, int64_t lde,
                                                              double *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int npairs = Hp * Hp;
    const int nlist = pair_ptr[npairs];
    // [ qdiag (H) | mus (H) | per wave: es (H) qs (S) | red (3*WAVES) | pair_ptr (npairs+1) | masks (S) | pair_states ]
    double *s_qdiag = reinterpret_cast<double *>(smem);
    double *s_mus = s_qdiag + H;
    double *s_wave = s_mus + H;
    double *s_red = s_wave + WAVES * (H + S);
    int32_t *s_pptr = reinterpret_cast<int32_t *>(s_red + 3 * WAVES);
    uint16_t *s_masks = reinterpret_cast<uint16_t *>(s_pptr + npairs + 1);
    uint16_t *s_plist = s_masks + S;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    for (int h = tid; h < H; h += 256) {
        s_qdiag[h] = 0.0;
        s_mus[h] = 0.0;
    }
    for (int i = tid; i <= npairs; i += 256) s_pptr[i] = pair_ptr[i];
    for (int s = tid; s < S; s += 256) s_masks[s] = masks[s];
    for (int i = tid; i < nlist; i += 256) s_plist[i] = pair_states[i];
    __syncthreads();

    double *s_es = s_wave + wave * (H + S);
    double *s_qs = s_es + H;
    double *Wq = stats + pm_bsc_stats_offset_wq_dev(H, D);
    const double ppil = P.prior_scale * P.pil_bar;
    const double inv_ecoef = 1.0 / P.ecoef;
    const int64_t wave0 = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t nwaves = (int64_t)gridDim.x * WAVES;

    double sig = 0.0, fs = 0.0, cnt = 0.0;  // per-lane partials of the scalar statistics

    for (int64_t n = wave0; n < N; n += nwaves) {
        const double l = lse[n];
        double *erow = expect + n * lde;
        if (!(l >= lse_cut)) {  // truncated datapoint (bsc_et.py:253-256): contributes nothing
            for (int h = lane; h < H; h += 64) erow[h] = 0.0;
            continue;
        }
        const double *f = logpj + n * ldl;
        const int32_t *cn = cand + n * Hp;

        if (lane == 0) {
            const double f0 = f[0];
            sig += exp(f0 - l) * (f0 * inv_ecoef);
            fs += l;
            cnt += 1.0;
        }
        for (int h = lane; h < H; h += 64) {
            const double fh = f[1 + h];
            const double q = exp(fh - l);
            sig += q * ((fh - ppil) * inv_ecoef);
            s_es[h] = q;
            atomicAdd(&s_qdiag[h], PM_Q(q, 0));
        }
        for (int s = lane; s < S; s += 64) {
            const double fv = f[1 + H + s];
            const double q = exp(fv - l);
            sig += q * ((fv - ppil * (double)__builtin_popcount((unsigned)s_masks[s])) * inv_ecoef);
            s_qs[s] = q;
        }
        wave_lds_sync();

        // second moments over candidate positions: m2[i][j] = sum_{s containing i and j} q_s
        for (int p = lane; p < npairs; p += 64) {
            const int i = p / Hp, j = p - i * Hp;
            if (j < i) continue;  // upper triangle only; mirrored on the host
            double m2 = 0.0;
            for (int t = s_pptr[p]; t < s_pptr[p + 1]; ++t) m2 += s_qs[s_plist[t]];
            const int ci = cn[i], cj = cn[j];
            if (i == j) {
                s_es[ci] += m2;  // E[s_c] = q1_c + sum_{s containing c} q_s ; candidates are distinct
                pm_atomic_add(Wq + (int64_t)ci * H + ci, PM_Q(m2, 0));
            } else {
                const int lo = ci < cj ? ci : cj, hi = ci < cj ? cj : ci;
                pm_atomic_add(Wq + (int64_t)lo * H + hi, PM_Q(m2, 0));
            }
        }
        wave_lds_sync();

        for (int h = lane; h < H; h += 64) {
            const double v = s_es[h];
            erow[h] = v;
            atomicAdd(&s_mus[h], PM_Q(v, 0));
        }
        wave_lds_sync();
    }

    // workgroup reduction of the scalar statistics, then one atomic per workgroup and value
    sig = pm_wave_sum(sig);
    fs = pm_wave_sum(fs);
    cnt = pm_wave_sum(cnt);
    if (lane == 0) {
        s_red[wave * 3 + 0] = sig;
        s_red[wave * 3 + 1] = fs;
        s_red[wave * 3 + 2] = cnt;
    }
    __syncthreads();
    double *sc = stats + pm_bsc_stats_offset_scalars_dev(H, D);
    if (tid < 3) {
        double v = 0.0;
        for (int w = 0; w < WAVES; ++w) v += s_red[w * 3 + tid];
        if (v != 0.0) pm_atomic_add(sc + tid, PM_Q(v, tid == 0 ? 1 : tid == 1 ? 2 : 0));
    }
    double *g_qdiag = stats + pm_bsc_stats_offset_qdiag_dev(H, D);
    double *g_mus = stats + pm_bsc_stats_offset_mus_dev(H, D);
    for (int h = tid; h < H; h += 256) {
        pm_atomic_add(g_qdiag + h, s_qdiag[h]);
        pm_atomic_add(g_mus + h, s_mus[h]);
    }
}

inline int64_t grid_for_rows(int64_t N) {
    int64_t blocks = (N + WAVES - 1) / WAVES;
    const int64_t cap = 256 * 8;  // 8 workgroups per CU, grid-stride beyond
    return blocks < cap ? (blocks < 1 ? 1 : blocks) : cap;
}

}  // namespace

// ---- statistics buffer layout -----------------------------------------------------------------
extern "C" int64_t pm_bsc_stats_offset_wq(int64_t H, int64_t D) { return pm_bsc_stats_offset_wq_dev(H, D); }
extern "C" int64_t pm_bsc_stats_offset_qdiag(int64_t H, int64_t D) { return pm_bsc_stats_offset_qdiag_dev(H, D); }
extern "C" int64_t pm_bsc_stats_offset_mus(int64_t H, int64_t D) { return pm_bsc_stats_offset_mus_dev(H, D); }
extern "C" int64_t pm_bsc_stats_offset_scalars(int64_t H, int64_t D) { return pm_bsc_stats_offset_scalars_dev(H, D); }
extern "C" int64_t pm_bsc_stats_len(int64_t H, int64_t D) {
    return pm_bsc_stats_offset_scalars_dev(H, D) + PM_BSC_NSCALARS;
}

static int allow_lds(const void *kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int pm_bsc_select_f64(const double *scores, int64_t lds, const double *wnorm2, int64_t wnorm2_stride,
                                 const double *ynorm2, int64_t N, int64_t H, int64_t Hprime, int32_t *cand,
                                 void *stream) {
    if (!scores || !wnorm2 || !ynorm2 || !cand || N < 0 || H <= 0 || Hprime <= 0 || lds < H || wnorm2_stride <= 0)
        return PM_EINVAL;
    if (H > PM_MAX_H || Hprime > PM_MAX_HPRIME || Hprime > H) return PM_ERANGE;
    if (N == 0) return PM_OK;
    dim3 grid((unsigned)grid_for_rows(N)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PM_SELECT(V)                                                                                              \
    hipLaunchKernelGGL(bsc_select_kernel<V>, grid, block, 0, s, scores, lds, wnorm2, wnorm2_stride, ynorm2, N, (int)H, \
                       (int)Hprime, cand)
    if (H <= 64) PM_SELECT(1);
    else if (H <= 128) PM_SELECT(2);
    else if (H <= 256) PM_SELECT(4);
    else if (H <= 512) PM_SELECT(8);
    else PM_SELECT(16);
#undef PM_SELECT
    return (int)hipGetLastError();
}

extern "C" int pm_bsc_estep_f64(const double *scores, int64_t lds, const double *gram, const double *ynorm2,
                                const double *wmu, const double *ymu, const int32_t *cand,
                                const uint16_t *state_masks, int64_t S, const pm_bsc_estep_params *params_host,
                                int64_t N, int64_t H, int64_t Hprime, double *logpj, int64_t ldl, double *lse,
                                void *stream) {
    if (!scores || !gram || !ynorm2 || !cand || !params_host || !logpj || N < 0 || H <= 0 || Hprime <= 0 || S < 0 ||
        lds < H || ldl < 1 + H + S || (S > 0 && !state_masks) || ((wmu == nullptr) != (ymu == nullptr)))
        return PM_EINVAL;
    if (H > PM_MAX_H || Hprime > PM_MAX_HPRIME || Hprime > H || S > 65535) return PM_ERANGE;
    if (N == 0) return PM_OK;
    const size_t shmem = sizeof(double) * (H + WAVES * (16 + 256)) + sizeof(uint16_t) * S;
    if (shmem > 160 * 1024) return PM_ERANGE;
    if (int e = allow_lds(reinterpret_cast<const void *>(bsc_estep_kernel), shmem)) return e;
    hipLaunchKernelGGL(bsc_estep_kernel, dim3((unsigned)grid_for_rows(N)), dim3(256), shmem,
                       static_cast<hipStream_t>(stream), scores, lds, gram, ynorm2, wmu, ymu, cand, state_masks, (int)S,
                       *params_host, N, (int)H, (int)Hprime, logpj, ldl, lse);
    return (int)hipGetLastError();
}

extern "C" int pm_bsc_mstep_rows_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                                     const int32_t *cand, const uint16_t *state_masks, int64_t S,
                                     const int32_t *pair_ptr, const uint16_t *pair_states, int64_t pair_len,
                                     const pm_bsc_estep_params *params_host, int64_t N, int64_t H, int64_t D,
                                     int64_t Hprime, double *expect, int64_t lde, double *stats, void *stream) {
    if (!logpj || !lse || !cand || !pair_ptr || !params_host || !expect || !stats || N < 0 || H <= 0 || D <= 0 ||
        Hprime <= 0 || S < 0 || pair_len < 0 || ldl < 1 + H + S || lde < H ||
        (S > 0 && (!state_masks || !pair_states)))
        return PM_EINVAL;
    if (H > PM_MAX_H || Hprime > PM_MAX_HPRIME || Hprime > H || S > 65535) return PM_ERANGE;
    if (N == 0) return PM_OK;
    const size_t shmem = sizeof(double) * (2 * H + WAVES * (H + S) + 3 * WAVES) +
                         sizeof(int32_t) * (Hprime * Hprime + 1) + sizeof(uint16_t) * (S + pair_len);
    if (shmem > 160 * 1024) return PM_ERANGE;
    if (int e = allow_lds(reinterpret_cast<const void *>(bsc_mstep_rows_kernel), shmem)) return e;
    hipLaunchKernelGGL(bsc_mstep_rows_kernel, dim3((unsigned)grid_for_rows(N)), dim3(256), shmem,
                       static_cast<hipStream_t>(stream), logpj, ldl, lse, lse_cut, cand, state_masks, (int)S, pair_ptr,
                       pair_states, *params_host, N, (int)H, (int)D, (int)Hprime, expect, lde, stats);
    return (int)hipGetLastError();
}

PM_DET_SETTER(bsc_kernels)
