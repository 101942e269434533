// Maximal Causes Analysis (prosper/em/camodels/mca_et.py) on gfx950.
//
//   mca_select_scores_kernel   sim[n,h] = sum_d max(W_hd - y_d, 0)        (mca_et.py:104-106)
//                              a register-tiled N x H x D max-plus contraction: no MFMA form exists,
//                              the bound is the f64 VALU (2 ops per (n,h,d) as sum max(W, y) - sum y)
//   mca_estep_kernel           log-pseudo-joints (mca_et.py:142-175): singletons from the scores GEMM
//                              (Gram identity), multi-cause states from
//                              Wbar_sd = (sum_{j in s} W_{c_j d}^rho)^(1/rho) -- one f64 log+exp per
//                              (state, d): the kernel is transcendental-bound, not HBM-bound
//   mca_mstep_rows_kernel      posterior weights q ~ exp(beta logpj), sufficient statistics
//                              (mca_et.py:236-327) with
//                              (W_j / Wbar_s)^(rho-1) = W_j^(rho-1) * Wbar_s / T_s   (T_s = Wbar_s^rho)
//                              so the M-step needs no transcendental beyond Wbar, and only for the few
//                              states whose posterior is not negligible
//
// One 64-lane wavefront per datapoint; lane l owns observed dimensions d = l + 64 i (i < DPL): the
// H' x D block W^rho[cand] sits in LDS (row reads are contiguous across lanes), the state loop is
// wave-uniform (every lane walks the same state), the sum over d of each state is a wave reduction.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

// documented part of the statistics buffer; the per-XCD copies of [Wp | Wq] follow it (pm_common.h)
__host__ __device__ static inline int64_t mca_stats_base(int64_t H, int64_t D) { return 3 * H * D + H + PM_MCA_NSCALARS; }

#ifndef PM_MCA_TSUM
#define PM_MCA_TSUM 0   // the states' row sums in the fused pass: 1 prefix in registers + the new row a trip ahead, 0 the same without the read-ahead, 2 two rows per state
#endif
#ifndef PM_MCA_VDEFER
#define PM_MCA_VDEFER 0 // a state's V update (and the rare rescaling) at the top of the next trip: one straight-line block for both stages
#endif
#ifndef PM_MCA_PAIR
#define PM_MCA_PAIR -1  // two states per trip of the fused pass's state loop: -1 where one wavefront per SIMD runs anyway, 0 never, 1 always
#endif
#ifndef PM_MCA_NS
#define PM_MCA_NS 2     // states per trip of that form
#endif
#ifndef PM_MCA_ABL
#define PM_MCA_ABL 0   // timing ablations (scratch/mca_abl.sh): 1 no global atomics, 2 no powers, 3 no V updates, 4 T sums from candidate 0 only, 5 no wave reduction, 6 no exponential, 7 no states at all (S = 0)
#endif

#ifndef PM_SCATTER_LDS_DOUBLES
#define PM_SCATTER_LDS_DOUBLES 16384  // LDS of a mca_defer_scatter_kernel workgroup, in doubles: 128 KB, one workgroup per CU (8192 = two per CU: 0.68 against 0.46 ms -- twice the latent ranges, every datapoint scanned and its y row read by twice as many workgroups)
#endif
#ifndef PM_SCATTER_ABL
#define PM_SCATTER_ABL 0          // timing-only ablations of mca_defer_scatter_kernel (wrong results): 1 no LDS atomics, 2 no record loads
#endif
// the fused pass's root / power table area: pm_load_root21 / pm_load_root6 (A/B builds) or pm_load_upow (any rho)
#define PM_FUSED_RT_LEN (PM_ROOT21_LEN + 1)      // (>= PM_UPOW_AB_LEN; NOT larger: four workgroups of two wavefronts fill a CU's LDS to within 1.6 KB at config 5)

namespace {

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// sim[n,h] = sum_d max(W[h,d] - Y[n,d], 0): 64 x 64 output tile per 256-thread workgroup, 4 x 4 per
// thread, D walked in 16-column slabs through LDS.
// ---------------------------------------------------------------------------------------------
constexpr int ST = 64, SK = 16, SLD = SK + 2;      // rows of 18 doubles: 16-byte aligned pairs, conflict-free ds_read_b128

__global__ __launch_bounds__(256) void mca_select_scores_kernel(const double *__restrict__ Y, int64_t ldy,
                                                                 const double *__restrict__ W, int64_t ldw,
                                                                 double *__restrict__ R, int64_t ldr, int64_t N,
                                                                 int H, int D, int tiles_h) {
    __shared__ __attribute__((aligned(16))) double sy[ST * SLD], sw[ST * SLD];
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)(blockIdx.x / tiles_h) * ST;
    const int h0 = (blockIdx.x % tiles_h) * ST;
    const int tr = tid >> 4, tc = tid & 15;  // thread tile: rows tr + 16 a, cols tc + 16 b
    double acc[4][4], ysum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

    // slab k0 + SK is requested into registers BEFORE slab k0 is used: with the load -> LDS -> barrier -> compute sequence of
    // the first version every workgroup sat out a memory round trip per slab (0.41 ms at config 5 whatever the inner loop
    // issued: two instructions per (n, h, d) instead of three, wider LDS reads -- nothing moved it)
    double ny[4], nw[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e, r = idx >> 4, c = idx & 15;
            const int64_t n = n0 + r;
            ny[e] = (n < N && k0 + c < D) ? Y[n * ldy + k0 + c] : 0.0;
            // padding columns: y = 0, w = 0 contribute max(0 - 0, 0) = 0
            nw[e] = (h0 + r < H && k0 + c < D) ? W[(int64_t)(h0 + r) * ldw + k0 + c] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < D; k0 += SK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e, r = idx >> 4, c = idx & 15;
            sy[r * SLD + c] = ny[e];
            sw[r * SLD + c] = nw[e];
        }
        __syncthreads();
        if (k0 + SK < D) fetch(k0 + SK);
#pragma unroll
        for (int k = 0; k < SK; k += 2) {
            d2 yv[4], wv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) yv[a] = *reinterpret_cast<const d2 *>(&sy[(tr + 16 * a) * SLD + k]);
#pragma unroll
            for (int b = 0; b < 4; ++b) wv[b] = *reinterpret_cast<const d2 *>(&sw[(tc + 16 * b) * SLD + k]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#ifndef PM_MCA_SEL3
                    // sum_d max(W - y, 0) = sum_d max(W, y) - sum_d y: two f64 instructions per (n, h, d) instead of three
                    // (one raw v_max_f64 -- fmax() canonicalises operands it cannot prove quiet -- and one add; the row
                    // sums of y ride along, one add per datapoint and d): 0.343 -> 0.283 ms at config 5 once the slabs are
                    // prefetched (before that the pass was latency-bound and this form measured no faster).  The difference
                    // of two sums of ~500 rounds at ~1e-13 absolute; the ranking it feeds separates candidates by O(1).
                    // -DPM_MCA_SEL3 builds the three-instruction form.
                    double m0, m1;
                    asm("v_max_f64 %0, %1, %2" : "=v"(m0) : "v"(wv[b].x), "v"(yv[a].x));
                    asm("v_max_f64 %0, %1, %2" : "=v"(m1) : "v"(wv[b].y), "v"(yv[a].y));
                    acc[a][b] += m0;
                    acc[a][b] += m1;
#else
                    acc[a][b] += fmax(wv[b].x - yv[a].x, 0.0);
                    acc[a][b] += fmax(wv[b].y - yv[a].y, 0.0);
#endif
                }
#ifndef PM_MCA_SEL3
#pragma unroll
            for (int a = 0; a < 4; ++a) ysum[a] += yv[a].x + yv[a].y;
#endif
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t n = n0 + tr + 16 * a;
        if (n >= N) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int h = h0 + tc + 16 * b;
#ifndef PM_MCA_SEL3
            if (h < H) R[n * ldr + h] = acc[a][b] - ysum[a];
#else
            (void)ysum;
            if (h < H) R[n * ldr + h] = acc[a][b];
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------
// E-step
// ---------------------------------------------------------------------------------------------
template <int DPL>  // observed dimensions per lane: D <= 64 * DPL
__global__ __launch_bounds__(256) void mca_estep_kernel(const double *__restrict__ scores, int64_t lds, const double *__restrict__ wnorm2,
                                 const double *__restrict__ ynorm2, const double *__restrict__ Y, int64_t ldy,
                                 const double *__restrict__ Wrho, const int32_t *__restrict__ cand,
                                 const uint16_t *__restrict__ masks, int S, pm_mca_params P, int64_t N, int H, int D,
                                 int Hp, double *__restrict__ logpj, int64_t ldl, double *__restrict__ lse1,
                                 double *__restrict__ lseb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ power tables (PM_POWTAB_LEN) | per wave: wr (Hp * DS) | e (S) ] ; DS = 64 * DPL
    constexpr int DS = 64 * DPL;
    const int waves = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *s_tab = reinterpret_cast<double *>(smem);
    double *s_wr = s_tab + PM_POWTAB_LEN + (size_t)wave * (Hp * DS + S);
    double *s_e = s_wr + Hp * DS;
    pm_load_powtab(s_tab, tid, blockDim.x);
    // rho = 21 (every temperature T <= 1.05), unsigned W: the log / exp-free power (pm_pow_m20_21)
    __shared__ __attribute__((aligned(16))) double s_rt[PM_ROOT21_LEN + 1];
    const bool r21 = P.signed_w == 0.0 && P.inv_rho > 0.0 && fabs(1.0 / P.inv_rho - 21.0) < 1e-9;
    const bool r6 = P.inv_rho > 0.0 && fabs(1.0 / P.inv_rho - 6.0) < 1e-9;       // (MMCA's steady rho; either sign of W)
    if (!PM_POW_HWSEED && r21) pm_load_root21(s_rt, pm_powtab_dev, tid, blockDim.x);       // (the table seed's tables: A/B builds)
    else if (!PM_POW_HWSEED && r6) pm_load_root6(s_rt, pm_powtab_dev, tid, blockDim.x);
    __syncthreads();

    const int64_t wave0 = (int64_t)blockIdx.x * waves + wave;
    const int64_t nwaves = (int64_t)gridDim.x * waves;
    for (int64_t n = wave0; n < N; n += nwaves) {
        const int32_t *cn = cand + n * Hp;
        double y[DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = lane + 64 * i;
            y[i] = (d < D) ? Y[n * ldy + d] : 0.0;
        }
        for (int j = 0; j < Hp; ++j) {
            const double *src = Wrho + (int64_t)cn[j] * D;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                s_wr[j * DS + d] = (d < D) ? src[d] : 0.0;
            }
        }
        wave_sync_lds();

        // multi-cause states: e_s = sum_d (Wbar_sd - y_d)^2.  Padding dimensions hold W^rho = 0 and y = 0:
        // Wbar = 0 there and they add nothing, so the loop body is branch-free and the DPL powers of a
        // lane are independent instruction streams.
        // Round 5: TWO states per trip (2 * DPL independent power chains and two interleaved wave reductions: without the
        // fused pass's second pipeline stage a lone state's chain is all latency), the 64 masks of a batch in a lane
        // register (one load per batch instead of one vector-memory round trip per state), and the row sums as in the
        // fused pass: the sum over all but the state's highest candidate stays in registers while consecutive states
        // share it (itertools.combinations order), so a state reads one row; same bits as the straight sum.
        auto states = [&](auto root_tag) {
            constexpr int ROOT = decltype(root_tag)::value;
            double Pf[DPL];
            unsigned pfx = 0xFFFFFFFFu;
            auto T_of = [&](unsigned m, double (&T)[DPL]) {
                const int hb = 31 - __builtin_clz(m | 1u);
                unsigned pm = m & ~(1u << hb);
                if (pm != pfx) {                  // uniform
                    pfx = pm;
#pragma unroll
                    for (int i = 0; i < DPL; ++i) Pf[i] = 0.0;
                    while (pm) {
                        const int j = __builtin_ctz(pm);
                        pm &= pm - 1;
                        const double *wr = s_wr + j * DS + lane;
#pragma unroll
                        for (int i = 0; i < DPL; ++i) Pf[i] += wr[64 * i];
                    }
                }
                const double *wr = s_wr + hb * DS + lane;
#pragma unroll
                for (int i = 0; i < DPL; ++i) T[i] = Pf[i] + wr[64 * i];
                if (m == 0u) {                    // (no candidate: never among the multi-cause states)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < DPL; ++i) T[i] = 0.0;
                }
            };
            auto energy = [&](const double (&T)[DPL]) {
                double part = 0.0;
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    // MMCA: T may be negative or 0 (mmca_et.py:191: sign(t) exp(log|t| / rho)); for MCA T > 0
                    const double aT = fabs(T[i]);
                    const double wbar = (aT > 0.0) ? (ROOT == 21 ? aT * pm_pow_m20_21(aT, s_rt)
                                                      : ROOT == 6 ? copysign(aT * pm_pow_m5_6(aT, s_rt), T[i])
                                                                  : copysign(pm_pow_tab(aT, P.inv_rho, s_tab), T[i]))
                                                   : 0.0;
                    const double df = wbar - y[i];
                    part = fma(df, df, part);
                }
                return part;
            };
            for (int s0 = 0; s0 < S; s0 += 64) {
                const unsigned ml = masks[s0 + lane < S ? s0 + lane : S - 1];
                const int cnt = S - s0 < 64 ? S - s0 : 64;
                for (int k = 0; k < cnt; k += 2) {
                    const unsigned m0 = (unsigned)__builtin_amdgcn_readlane((int)ml, k);
                    const unsigned m1 = (unsigned)__builtin_amdgcn_readlane((int)ml, k + 1 < 64 ? k + 1 : k);   // (a repeat at the odd end)
                    double T0[DPL], T1[DPL];
                    T_of(m0, T0);
                    T_of(m1, T1);
                    double p0 = energy(T0), p1 = energy(T1);
                    p0 = pm_wave_sum_dpp(p0);
                    p1 = pm_wave_sum_dpp(p1);
                    if (lane == 0) {
                        s_e[s0 + k] = p0;
                        if (k + 1 < cnt) s_e[s0 + k + 1] = p1;
                    }
                }
            }
        };
        if (r21) states(std::integral_constant<int, 21>{});
        else if (r6) states(std::integral_constant<int, 6>{});
        else states(std::integral_constant<int, 0>{});
        wave_sync_lds();

        // log-pseudo-joints and the two log-evidences (beta = 1 for Q, beta = 1/T for the weights)
        const double yn = ynorm2[n];
        const double *arow = scores + n * lds;
        double *out = logpj + n * ldl;
        double m1 = -INFINITY;
        if (lane == 0) {
            const double f0 = P.pre1 * yn;
            out[0] = f0;
            m1 = f0;
        }
        for (int h = lane; h < H; h += 64) {
            const double f = P.pil_bar + P.pre1 * (wnorm2[h] - 2.0 * arow[h] + yn);
            out[1 + h] = f;
            m1 = fmax(m1, f);
        }
        for (int s = lane; s < S; s += 64) {
            const double f = P.pil_bar * (double)__builtin_popcount((unsigned)masks[s]) + P.pre1 * s_e[s];
            out[1 + H + s] = f;
            s_e[s] = f;
            m1 = fmax(m1, f);
        }
        m1 = pm_wave_max(m1);
        double s1 = 0.0, sb = 0.0;
        if (lane == 0) {
            const double dlt = out[0] - m1;  // own store, same lane
            s1 += exp(dlt);
            sb += exp(P.beta * dlt);
        }
        for (int h = lane; h < H; h += 64) {
            const double dlt = (P.pil_bar + P.pre1 * (wnorm2[h] - 2.0 * arow[h] + yn)) - m1;
            if (dlt > -745.0) {
                s1 += exp(dlt);
                sb += exp(P.beta * dlt);
            }
        }
        for (int s = lane; s < S; s += 64) {
            const double dlt = s_e[s] - m1;
            if (dlt > -745.0) {
                s1 += exp(dlt);
                sb += exp(P.beta * dlt);
            }
        }
        s1 = pm_wave_sum(s1);
        sb = pm_wave_sum(sb);
        if (lane == 0) {
            lse1[n] = m1 + log(s1);
            lseb[n] = P.beta * m1 + log(sb);
        }
        wave_sync_lds();
    }
}

// ---------------------------------------------------------------------------------------------
// E-step + per-datapoint M-step statistics in ONE pass (no data truncation): the multi-cause powers are the
// cost of both mca_estep_kernel and mca_mstep_rows_kernel, here they are evaluated once.  The posterior
// weights of the states are not known until every state has been seen, so the Aid accumulators
// V[j][d] = sum_s w_s T_sd^(1/rho-1) are kept relative to a running maximum M of beta*f_s (rescaled when it
// moves, like an online softmax) and normalised by exp(M - lse_beta) at the end.  A stored term is never
// smaller than its final value, so nothing underflows that would survive in the two-pass form.
// ---------------------------------------------------------------------------------------------

// ROOT = 21: rho = 21 (every temperature T <= 1.05, unsigned W): the states' power through pm_pow_m20_21 (no log / exp);
// ROOT = 6: rho = 6 (MMCA at every T <= 1.2): pm_pow_m5_6; ROOT = 0: any rho, the table power
// (The body is a device function shared by two kernels: mca_estep_fused_kernel keeps the argument list it had before the
// deferred form existed -- two more pointer arguments alone moved the rho = 21 instantiation from 202 to 235 registers and
// from 6.6 to 7.7 ms --, mca_estep_fused_defer_kernel adds the record pointers.)
template <int DPL, int HP, bool SIGNED, int ROOT, bool DEFER>      // ROOT: 21 / 6 = the log / exp-free powers of those rho, 0 = the uniform-exponent power; DEFER: statistics as per-datapoint records
__device__ __forceinline__ void mca_estep_fused_body(const double *__restrict__ scores, int64_t lds,
                                       const double *__restrict__ wnorm2, const double *__restrict__ ynorm2,
                                       const double *__restrict__ Y, int64_t ldy, const double *__restrict__ Wrho,
                                       const double *__restrict__ Wrm1, const int32_t *__restrict__ cand,
                                       const uint16_t *__restrict__ masks, int S, pm_mca_params P, int64_t N, int H,
                                       int D, int Hp, double *__restrict__ logpj, int64_t ldl,
                                       double *__restrict__ lse1, double *__restrict__ lseb,
                                       double *__restrict__ q1, int64_t ldq, double *__restrict__ stats,
                                       double *__restrict__ defer_rec, double *__restrict__ defer_sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // DEFERRED statistics (defer_rec given; round 6): a data-truncation step (mca_et.py:248-262) keeps the N_use datapoints
    // with the largest log-denominator, known only once every rank's E-step is through -- the pass then ACCUMULATES NOTHING:
    // it leaves each datapoint's Aid block (Hp x D, what it would have scattered into Wq; times y into Wp) in defer_rec, its
    // three scalars in defer_sc, its singleton posteriors in q1 as always, and mca_defer_apply_kernel adds the kept ones.
    // [ power tables (PM_POWTAB_LEN) | root table (PM_ROOT21_LEN) | q1sum (H) | red (4 * waves) | per wave: wr (HP*DS)
    //   [wm (HP*DS)] e (S) ]
    constexpr int DS = 64 * DPL;
    const int waves = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *s_tab = reinterpret_cast<double *>(smem);
    double *s_rt = s_tab + PM_POWTAB_LEN;
    double *s_q1sum = s_rt + PM_FUSED_RT_LEN;            // (an even number of doubles: 16-byte alignment of what follows)
    if (ROOT == 0) {
        // any other rho (every step of an annealing ramp): the tables of THIS launch's exponent 1 / rho - 1 (pm_pow_uni) --
        // its (r_i, r_i^-c) pairs take the place of pm_pow_tab's (r_i, L_i) in the power table, pm_exp_tab's E_j stay
        for (int i = 256 + tid; i < PM_POWTAB_LEN; i += blockDim.x) s_tab[i] = pm_powtab_dev[i];
        pm_load_upow(s_tab, s_rt, pm_powtab_dev, P.inv_rho - 1.0, tid, blockDim.x);
    } else {
        pm_load_powtab(s_tab, tid, blockDim.x);
    }
    if (!PM_POW_HWSEED && ROOT == 21) pm_load_root21(s_rt, pm_powtab_dev, tid, blockDim.x);  // (the table seed's tables: A/B builds)
    if (!PM_POW_HWSEED && ROOT == 6) pm_load_root6(s_rt, pm_powtab_dev, tid, blockDim.x);
    double *s_red = s_q1sum + H;
    const size_t per_wave = (size_t)(SIGNED ? 2 : 1) * HP * DS + S;
    double *s_wr = s_red + 4 * waves + (size_t)wave * per_wave;
    double *s_wm = s_wr + HP * DS;                       // SIGNED only
    double *s_e = s_wr + (SIGNED ? 2 : 1) * HP * DS;
    for (int h = tid; h < H; h += blockDim.x) s_q1sum[h] = 0.0;
    __syncthreads();

    // multi-cause numerator / denominator: this XCD's copy (pm_common.h), folded by the launcher
    double *Wp = pm_xcd_copy(stats + (int64_t)H * D, stats + mca_stats_base(H, D), 2 * (int64_t)H * D);
    double *Wq = Wp + (int64_t)H * D;
    double st_pi = 0.0, st_sigma = 0.0, st_ld = 0.0, st_cnt = 0.0;

    const int64_t wave0 = (int64_t)blockIdx.x * waves + wave;
    const int64_t nwaves = (int64_t)gridDim.x * waves;
    for (int64_t n = wave0; n < N; n += nwaves) {
        const int32_t *cn = cand + n * Hp;
        double y[DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = lane + 64 * i;
            y[i] = (d < D) ? Y[n * ldy + d] : 0.0;
        }
        // (round 3, tried: the candidates' W^rho rows in registers instead of LDS, so that the states' T sums are register
        // adds -- 304 registers, one wavefront per SIMD: 9.98 vs 8.12 ms)
        for (int j = 0; j < HP; ++j) {
            const bool have = j < Hp;
            const int64_t base = have ? (int64_t)cn[j] * D : 0;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                s_wr[j * DS + d] = (have && d < D) ? Wrho[base + d] : 0.0;
                if (SIGNED) s_wm[j * DS + d] = (have && d < D) ? Wrm1[base + d] : 1.0;
            }
        }
        wave_sync_lds();

        double V[HP][DPL];
#pragma unroll
        for (int j = 0; j < HP; ++j)
#pragma unroll
            for (int i = 0; i < DPL; ++i) V[j][i] = 0.0;
        // M: reference level of beta * f_s for the stored terms.  It follows the running maximum LAZILY -- only when a
        // state exceeds it by more than 50 (terms up to e^50 relative to it are stored meanwhile: nothing overflows, and
        // a stored term is still never smaller than its final value) -- so the rescaling branch is all but never taken
        // after the first state.
        double M = -INFINITY;

        // Two-stage software pipeline over the states.  Stage A (state s + 1): T = sum of the state's W^rho rows, one
        // power per element, the lane's share of the squared error.  Stage B (state s): wave reduction of that error,
        // beta * f_s, its weight e^(beta f_s - M), the V updates.  Each stage is a long dependent chain (a 36-deep power;
        // a 6-step DPP reduction feeding a 14-deep exponential), and two wavefronts per SIMD (183 VGPRs, 17 KB of LDS
        // each) cannot hide either: 5100 cycles per state for ~300 instructions.  A(s + 1) does not depend on B(s), so
        // both sit in ONE straight-line block and the scheduler interleaves them; the V update is predicated with 0/1
        // factors instead of per-candidate branches for the same reason.
        double wbP[DPL], partP = 0.0;      // state s: |T|^(1/rho - 1) (0 / +inf for T = 0, see below), squared error
        // T of a state = the sum of its candidates' rows in ascending order = (the sum over all but its highest candidate)
        // + that candidate's row.  The states arrive as itertools.combinations lists them: consecutive ones share that
        // prefix (56 of config 5's 84), so the prefix sum stays in registers (Pf, its mask in a scalar) and a state costs ONE
        // row read -- requested a trip ahead, its LDS latency under the previous state's work -- and DPL additions, with
        // no branch in the common case; the bits are those of the straight sum (0 + x = x).  [round 5; before: eight bit
        // tests and up to gamma read -> wait -> add blocks per state, 1.0 of the pass's 7.4 ms]
        double Pf[DPL];
        unsigned pfx = 0xFFFFFFFFu;
        auto sum_rows = [&](unsigned mm, double (&out)[DPL]) {      // the straight sum (uniform bit tests)
#pragma unroll
            for (int i = 0; i < DPL; ++i) out[i] = 0.0;
#pragma unroll
            for (int j = 0; j < HP; ++j)
                if ((mm >> j) & 1u) {
#pragma unroll
                    for (int i = 0; i < DPL; ++i) out[i] += s_wr[j * DS + lane + 64 * i];
                }
        };
        auto row_of = [&](unsigned m, double (&row)[DPL]) {
            const int hb = 31 - __builtin_clz(m | 1u);
#pragma unroll
            for (int i = 0; i < DPL; ++i) row[i] = s_wr[hb * DS + lane + 64 * i];
        };
#if PM_MCA_TSUM == 2
        // two rows per state (its two highest candidates), the sum of the others cached: no branch for 77 of 84 states
        auto state_T = [&](unsigned m, const double (&)[DPL], double (&T)[DPL]) {
            const int hb = 31 - __builtin_clz(m | 1u);
            const unsigned pm = m & ~(1u << hb);
            const int hb2 = 31 - __builtin_clz(pm | 1u);
            const unsigned pm2 = pm & ~(1u << hb2);
            if (pm == 0u) {                       // fewer than two candidates: never among the multi-cause states
                asm volatile("" ::: "memory");
                sum_rows(m, T);
                return;
            }
            if (pm2 != pfx) {
                sum_rows(pm2, Pf);
                pfx = pm2;
            }
#pragma unroll
            for (int i = 0; i < DPL; ++i) T[i] = (Pf[i] + s_wr[hb2 * DS + lane + 64 * i]) + s_wr[hb * DS + lane + 64 * i];
        };
#else
        auto state_T = [&](unsigned m, const double (&row)[DPL], double (&T)[DPL]) {
            const int hb = 31 - __builtin_clz(m | 1u);
            const unsigned pm = m & ~(1u << hb);
            if (pm != pfx) {                      // uniform: the prefix changed (28 of 84 states)
                sum_rows(pm, Pf);
                pfx = pm;
            }
#pragma unroll
            for (int i = 0; i < DPL; ++i) T[i] = Pf[i] + (PM_MCA_TSUM == 0 ? s_wr[hb * DS + lane + 64 * i] : row[i]);
            if (m == 0u) {                        // (no candidate at all: never among the multi-cause states)
                asm volatile("" ::: "memory");    // (a real branch, not eight selects per state)
#pragma unroll
                for (int i = 0; i < DPL; ++i) T[i] = 0.0;
            }
        };
#endif
        auto state_pow = [&](const double (&T)[DPL], double (&wb_out)[DPL], double &part) {
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                // ONE power per element: r = |T|^(1/rho - 1) gives |Wbar| = |T| r here and Wbar / T = r for the
                // M-step weights (no division).  Padding dimensions have T = 0: Wbar = 0, never scattered.
                const double aT = SIGNED ? fabs(T[i]) : T[i];      // (unsigned W: T is a sum of W^rho >= 0)
                const double r = (PM_MCA_ABL == 2) ? aT * 0.37
                                 : (ROOT == 21 ? pm_pow_m20_21(aT, s_rt) : ROOT == 6 ? pm_pow_m5_6(aT, s_rt)
                                                                                 : pm_pow_uni(aT, s_tab, s_rt));
                const double wb = (aT > 0.0) ? aT * r : 0.0;
                const double df = (SIGNED ? copysign(wb, T[i]) : wb) - y[i];
                part = fma(df, df, part);
                wb_out[i] = (aT > 0.0) ? r : (SIGNED ? INFINITY : 0.0);
            }
        };
        // state s's V update: w |T|^(1/rho - 1) into the rows of its own candidates [signed W: times W^(rho - 1)'s row]
        double vD[DPL], wD = 0.0, scD = 0.0;
        unsigned maskD = 0u;
        bool rescD = false;
#pragma unroll
        for (int i = 0; i < DPL; ++i) vD[i] = 0.0;
        auto v_update = [&]() {
            if (PM_MCA_VDEFER && rescD) {
                const double sc = exp(scD);
#pragma unroll
                for (int j = 0; j < HP; ++j)
#pragma unroll
                    for (int i = 0; i < DPL; ++i) V[j][i] *= sc;
            }
            if (!SIGNED) {
#ifndef PM_MCA_VPRED
                // (round 3: uniform branches instead of 0/1 factors over all H' candidates -- 2.7 of 8 rows per state at
                // config 5: 8.15 -> 7.82 ms; -DPM_MCA_VPRED restores the predicated form)
#pragma unroll
                for (int j = 0; j < HP; ++j) {
                    if ((maskD >> j) & 1u) {      // uniform (scalar) branch: only the state's own candidates are touched
#pragma unroll
                        for (int i = 0; i < DPL; ++i) V[j][i] += vD[i];
                    }
                }
#else
#pragma unroll
                for (int j = 0; j < HP; ++j) {
                    const double sel = (((maskD >> j) & 1u) && (PM_MCA_ABL != 3 || wD == 1.2345e-300)) ? 1.0 : 0.0;   // uniform
#pragma unroll
                    for (int i = 0; i < DPL; ++i) V[j][i] = fma(sel, vD[i], V[j][i]);
                }
#endif
            } else {
#pragma unroll
                for (int j = 0; j < HP; ++j)
                    if ((maskD >> j) & 1u) {
#pragma unroll
                        for (int i = 0; i < DPL; ++i)   // T = 0: w * inf = inf (w > 0) or NaN (w = 0); fmin returns w
                            V[j][i] += fmin(wD, wD * vD[i] * s_wm[j * DS + lane + 64 * i]);
                    }
            }
        };
        // TWO states per trip where the pass runs ONE wavefront per SIMD anyway (signed W: two row sets, >= 24 KB of LDS per
        // wavefront -- registers are free there, 278 of 512): stage A evaluates states s + 2 and s + 3 (2 * DPL independent power
        // chains), stage B reduces, weighs and scatters states s and s + 1 -- their two wave reductions and exponentials
        // interleave.  Same operations per state in the same order: the bits of the one-state loop.  MMCA config-5 dimensions:
        // 10.67 -> 9.72 ms.  (At two wavefronts per SIMD the pair form needs 272 registers -- one wavefront, 9.0 against 6.65 ms.)
        constexpr bool PAIRED = (PM_MCA_PAIR == 1) || (PM_MCA_PAIR < 0 && SIGNED && DPL <= 4 && HP * DPL >= 24);
        if constexpr (PAIRED) {
        constexpr int NS = PM_MCA_NS;                 // states per trip
        auto mask_at = [&](int s) { return masks[s < S ? s : S - 1]; };
        unsigned mP[NS], ld[NS];
        double wbQ[NS][DPL], partQ[NS], rowX[DPL];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            mP[u] = S > 0 ? (unsigned)__builtin_amdgcn_readfirstlane((int)mask_at(u)) : 0u;
            ld[u] = S > 0 ? mask_at(NS + u) : 0u;
            partQ[u] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            double T[DPL];
            state_T(mP[u], rowX, T);
            state_pow(T, wbQ[u], partQ[u]);
        }
        auto weigh = [&](unsigned m, double bf, const double (&wb)[DPL]) {      // one state, the reference level may move
            double w = pm_exp_tab(bf - M, s_tab);
            if (bf > M + 50.0) {
                const double sc = exp(M - bf);
#pragma unroll
                for (int j = 0; j < HP; ++j)
#pragma unroll
                    for (int i = 0; i < DPL; ++i) V[j][i] *= sc;
                M = bf;
                w = 1.0;
            }
            maskD = m;
            wD = w;
#pragma unroll
            for (int i = 0; i < DPL; ++i) vD[i] = SIGNED ? wb[i] : w * wb[i];
            v_update();
        };
        for (int s = 0; s < ((PM_MCA_ABL == 7) ? 0 : S); s += NS) {
            unsigned mN[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                mN[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)ld[u]);
                ld[u] = mask_at(s + 2 * NS + u);
            }
            // ---- stage A, states s + NS .. s + 2 NS - 1 (dummies past the end) ----
            double TN[NS][DPL], wbN[NS][DPL], partN[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                partN[u] = 0.0;
                state_T(mN[u], rowX, TN[u]);
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) state_pow(TN[u], wbN[u], partN[u]);
            // ---- stage B, states s .. s + NS - 1 ----
            double part[NS], bf[NS];
            bool moves = false;
#pragma unroll
            for (int u = 0; u < NS; ++u) part[u] = pm_wave_sum_dpp(partQ[u]);   // wave-uniform
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                if (lane == 0 && s + u < S) s_e[s + u] = part[u];
                bf[u] = P.beta * (P.pil_bar * (double)__builtin_popcount(mP[u]) + P.pre1 * part[u]);
                moves = moves || (s + u < S && bf[u] > M + 50.0);
            }
            if (moves) {                               // uniform; the first trip, then hardly ever: one by one
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    if (s + u < S) weigh(mP[u], bf[u], wbQ[u]);
            } else {
                double w[NS];
#pragma unroll
                for (int u = 0; u < NS; ++u) w[u] = pm_exp_tab(bf[u] - M, s_tab);
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    if (s + u < S) {
                        maskD = mP[u];
                        wD = w[u];
#pragma unroll
                        for (int i = 0; i < DPL; ++i) vD[i] = SIGNED ? wbQ[u][i] : w[u] * wbQ[u][i];
                        v_update();
                    }
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                partQ[u] = partN[u];
                mP[u] = mN[u];
#pragma unroll
                for (int i = 0; i < DPL; ++i) wbQ[u][i] = wbN[u][i];
            }
        }
        } else {
        unsigned maskP = S > 0 ? (unsigned)__builtin_amdgcn_readfirstlane((int)masks[0]) : 0u;      // (scalar registers)
        unsigned maskN = S > 0 ? (unsigned)__builtin_amdgcn_readfirstlane((int)masks[S > 1 ? 1 : 0]) : 0u;
        unsigned mask_load = S > 0 ? masks[S > 2 ? 2 : S - 1] : 0u;
        double rowN[DPL];
        {
            double T[DPL], row0[DPL];
            if (PM_MCA_TSUM == 1) row_of(maskP, row0);
            state_T(maskP, row0, T);
            if (PM_MCA_TSUM == 1) row_of(maskN, rowN);
            state_pow(T, wbP, partP);
        }
        for (int s = 0; s < ((PM_MCA_ABL == 7) ? 0 : S); ++s) {
            // wave-uniform masks; the one three states on is requested now (vector-memory latency under this trip's
            // work), the row of the state after next as soon as its mask is here
            const unsigned maskNN = (unsigned)__builtin_amdgcn_readfirstlane((int)mask_load);
            mask_load = masks[s + 3 < S ? s + 3 : S - 1];
            if (PM_MCA_VDEFER) v_update();         // state s - 1 (nothing in the first trip: maskD = 0)
            double rowNN[DPL];
            if (PM_MCA_TSUM == 1) row_of(maskNN, rowNN);
            // ---- stage A, state s + 1 (the last trip computes a dummy) ----
            double T[DPL], wbN[DPL], partN = 0.0;
            if (PM_MCA_ABL == 4) {
#pragma unroll
                for (int i = 0; i < DPL; ++i) T[i] = s_wr[lane + 64 * i];
            } else {
                state_T(maskN, rowN, T);
            }
            state_pow(T, wbN, partN);
            // ---- stage B, state s ----
            const double part = (PM_MCA_ABL == 5) ? partP * 64.0 : pm_wave_sum_dpp(partP);   // wave-uniform
            if (lane == 0) s_e[s] = part;
            const double bf = P.beta * (P.pil_bar * (double)__builtin_popcount(maskP) + P.pre1 * part);
            double w = (PM_MCA_ABL == 6) ? (bf - M) * 0.001 + 1.0 : pm_exp_tab(bf - M, s_tab);   // (meaningless if the reference level moves)
#if PM_MCA_VDEFER
            // the reference level moves (uniform; the first state, then hardly ever): selects here, the rescaling of V and
            // this state's V update at the top of the next trip -- nothing branches between the two stages' chains
            rescD = bf > M + 50.0;
            scD = M - bf;
            M = rescD ? bf : M;
            w = rescD ? 1.0 : w;
            maskD = maskP;
            wD = w;
#pragma unroll
            for (int i = 0; i < DPL; ++i) vD[i] = SIGNED ? wbP[i] : w * wbP[i];
#else
            if (bf > M + 50.0) {                                        // uniform; the first state, then hardly ever
                const double sc = exp(M - bf);
#pragma unroll
                for (int j = 0; j < HP; ++j)
#pragma unroll
                    for (int i = 0; i < DPL; ++i) V[j][i] *= sc;
                M = bf;
                w = 1.0;
            }
            maskD = maskP;
            wD = w;
#pragma unroll
            for (int i = 0; i < DPL; ++i) vD[i] = SIGNED ? wbP[i] : w * wbP[i];
            v_update();
#endif
            partP = partN;
            maskP = maskN;
            maskN = maskNN;
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                wbP[i] = wbN[i];
                if (PM_MCA_TSUM == 1) rowN[i] = rowNN[i];
            }
        }
        if (PM_MCA_VDEFER) v_update();             // the last state
        }
        wave_sync_lds();

        // log-pseudo-joints and the two log-evidences (as mca_estep_kernel)
        const double yn = ynorm2[n];
        const double *arow = scores + n * lds;
        double *out = logpj + n * ldl;
        double m1 = -INFINITY;
        const double f0 = P.pre1 * yn;
        if (lane == 0) {
            out[0] = f0;
            m1 = f0;
        }
        for (int h = lane; h < H; h += 64) {
            const double f = P.pil_bar + P.pre1 * (wnorm2[h] - 2.0 * arow[h] + yn);
            out[1 + h] = f;
            m1 = fmax(m1, f);
        }
        for (int s = lane; s < S; s += 64) {
            const double f = P.pil_bar * (double)__builtin_popcount((unsigned)masks[s]) + P.pre1 * s_e[s];
            out[1 + H + s] = f;
            s_e[s] = f;
            m1 = fmax(m1, f);
        }
        m1 = pm_wave_max(m1);
        double s1 = 0.0, sb = 0.0;
        if (lane == 0) {
            const double dlt = f0 - m1;
            s1 += exp(dlt);
            sb += exp(P.beta * dlt);
        }
        for (int h = lane; h < H; h += 64) {
            const double dlt = (P.pil_bar + P.pre1 * (wnorm2[h] - 2.0 * arow[h] + yn)) - m1;
            if (dlt > -745.0) {
                s1 += exp(dlt);
                sb += exp(P.beta * dlt);
            }
        }
        for (int s = lane; s < S; s += 64) {
            const double dlt = s_e[s] - m1;
            if (dlt > -745.0) {
                s1 += exp(dlt);
                sb += exp(P.beta * dlt);
            }
        }
        s1 = pm_wave_sum(s1);
        sb = pm_wave_sum(sb);
        const double l1 = m1 + log(s1), lb = P.beta * m1 + log(sb);
        if (lane == 0) {
            lse1[n] = l1;
            lseb[n] = lb;
        }

        // ---- M-step statistics of this datapoint (mca_et.py:274-327) ----
        double *qrow = q1 + n * ldq;
        if constexpr (!DEFER) {
        if (lane == 0) {
            st_sigma += exp(P.beta * f0 - lb) * yn;
            st_ld += l1;
            st_cnt += 1.0;
        }
        for (int h = lane; h < H; h += 64) {
            const double e = wnorm2[h] - 2.0 * arow[h] + yn;
            const double q = exp(P.beta * (P.pil_bar + P.pre1 * e) - lb);
            if (q != 0.0) {
                st_sigma += q * e;
                st_pi += q;
                atomicAdd(&s_q1sum[h], PM_Q(q, 2));
            }
            qrow[h] = q;
        }
        for (int s = lane; s < S; s += 64) {
            const double fs = s_e[s];
            const double q = exp(P.beta * fs - lb);
            const double ns = (double)__builtin_popcount((unsigned)masks[s]);
            st_pi += q * ns;
            st_sigma += q * ((fs - P.pil_bar * ns) / P.pre1);
        }
        }
        const double g = exp(M - lb);                   // <= 1: every multi-cause beta*f_s is <= lb
        if constexpr (DEFER) {
            // this datapoint's share of sum q |s| and sum q e (lane partials -> one record), its Aid block as it stands
            double r_pi = 0.0, r_sigma = 0.0;
            if (lane == 0) r_sigma += exp(P.beta * f0 - lb) * yn;
            for (int h = lane; h < H; h += 64) {
                const double e = wnorm2[h] - 2.0 * arow[h] + yn;
                const double q = exp(P.beta * (P.pil_bar + P.pre1 * e) - lb);
                if (q != 0.0) {
                    r_sigma += q * e;
                    r_pi += q;
                }
                qrow[h] = q;
            }
            for (int s = lane; s < S; s += 64) {
                const double fs = s_e[s];
                const double q = exp(P.beta * fs - lb);
                const double ns = (double)__builtin_popcount((unsigned)masks[s]);
                r_pi += q * ns;
                r_sigma += q * ((fs - P.pil_bar * ns) / P.pre1);
            }
            r_pi = pm_wave_sum(r_pi);
            r_sigma = pm_wave_sum(r_sigma);
            if (lane == 0) {
                double *sc_n = defer_sc + 4 * n;
                sc_n[0] = r_pi;
                sc_n[1] = r_sigma;
                sc_n[2] = l1;
                sc_n[3] = 0.0;
            }
            double *rn = defer_rec + n * (int64_t)Hp * D;
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                if (j < Hp) {
                    const int64_t base = (int64_t)cn[j] * D;
#pragma unroll
                    for (int i = 0; i < DPL; ++i) {
                        const int d = lane + 64 * i;
                        if (d < D) rn[j * D + d] = (S > 0 && g != 0.0) ? (SIGNED ? V[j][i] * g : V[j][i] * g * Wrm1[base + d]) : 0.0;
                    }
                }
            }
        }
        if (!DEFER && S > 0 && g != 0.0) {
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                if (j < Hp) {
                    const int64_t base = (int64_t)cn[j] * D;
#pragma unroll
                    for (int i = 0; i < DPL; ++i) {
                        const int d = lane + 64 * i;
                        if (d < D) {
                            const double aid = SIGNED ? V[j][i] * g : V[j][i] * g * Wrm1[base + d];
                            if (aid != 0.0 && (PM_MCA_ABL != 1 || aid == 1.2345e-300)) {
                                pm_atomic_add(Wp + base + d, PM_Q(aid * y[i], 1));
                                pm_atomic_add(Wq + base + d, PM_Q(aid, 0));
                            }
                        }
                    }
                }
            }
        }
        wave_sync_lds();
    }

    st_pi = pm_wave_sum(st_pi);
    st_sigma = pm_wave_sum(st_sigma);
    st_ld = pm_wave_sum(st_ld);
    st_cnt = pm_wave_sum(st_cnt);
    if (lane == 0) {
        s_red[wave * 4 + 0] = st_pi;
        s_red[wave * 4 + 1] = st_sigma;
        s_red[wave * 4 + 2] = st_ld;
        s_red[wave * 4 + 3] = st_cnt;
    }
    __syncthreads();
    double *g_q1sum = stats + 3 * (int64_t)H * D;
    double *sc = g_q1sum + H;
    if (tid < 4) {
        double v = 0.0;
        for (int w = 0; w < waves; ++w) v += s_red[w * 4 + tid];
        if (v != 0.0) pm_atomic_add(sc + tid, PM_Q(v, tid == 1 ? 3 : tid == 2 ? 4 : 2));    // pi | sum q e | sum lse | count
    }
    for (int h = tid; h < H; h += blockDim.x) {
        const double v = s_q1sum[h];
        if (v != 0.0) pm_atomic_add(g_q1sum + h, v);
    }
}

#define PM_MCA_FUSED_BOUNDS __launch_bounds__(256, (HP <= 8 && DPL <= 4 && ROOT == 0 && !(SIGNED && HP * DPL >= 24) ? 2 : 1))     // (the table power at H' <= 8, D <= 256 would take 258 registers: keep two wavefronts per SIMD; the others fit uncapped, and schedule better so; signed W with >= 24 KB of rows per wavefront runs one per SIMD anyway)
template <int DPL, int HP, bool SIGNED, int ROOT>
__global__ PM_MCA_FUSED_BOUNDS void mca_estep_fused_kernel(const double *__restrict__ scores, int64_t lds,
                                       const double *__restrict__ wnorm2, const double *__restrict__ ynorm2,
                                       const double *__restrict__ Y, int64_t ldy, const double *__restrict__ Wrho,
                                       const double *__restrict__ Wrm1, const int32_t *__restrict__ cand,
                                       const uint16_t *__restrict__ masks, int S, pm_mca_params P, int64_t N, int H,
                                       int D, int Hp, double *__restrict__ logpj, int64_t ldl,
                                       double *__restrict__ lse1, double *__restrict__ lseb,
                                       double *__restrict__ q1, int64_t ldq, double *__restrict__ stats) {
    mca_estep_fused_body<DPL, HP, SIGNED, ROOT, false>(scores, lds, wnorm2, ynorm2, Y, ldy, Wrho, Wrm1, cand, masks, S, P, N, H,
                                                       D, Hp, logpj, ldl, lse1, lseb, q1, ldq, stats, nullptr, nullptr);
}
template <int DPL, int HP, bool SIGNED, int ROOT>
__global__ PM_MCA_FUSED_BOUNDS void mca_estep_fused_defer_kernel(const double *__restrict__ scores, int64_t lds,
                                       const double *__restrict__ wnorm2, const double *__restrict__ ynorm2,
                                       const double *__restrict__ Y, int64_t ldy, const double *__restrict__ Wrho,
                                       const double *__restrict__ Wrm1, const int32_t *__restrict__ cand,
                                       const uint16_t *__restrict__ masks, int S, pm_mca_params P, int64_t N, int H,
                                       int D, int Hp, double *__restrict__ logpj, int64_t ldl,
                                       double *__restrict__ lse1, double *__restrict__ lseb,
                                       double *__restrict__ q1, int64_t ldq, double *__restrict__ stats,
                                       double *__restrict__ defer_rec, double *__restrict__ defer_sc) {
    mca_estep_fused_body<DPL, HP, SIGNED, ROOT, true>(scores, lds, wnorm2, ynorm2, Y, ldy, Wrho, Wrm1, cand, masks, S, P, N, H,
                                                      D, Hp, logpj, ldl, lse1, lseb, q1, ldq, stats, defer_rec, defer_sc);
}

// (A variant with TWO wavefronts per datapoint -- the observed dimensions split between them, four wavefronts per SIMD --
// measured 8.93 ms against 8.03 ms: everything per state that does not shrink with the dimensions is paid twice, 3.81 G
// instead of 2.54 G VALU wave-instructions per launch; profiles/r03_mca_pair.txt, code in scratch/mca_pair_kernel_r03.hip.)

// ---------------------------------------------------------------------------------------------
// M-step, per-datapoint part
// ---------------------------------------------------------------------------------------------
template <int DPL, int HP, bool SIGNED>
__global__ __launch_bounds__(256) void mca_mstep_rows_kernel(const double *__restrict__ logpj, int64_t ldl, const double *__restrict__ lse1,
                                      const double *__restrict__ lseb, double lse_cut,
                                      const double *__restrict__ Y, int64_t ldy, const double *__restrict__ Wrho,
                                      const double *__restrict__ Wrm1, const int32_t *__restrict__ cand,
                                      const uint16_t *__restrict__ masks, int S, pm_mca_params P, int64_t N, int H,
                                      int D, int d0, int Dl, int Hp, double *__restrict__ q1, int64_t ldq,
                                      double *__restrict__ stats) {
    // this launch covers the observed dimensions [d0, d0 + Dl) (Dl <= 64 * DPL); D is the row length of the
    // tables and of Wp / Wq.  The singleton weights and the scalar statistics belong to the slab d0 == 0.
    // HP = register-tile height (Hp rounded up to 4 / 8 / 12); state masks only use bits < Hp
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ power tables (PM_POWTAB_LEN) | q1sum (H) | red (4 * waves) | per wave: wr (HP * DS) ]
    constexpr int DS = 64 * DPL;
    const int waves = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *s_tab = reinterpret_cast<double *>(smem);
    double *s_q1sum = s_tab + PM_POWTAB_LEN;
    pm_load_powtab(s_tab, tid, blockDim.x);
    double *s_red = s_q1sum + H;
    double *s_wr = s_red + 4 * waves + (size_t)wave * ((SIGNED ? 2 : 1) * HP * DS);
    double *s_wm = s_wr + HP * DS;   // SIGNED only: |W|^(rho-1)[cand]
    for (int h = tid; h < H; h += blockDim.x) s_q1sum[h] = 0.0;
    __shared__ __attribute__((aligned(16))) double s_rt[PM_ROOT21_LEN + 1];       // (see mca_estep_kernel)
    const bool r21 = !SIGNED && P.inv_rho > 0.0 && fabs(1.0 / P.inv_rho - 21.0) < 1e-9;
    const bool r6 = SIGNED && P.inv_rho > 0.0 && fabs(1.0 / P.inv_rho - 6.0) < 1e-9;
    if (!PM_POW_HWSEED && r21) pm_load_root21(s_rt, pm_powtab_dev, tid, blockDim.x);       // (the table seed's tables: A/B builds)
    else if (!PM_POW_HWSEED && r6) pm_load_root6(s_rt, pm_powtab_dev, tid, blockDim.x);
    __syncthreads();

    // multi-cause numerator / denominator (stats[0 .. H*D) = Q1^T Y by the GEMM): this XCD's copy, folded by
    // the launcher
    double *Wp = pm_xcd_copy(stats + (int64_t)H * D, stats + mca_stats_base(H, D), 2 * (int64_t)H * D);
    double *Wq = Wp + (int64_t)H * D;
    // W_new = Wp / Wq is an ELEMENT-wise ratio (mca_et.py:348): an element (h,d) can be dominated by a
    // state of vanishing posterior whose factor (W_hd / Wbar_sd)^(rho-1) is ~1 while the likely states'
    // factors are ~1e-100 there.  Only weights that underflow to exactly 0 -- as they do in the
    // reference -- may be dropped.
    const double qcut = -745.2;
    const bool first = d0 == 0;
    double st_pi = 0.0, st_sigma = 0.0, st_ld = 0.0, st_cnt = 0.0;

    const int64_t wave0 = (int64_t)blockIdx.x * waves + wave;
    const int64_t nwaves = (int64_t)gridDim.x * waves;
    for (int64_t n = wave0; n < N; n += nwaves) {
        double *qrow = q1 + n * ldq;
        const double lb = lseb[n];
        if (!(lb >= lse_cut)) {  // truncated (mca_et.py:243-253)
            if (first)
                for (int h = lane; h < H; h += 64) qrow[h] = 0.0;
            continue;
        }
        const double *f = logpj + n * ldl;
        const int32_t *cn = cand + n * Hp;
        if (first && lane == 0) {
            const double f0 = f[0];
            const double dlt = P.beta * f0 - lb;
            st_sigma += exp(dlt) * (f0 / P.pre1);
            st_ld += lse1[n];
            st_cnt += 1.0;
        }
        for (int h = lane; first && h < H; h += 64) {
            const double fh = f[1 + h];
            const double dlt = P.beta * fh - lb;
            const double q = exp(dlt);       // singletons are never dropped (underflow aside)
            if (q != 0.0) {
                st_sigma += q * ((fh - P.pil_bar) / P.pre1);
                st_pi += q;
                atomicAdd(&s_q1sum[h], PM_Q(q, 2));
            }
            qrow[h] = q;
        }

        // significant multi-cause states (wave-uniform walk; most are skipped)
        bool staged = false;
        double V[HP][DPL], y[DPL];
#pragma unroll
        for (int j = 0; j < HP; ++j)
#pragma unroll
            for (int i = 0; i < DPL; ++i) V[j][i] = 0.0;
        unsigned touched = 0;
        // Round 5: the batch's masks ride in a lane register beside the states' log-joints (before: one vector-memory
        // round trip per live state), the row sums keep the shared prefix in registers as in the fused pass (one row
        // read per state, same bits), and TWO live states go through the powers together (2 * DPL independent chains).
        double Pf[DPL];
        unsigned pfx = 0xFFFFFFFFu;
        auto T_of = [&](unsigned m, double (&T)[DPL]) {
            const int hb = 31 - __builtin_clz(m | 1u);
            unsigned pm = m & ~(1u << hb);
            if (pm != pfx) {                      // uniform
                pfx = pm;
#pragma unroll
                for (int i = 0; i < DPL; ++i) Pf[i] = 0.0;
                while (pm) {
                    const int j = __builtin_ctz(pm);
                    pm &= pm - 1;
                    const double *wr = s_wr + j * DS + lane;
#pragma unroll
                    for (int i = 0; i < DPL; ++i) Pf[i] += wr[64 * i];
                }
            }
            const double *wr = s_wr + hb * DS + lane;
#pragma unroll
            for (int i = 0; i < DPL; ++i) T[i] = Pf[i] + wr[64 * i];
            if (m == 0u) {                        // (no candidate: never among the multi-cause states)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < DPL; ++i) T[i] = 0.0;
            }
        };
        auto weights = [&](const double (&T)[DPL], double q, double (&v)[DPL]) {
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                if (!SIGNED) {
                    // q_s Wbar_sd / T_sd = q_s T^(1/rho - 1); padding: T = 0
                    v[i] = (T[i] > 0.0) ? q * (r21 ? pm_pow_m20_21(T[i], s_rt) : pm_pow_tab(T[i], P.inv_rho - 1.0, s_tab)) : 0.0;
                } else {
                    // q_s min(1, (|W_jd| / |Wbar_sd|)^(rho-1)), (.)^(rho-1) = |W_jd|^(rho-1) |Wbar_sd| / |t_sd|;
                    // t = 0 gives Wbar = 0 and the factor 1 (mmca_et.py:316-324: max(-inf - Wl, 0) = 0)
                    const double aT = fabs(T[i]);
                    v[i] = (aT > 0.0) ? q * (r6 ? pm_pow_m5_6(aT, s_rt) : pm_pow_tab(aT, P.inv_rho - 1.0, s_tab)) : INFINITY;
                }
            }
        };
        auto v_update = [&](unsigned mask, double q, const double (&v)[DPL]) {
#pragma unroll
            for (int j = 0; j < HP; ++j)
                if ((mask >> j) & 1u) {
#pragma unroll
                    for (int i = 0; i < DPL; ++i)
                        V[j][i] += SIGNED ? fmin(q, v[i] * s_wm[j * DS + lane + 64 * i]) : v[i];
                }
        };
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int sl = s0 + lane;
            double dl = -INFINITY, fl = 0.0;
            unsigned ml = 0u;
            if (sl < S) {
                fl = f[1 + H + sl];
                dl = P.beta * fl - lb;
                ml = masks[sl];
            }
            const double ql = exp(dl);        // this lane's state; broadcast below (one exp per 64 states)
            unsigned long long live = __ballot(dl > qcut);
            while (live) {
                const int src = __builtin_ctzll(live);
                live &= live - 1;
                const bool two = live != 0ull;
                const int src2 = two ? __builtin_ctzll(live) : src;
                live &= live - 1;             // (0 & anything = 0)
                const unsigned mask = (unsigned)__builtin_amdgcn_readlane((int)ml, src);
                const unsigned mask2 = (unsigned)__builtin_amdgcn_readlane((int)ml, src2);
                const double fs = pm_readlane_f64(fl, src), fs2 = pm_readlane_f64(fl, src2);
                const double q = pm_readlane_f64(ql, src), q2 = pm_readlane_f64(ql, src2);
                if (first && lane == 0) {
                    const double ns = (double)__builtin_popcount(mask);
                    st_pi += q * ns;
                    st_sigma += q * ((fs - P.pil_bar * ns) / P.pre1);
                    if (two) {
                        const double ns2 = (double)__builtin_popcount(mask2);
                        st_pi += q2 * ns2;
                        st_sigma += q2 * ((fs2 - P.pil_bar * ns2) / P.pre1);
                    }
                }
                if (!staged) {  // first significant state: stage y and W^rho[cand] once
#pragma unroll
                    for (int i = 0; i < DPL; ++i) {
                        const int d = lane + 64 * i;
                        y[i] = (d < Dl) ? Y[n * ldy + d0 + d] : 0.0;
                    }
                    for (int j = 0; j < Hp; ++j) {
                        const double *srcw = Wrho + (int64_t)cn[j] * D + d0;
#pragma unroll
                        for (int i = 0; i < DPL; ++i) {
                            const int d = lane + 64 * i;
                            s_wr[j * DS + d] = (d < Dl) ? srcw[d] : 1.0;
                            if (SIGNED) s_wm[j * DS + d] = (d < Dl) ? Wrm1[(int64_t)cn[j] * D + d0 + d] : 1.0;
                        }
                    }
                    wave_sync_lds();
                    staged = true;
                    pfx = 0xFFFFFFFFu;        // (the rows changed under the cached prefix)
                }
                touched |= mask | mask2;
                // T for all of the lane's dimensions, then 2 * DPL independent powers, then the scatter into V
                double T[DPL], T2[DPL], v[DPL], v2[DPL];
                T_of(mask, T);
                T_of(mask2, T2);
                weights(T, q, v);
                weights(T2, q2, v2);
                v_update(mask, q, v);
                if (two) v_update(mask2, q2, v2);
            }
        }
        if (staged) {
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                if ((touched >> j) & 1u) {
                    const int64_t base = (int64_t)cn[j] * D + d0;
#pragma unroll
                    for (int i = 0; i < DPL; ++i) {
                        const int d = lane + 64 * i;
                        if (d < Dl) {
                            const double aid = SIGNED ? V[j][i] : V[j][i] * Wrm1[base + d];  // Aid[j,d] (mca_et.py:309)
                            pm_atomic_add(Wp + base + d, PM_Q(aid * y[i], 1));
                            pm_atomic_add(Wq + base + d, PM_Q(aid, 0));
                        }
                    }
                }
            }
            wave_sync_lds();
        }
    }

    st_pi = pm_wave_sum(st_pi);
    st_sigma = pm_wave_sum(st_sigma);
    st_ld = pm_wave_sum(st_ld);
    st_cnt = pm_wave_sum(st_cnt);
    if (lane == 0) {
        s_red[wave * 4 + 0] = st_pi;
        s_red[wave * 4 + 1] = st_sigma;
        s_red[wave * 4 + 2] = st_ld;
        s_red[wave * 4 + 3] = st_cnt;
    }
    __syncthreads();
    double *g_q1sum = stats + 3 * (int64_t)H * D;
    double *sc = g_q1sum + H;
    if (tid < 4) {
        double v = 0.0;
        for (int w = 0; w < waves; ++w) v += s_red[w * 4 + tid];
        if (v != 0.0) pm_atomic_add(sc + tid, PM_Q(v, tid == 1 ? 3 : tid == 2 ? 4 : 2));    // pi | sum q e | sum lse | count
    }
    for (int h = tid; h < H; h += blockDim.x) {
        const double v = s_q1sum[h];
        if (v != 0.0) pm_atomic_add(g_q1sum + h, v);
    }
}

static int allow_lds_mca(const void *kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// wavefronts per workgroup so that the per-wave LDS areas fit ~64 KB
inline int pick_waves(size_t per_wave_bytes, size_t shared_bytes) {
    int w = 4;
    while (w > 1 && shared_bytes + w * per_wave_bytes > 64 * 1024) w >>= 1;
    return w;
}

inline int64_t grid_waves(int64_t N, int waves) {
    int64_t blocks = (N + waves - 1) / waves;
    const int64_t cap = 256 * 8;
    return blocks < cap ? (blocks < 1 ? 1 : blocks) : cap;
}

}  // namespace

extern "C" int64_t pm_mca_stats_len(int64_t H, int64_t D) {
    return mca_stats_base(H, D) + (PM_XCD_COPIES - 1) * 2 * H * D;
}

static void mca_fold(double *stats, int64_t H, int64_t D, hipStream_t s) {
    const int64_t len = 2 * H * D;
    hipLaunchKernelGGL(pm_fold_copies_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, stats + H * D,
                       stats + mca_stats_base(H, D), len);
}

extern "C" int pm_mca_select_scores_f64(const double *Y, int64_t ldy, const double *W, int64_t ldw, double *R,
                                        int64_t ldr, int64_t N, int64_t H, int64_t D, void *stream) {
    if (N == 0) return PM_OK;
    if (!Y || !W || !R || N < 0 || H <= 0 || D <= 0 || ldy < D || ldw < D || ldr < H) return PM_EINVAL;
    if (H > INT32_MAX || D > INT32_MAX) return PM_ERANGE;
    const int tiles_h = (int)((H + ST - 1) / ST);
    const int64_t blocks = (N + ST - 1) / ST * tiles_h;
    if (blocks > INT32_MAX) return PM_ERANGE;
    hipLaunchKernelGGL(mca_select_scores_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), Y, ldy, W, ldw, R, ldr, N, (int)H, (int)D, tiles_h);
    return (int)hipGetLastError();
}

extern "C" int pm_mca_estep_f64(const double *scores, int64_t lds, const double *wnorm2, const double *ynorm2,
                                const double *Y, int64_t ldy, const double *Wrho, const int32_t *cand,
                                const uint16_t *state_masks, int64_t S, const pm_mca_params *params_host, int64_t N,
                                int64_t H, int64_t D, int64_t Hprime, double *logpj, int64_t ldl, double *lse1,
                                double *lseb, void *stream) {
    if (N == 0) return PM_OK;
    if (!scores || !wnorm2 || !ynorm2 || !Y || !Wrho || !cand || !params_host || !logpj || !lse1 || !lseb || N < 0 ||
        H <= 0 || D <= 0 || Hprime <= 0 || S < 0 || lds < H || ldy < D || ldl < 1 + H + S || (S > 0 && !state_masks))
        return PM_EINVAL;
    if (D > 1024 || Hprime > PM_MAX_HPRIME || Hprime > H || S > 65535) return PM_ERANGE;
    const int dpl = D <= 64 ? 1 : D <= 128 ? 2 : D <= 256 ? 4 : D <= 512 ? 8 : 16;
    const size_t per_wave = sizeof(double) * ((size_t)Hprime * 64 * dpl + S);
    if (per_wave > 150 * 1024) return PM_ERANGE;
    const size_t shared = sizeof(double) * PM_POWTAB_LEN;
    const int waves = pick_waves(per_wave, shared);
    const size_t shmem = shared + per_wave * waves;
    dim3 grid((unsigned)grid_waves(N, waves)), block(64 * waves);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PM_LAUNCH(V)                                                                                            \
    do {                                                                                                        \
        if (int e = allow_lds_mca(reinterpret_cast<const void *>(mca_estep_kernel<V>), shmem)) return e;        \
        hipLaunchKernelGGL(mca_estep_kernel<V>, grid, block, shmem, s, scores, lds, wnorm2, ynorm2, Y, ldy, Wrho, \
                           cand, state_masks, (int)S, *params_host, N, (int)H, (int)D, (int)Hprime, logpj, ldl, \
                           lse1, lseb);                                                                         \
    } while (0)
    switch (dpl) {
        case 1: PM_LAUNCH(1); break;
        case 2: PM_LAUNCH(2); break;
        case 4: PM_LAUNCH(4); break;
        case 8: PM_LAUNCH(8); break;
        default: PM_LAUNCH(16); break;
    }
#undef PM_LAUNCH
    return (int)hipGetLastError();
}

namespace {
template <int DPL, bool SIGNED>
int launch_mstep_hp(int hp, dim3 grid, dim3 block, size_t shmem, hipStream_t s, const double *logpj, int64_t ldl,
                    const double *lse1, const double *lseb, double lse_cut, const double *Y, int64_t ldy,
                    const double *Wrho, const double *Wrm1, const int32_t *cand, const uint16_t *masks, int S,
                    pm_mca_params P, int64_t N, int H, int D, int d0, int Dl, double *q1, int64_t ldq, double *stats) {
    const int Hp = hp;
    hp = hp <= 4 ? 4 : hp <= 8 ? 8 : hp <= 12 ? 12 : 16;
#define PM_CASE(HPV)                                                                                              \
    case HPV: {                                                                                                   \
        if (int e = allow_lds_mca(reinterpret_cast<const void *>(mca_mstep_rows_kernel<DPL, HPV, SIGNED>), shmem)) return e; \
        hipLaunchKernelGGL((mca_mstep_rows_kernel<DPL, HPV, SIGNED>), grid, block, shmem, s, logpj, ldl, lse1, lseb, lse_cut, Y, \
                           ldy, Wrho, Wrm1, cand, masks, S, P, N, H, D, d0, Dl, Hp, q1, ldq, stats);               \
        return (int)hipGetLastError();                                                                            \
    }
    switch (hp) {
        PM_CASE(4) PM_CASE(8) PM_CASE(12)
        case 16:      // (V[16][DPL <= 2]: slabs of 128 observed dimensions)
            if (DPL <= 2) {
                if (int e = allow_lds_mca(reinterpret_cast<const void *>(mca_mstep_rows_kernel<(DPL <= 2 ? DPL : 1), 16, SIGNED>), shmem)) return e;
                hipLaunchKernelGGL((mca_mstep_rows_kernel<(DPL <= 2 ? DPL : 1), 16, SIGNED>), grid, block, shmem, s, logpj, ldl, lse1,
                                   lseb, lse_cut, Y, ldy, Wrho, Wrm1, cand, masks, S, P, N, H, D, d0, Dl, Hp, q1, ldq, stats);
                return (int)hipGetLastError();
            }
            return PM_ERANGE;
        default: return PM_ERANGE;
    }
#undef PM_CASE
}
}  // namespace

extern "C" int pm_mca_mstep_rows_f64(const double *logpj, int64_t ldl, const double *lse1, const double *lseb,
                                     double lse_cut, const double *Y, int64_t ldy, const double *Wrho,
                                     const double *Wrm1, const int32_t *cand, const uint16_t *state_masks, int64_t S,
                                     const pm_mca_params *params_host, int64_t N, int64_t H, int64_t D,
                                     int64_t Hprime, double *q1, int64_t ldq, double *stats, void *stream) {
    if (N == 0) return PM_OK;
    if (!logpj || !lse1 || !lseb || !Y || !Wrho || !Wrm1 || !cand || !params_host || !q1 || !stats || N < 0 ||
        H <= 0 || D <= 0 || Hprime <= 0 || S < 0 || ldl < 1 + H + S || ldy < D || ldq < H || (S > 0 && !state_masks))
        return PM_EINVAL;
    if (D > (1 << 20) || Hprime > PM_MAX_HPRIME || Hprime > H || S > 65535) return PM_ERANGE;
    const int hp_tile = Hprime <= 4 ? 4 : Hprime <= 8 ? 8 : Hprime <= 12 ? 12 : 16;
    // observed dimensions are walked in slabs whose V[HP][DPL] register tile stays within 48 doubles per lane
    const int dpl_max = hp_tile == 16 ? 2 : hp_tile == 12 ? 4 : 8;
    const int slab = 64 * dpl_max;
    const bool sgn = params_host->signed_w != 0.0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int64_t d0 = 0; d0 < D; d0 += slab) {
        const int Dl = (int)((D - d0) < slab ? (D - d0) : slab);
        const int dpl = Dl <= 64 ? 1 : Dl <= 128 ? 2 : Dl <= 256 ? 4 : 8;
        const size_t per_wave = sizeof(double) * ((size_t)hp_tile * 64 * dpl) * (sgn ? 2 : 1);
        const size_t shared = sizeof(double) * (PM_POWTAB_LEN + H + 16);
        const int waves = pick_waves(per_wave, shared);
        const size_t shmem = shared + per_wave * waves;
        if (shmem > 150 * 1024) return PM_ERANGE;
        dim3 grid((unsigned)grid_waves(N, waves)), block(64 * waves);
#define PM_ARGS (int)Hprime, grid, block, shmem, s, logpj, ldl, lse1, lseb, lse_cut, Y, ldy, Wrho, Wrm1, cand, state_masks, \
                (int)S, *params_host, N, (int)H, (int)D, (int)d0, Dl, q1, ldq, stats
        int rc;
        switch (dpl) {
            case 1: rc = sgn ? launch_mstep_hp<1, true>(PM_ARGS) : launch_mstep_hp<1, false>(PM_ARGS); break;
            case 2: rc = sgn ? launch_mstep_hp<2, true>(PM_ARGS) : launch_mstep_hp<2, false>(PM_ARGS); break;
            case 4: rc = sgn ? launch_mstep_hp<4, true>(PM_ARGS) : launch_mstep_hp<4, false>(PM_ARGS); break;
            default: rc = sgn ? launch_mstep_hp<8, true>(PM_ARGS) : launch_mstep_hp<8, false>(PM_ARGS); break;
        }
#undef PM_ARGS
        if (rc) return rc;
    }
    mca_fold(stats, H, D, s);
    return (int)hipGetLastError();
}

namespace {
template <int DPL, bool SIGNED>
int launch_fused_hp(int Hp, dim3 grid, dim3 block, size_t shmem, hipStream_t s, const double *scores, int64_t lds,
                    const double *wnorm2, const double *ynorm2, const double *Y, int64_t ldy, const double *Wrho,
                    const double *Wrm1, const int32_t *cand, const uint16_t *masks, int S, pm_mca_params P, int64_t N,
                    int H, int D, double *logpj, int64_t ldl, double *lse1, double *lseb, double *q1, int64_t ldq,
                    double *stats, double *defer_rec, double *defer_sc) {
    const int hp = Hp <= 4 ? 4 : Hp <= 8 ? 8 : 12;
    // Round 6: unsigned W takes the uniform-exponent power (pm_pow_uni, ROOT = 0) at EVERY rho -- 6.45 ms against 6.74 for the
    // log / exp-free rho = 21 power at config 5 (scratch/mca_T_sweep.py; -DPM_MCA_ROOT21 brings that one back for A/B builds).
    // Signed W (MMCA) keeps its rho = 6 power (pm_pow_m5_6: 9.65 against 10.48 ms, scratch/mmca_time.py -- that pass runs one
    // wavefront per SIMD, where the uniform power's three LDS lookups are not hidden; -DPM_MCA_NO_ROOT6: A/B) and takes the
    // uniform power at every other rho.
#ifdef PM_MCA_ROOT21
    const bool rho21 = !SIGNED && P.inv_rho > 0.0 && fabs(1.0 / P.inv_rho - 21.0) < 1e-9;
#else
    const bool rho21 = false;
#endif
#ifndef PM_MCA_NO_ROOT6
    const bool rho6 = SIGNED && P.inv_rho > 0.0 && fabs(1.0 / P.inv_rho - 6.0) < 1e-9;
#else
    const bool rho6 = false;
#endif
#define PM_LAUNCH_F(HPV, SG, RT, DF)                                                                                  \
    do {                                                                                                              \
        if (DF) {                                                                                                     \
            if (int e = allow_lds_mca(reinterpret_cast<const void *>(mca_estep_fused_defer_kernel<DPL, HPV, SG, RT>), shmem)) \
                return e;                                                                                             \
            hipLaunchKernelGGL((mca_estep_fused_defer_kernel<DPL, HPV, SG, RT>), grid, block, shmem, s, scores, lds, wnorm2, \
                               ynorm2, Y, ldy, Wrho, Wrm1, cand, masks, S, P, N, H, D, Hp, logpj, ldl, lse1, lseb, q1, ldq, \
                               stats, defer_rec, defer_sc);                                                           \
        } else {                                                                                                      \
            if (int e = allow_lds_mca(reinterpret_cast<const void *>(mca_estep_fused_kernel<DPL, HPV, SG, RT>), shmem)) \
                return e;                                                                                             \
            hipLaunchKernelGGL((mca_estep_fused_kernel<DPL, HPV, SG, RT>), grid, block, shmem, s, scores, lds, wnorm2, \
                               ynorm2, Y, ldy, Wrho, Wrm1, cand, masks, S, P, N, H, D, Hp, logpj, ldl, lse1, lseb, q1, ldq, \
                               stats);                                                                                \
        }                                                                                                             \
        return (int)hipGetLastError();                                                                                \
    } while (0)
#ifdef PM_MCA_ROOT21
#define PM_CASE21(HPV)                                                \
    if (rho21) {                                                      \
        if (defer_rec) PM_LAUNCH_F(HPV, false, 21, true);             \
        PM_LAUNCH_F(HPV, false, 21, false);                           \
    }
#else
#define PM_CASE21(HPV)
#endif
#ifndef PM_MCA_NO_ROOT6
#define PM_CASE6(HPV)                                                 \
    if (rho6) {                                                       \
        if (defer_rec) PM_LAUNCH_F(HPV, true, 6, true);               \
        PM_LAUNCH_F(HPV, true, 6, false);                             \
    }
#else
#define PM_CASE6(HPV)
#endif
#define PM_CASE(HPV)                                                  \
    case HPV: {                                                       \
        PM_CASE21(HPV)                                                \
        PM_CASE6(HPV)                                                 \
        if (defer_rec) PM_LAUNCH_F(HPV, SIGNED, 0, true);             \
        PM_LAUNCH_F(HPV, SIGNED, 0, false);                           \
    }
    switch (hp) {
        PM_CASE(4) PM_CASE(8) PM_CASE(12)
        default: return PM_ERANGE;
    }
#undef PM_CASE
#undef PM_CASE21
#undef PM_CASE6
#undef PM_LAUNCH_F
    (void)rho21;
    (void)rho6;
}
}  // namespace

extern "C" int pm_mca_estep_mstats_f64(const double *scores, int64_t lds, const double *wnorm2, const double *ynorm2,
                                       const double *Y, int64_t ldy, const double *Wrho, const double *Wrm1,
                                       const int32_t *cand, const uint16_t *state_masks, int64_t S,
                                       const pm_mca_params *params_host, int64_t N, int64_t H, int64_t D,
                                       int64_t Hprime, double *logpj, int64_t ldl, double *lse1, double *lseb,
                                       double *q1, int64_t ldq, double *stats, void *stream) {
    return pm_mca_estep_mstats_defer_f64(scores, lds, wnorm2, ynorm2, Y, ldy, Wrho, Wrm1, cand, state_masks, S, params_host, N,
                                         H, D, Hprime, logpj, ldl, lse1, lseb, q1, ldq, stats, nullptr, nullptr, stream);
}

namespace {
// The deferred statistics of a data-truncation step (mca_estep_fused_kernel<.., DEFER = true>), added once the cut is known
// (mca_et.py:250-258: the stabilised log-denominators of the annealed weights; the cut is a DEVICE double, the radix
// select's result never visits the host).  Two kernels:
//  * mca_defer_q1_kernel, one wavefront per datapoint: the q1 row of a dropped datapoint is zeroed (the G1 = q1^T Y product
//    behind sees kept datapoints only), a kept one's goes into q1sum (register accumulators: a lane owns latents lane,
//    lane + 64, ...) and its three scalars into the sums;
//  * mca_defer_scatter_kernel: Wq[c_j, d] += Aid[j, d], Wp[c_j, d] += Aid[j, d] y_d over the kept datapoints.  Global f64
//    atomics sustain ~0.1 T/s on this chip -- 2 H' D of them per datapoint took 2.7 ms at config 5 (the first form of this
//    kernel) -- so a workgroup of sixteen wavefronts owns a slice of SW observed dimensions for 1/G of the datapoints,
//    accumulates the H x SW slices of Wp and Wq in LDS (ds_add_f64; 128 KB) and flushes them once: 2 H D atomics per
//    workgroup instead of per datapoint.  Bound by reading the records once (N_use H' D doubles).
__global__ __launch_bounds__(256) void mca_defer_q1_kernel(const double *__restrict__ lseb, const double *__restrict__ cut_dev,
                                                            const double *__restrict__ sc_in, double *__restrict__ q1,
                                                            int64_t ldq, double *__restrict__ stats, int64_t N, int H, int D) {
    __shared__ double s_q1sum[512];
    __shared__ double s_red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int h = tid; h < 512; h += 256) s_q1sum[h] = 0.0;
    __syncthreads();
    const double cut = cut_dev[0];
    double st_pi = 0.0, st_sigma = 0.0, st_ld = 0.0, st_cnt = 0.0;
    double qs[8];                                  // H <= 512: latents lane + 64 i
#pragma unroll
    for (int i = 0; i < 8; ++i) qs[i] = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        double *qrow = q1 + n * ldq;
        if (!(lseb[n] >= cut)) {
            for (int h = lane; h < H; h += 64) qrow[h] = 0.0;
            continue;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int h = lane + 64 * i;
            if (h < H) qs[i] += PM_Q(qrow[h], 2);
        }
        if (lane == 0) {
            const double *sc_n = sc_in + 4 * n;
            st_pi += sc_n[0];
            st_sigma += sc_n[1];
            st_ld += sc_n[2];
            st_cnt += 1.0;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int h = lane + 64 * i;
        if (h < H && qs[i] != 0.0) atomicAdd(&s_q1sum[h], qs[i]);
    }
    if (lane == 0) {
        s_red[wave][0] = st_pi;
        s_red[wave][1] = st_sigma;
        s_red[wave][2] = st_ld;
        s_red[wave][3] = st_cnt;
    }
    __syncthreads();
    double *g_q1sum = stats + 3 * (int64_t)H * D;
    double *sc = g_q1sum + H;
    if (tid < 4) {
        const double v = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
        if (v != 0.0) pm_atomic_add(sc + tid, PM_Q(v, tid == 1 ? 3 : tid == 2 ? 4 : 2));    // pi | sum q e | sum lse | count
    }
    for (int h = tid; h < H; h += 256) {
        const double v = s_q1sum[h];
        if (v != 0.0) pm_atomic_add(g_q1sum + h, v);
    }
}

constexpr int DEFER_GROUPS = 64;      // datapoint groups of the scatter kernel (workspace: one [Wp | Wq] partial per group)

// A workgroup of sixteen wavefronts owns HR rows (latents) of Wp and Wq -- all D observed dimensions of them, in LDS -- for
// 1/G of the datapoints.  [First form: slices of 64 observed dimensions of ALL latents per workgroup -- every 2 KB record row
// was then read in four 512-byte pieces by four workgroups at four different times, and the kernel ran at 2 TB/s; a
// workgroup that reads the whole rows whose latent it owns streams them.]  A wavefront takes the log-denominators and the
// candidates of 64 of its datapoints at once (lane = datapoint: an 8-bit mask of the candidate positions whose latent is in
// the workgroup's range, 0 for a dropped datapoint), then walks the non-empty ones: the datapoint's y row once, each owned
// record row in full (DPL doubles per lane), two ds_add_f64 per element.  [Measured on top of this form and not kept, all
// within 0.46-0.51 ms: two datapoints per trip, two trips in flight (a software pipeline over the walk; with the candidates kept
// in registers and read by v_readlane it spills at 128 VGPRs: 0.58), the candidates in registers alone (0.46), and the ablations -- no
// atomics: the same time; no record loads: half.  The kernel moves 1.5 GB (1.03 GB of kept records + the y rows once per latent
// range) at 3 TB/s with one sixteen-wavefront workgroup per CU; a plain torch.sum streams the same buffer at 5.7
// (scratch/bw_probe.py).]
template <int DPL>      // doubles per lane and row: D <= 64 DPL
__global__ __launch_bounds__(1024) void mca_defer_scatter_kernel(const double *__restrict__ lseb, const double *__restrict__ cut_dev,
                                                                  const double *__restrict__ Y, int64_t ldy,
                                                                  const int32_t *__restrict__ cand,
                                                                  const double *__restrict__ rec, double *__restrict__ part,
                                                                  int64_t N, int H, int D, int Hp, int HR, int nranges) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int DS = 64 * DPL;
    double *s_wp = reinterpret_cast<double *>(smem_raw), *s_wq = s_wp + (size_t)HR * DS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * HR * DS; i += 1024) s_wp[i] = 0.0;
    __syncthreads();
    const double cut = cut_dev[0];
    const int range = (int)(blockIdx.x % (unsigned)nranges), h0 = range * HR;
    const int64_t group = blockIdx.x / (unsigned)nranges, G = gridDim.x / (unsigned)nranges;
    const int64_t per = (N + G - 1) / G, n_lo = group * per, n_hi = (n_lo + per < N) ? n_lo + per : N;
    const int64_t first = n_lo + wave;
    const int64_t cnt = first < n_hi ? (n_hi - first + 15) / 16 : 0;          // this wavefront's datapoints: first + 16 p
    for (int64_t p0 = 0; p0 < cnt; p0 += 64) {
        const int64_t p = p0 + lane;
        unsigned mine = 0u;
        if (p < cnt) {
            const int64_t n = first + 16 * p;
            if (lseb[n] >= cut) {
                const int32_t *cn = cand + n * Hp;
                for (int j = 0; j < Hp; ++j) mine |= ((unsigned)(cn[j] - h0) < (unsigned)HR) ? (1u << j) : 0u;
            }
        }
        unsigned long long todo = __ballot(mine != 0u);
        while (todo) {
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            unsigned m = (unsigned)__builtin_amdgcn_readlane((int)mine, b);
            const int64_t n = first + 16 * (p0 + b);
            const int32_t *cn = cand + n * Hp;
            double y[DPL];
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                y[i] = d < D ? Y[n * ldy + d] : 0.0;
            }
            while (m) {          // up to two owned rows per trip: their loads leave together
                const int j0 = __builtin_ctz(m);
                m &= m - 1u;
                const bool two = m != 0u;
                const int j1 = two ? __builtin_ctz(m) : j0;
                m &= m - 1u;      // (m == 0 stays 0)
                const int r0 = __builtin_amdgcn_readfirstlane(cn[j0]) - h0, r1 = __builtin_amdgcn_readfirstlane(cn[j1]) - h0;
                const double *a0 = rec + (n * Hp + j0) * (int64_t)D, *a1 = rec + (n * Hp + j1) * (int64_t)D;
                double v0[DPL], v1[DPL];
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    const int d = lane + 64 * i;
                    v0[i] = d < D ? a0[d] : 0.0;
                    v1[i] = (two && d < D) ? a1[d] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    if (v0[i] != 0.0 && (PM_SCATTER_ABL != 1 || v0[i] == 1.2345e-300)) {
                        atomicAdd(&s_wq[r0 * DS + lane + 64 * i], PM_Q(v0[i], 0));
                        atomicAdd(&s_wp[r0 * DS + lane + 64 * i], PM_Q(v0[i] * y[i], 1));
                    }
                    if (v1[i] != 0.0 && (PM_SCATTER_ABL != 1 || v1[i] == 1.2345e-300)) {
                        atomicAdd(&s_wq[r1 * DS + lane + 64 * i], PM_Q(v1[i], 0));
                        atomicAdd(&s_wp[r1 * DS + lane + 64 * i], PM_Q(v1[i] * y[i], 1));
                    }
                }
            }
        }
    }
    __syncthreads();
    // this group's partial [Wp | Wq] (H x D each): plain stores -- mca_defer_reduce_kernel sums the groups in a fixed order
    double *pw = part + group * 2 * (int64_t)H * D;
    for (int i = tid; i < HR * DS; i += 1024) {
        const int r = i / DS, d = i % DS, h = h0 + r;
        if (h < H && d < D) {
            pw[(int64_t)h * D + d] = s_wp[i];
            pw[(int64_t)(H + h) * D + d] = s_wq[i];
        }
    }
}

// out[i] += sum_g part[g][i], g in a fixed order (out = the [Wp_m | Wq_m] block of the statistics: 2 H D doubles)
__global__ __launch_bounds__(256) void mca_defer_reduce_kernel(const double *__restrict__ part, int G, double *__restrict__ out,
                                                                int64_t len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int g = 0;
    for (; g + 4 <= G; g += 4) {
        a0 += part[(int64_t)g * len + i];
        a1 += part[(int64_t)(g + 1) * len + i];
        a2 += part[(int64_t)(g + 2) * len + i];
        a3 += part[(int64_t)(g + 3) * len + i];
    }
    for (; g < G; ++g) a0 += part[(int64_t)g * len + i];
    out[i] += (a0 + a1) + (a2 + a3);
}
}  // namespace

extern "C" int64_t pm_mca_defer_apply_work_len(int64_t H, int64_t D) {
    return (H > 0 && D > 0) ? (int64_t)DEFER_GROUPS * 2 * H * D : 0;
}

extern "C" int pm_mca_defer_apply_f64(const double *lseb, const double *cut, const double *Y, int64_t ldy, const int32_t *cand,
                                      const double *records, const double *scalars, double *q1, int64_t ldq, double *stats,
                                      double *work, int64_t N, int64_t H, int64_t D, int64_t Hprime, void *stream) {
    if (N == 0) return PM_OK;
    if (!lseb || !cut || !Y || !cand || !records || !scalars || !q1 || !stats || !work || N < 0 || H <= 0 || D <= 0 ||
        Hprime <= 0 || ldy < D || ldq < H)
        return PM_EINVAL;
    if (H > 512 || D > 512 || Hprime > 12) return PM_ERANGE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t blocks = (N + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mca_defer_q1_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lseb, cut, scalars, q1, ldq, stats, N,
                       (int)H, (int)D);
    const int dpl = D <= 64 ? 1 : D <= 128 ? 2 : D <= 256 ? 4 : 8;
    int hr = PM_SCATTER_LDS_DOUBLES / (2 * 64 * dpl);          // rows of Wp and Wq per workgroup: 32 at D <= 256
    if (hr > H) hr = (int)H;
    const int nranges = (int)((H + hr - 1) / hr);
    int64_t groups = DEFER_GROUPS;
    if (groups > (N + 255) / 256) groups = (N + 255) / 256;
    const size_t shmem = (size_t)2 * hr * 64 * dpl * sizeof(double);
#define PM_SCATTER(DPLV)                                                                                                  \
    do {                                                                                                                  \
        if (int e = allow_lds_mca(reinterpret_cast<const void *>(mca_defer_scatter_kernel<DPLV>), shmem)) return e;       \
        hipLaunchKernelGGL((mca_defer_scatter_kernel<DPLV>), dim3((unsigned)(groups * nranges)), dim3(1024), shmem, s, lseb, cut, \
                           Y, ldy, cand, records, work, N, (int)H, (int)D, (int)Hprime, hr, nranges);                     \
    } while (0)
    if (dpl == 1) PM_SCATTER(1);
    else if (dpl == 2) PM_SCATTER(2);
    else if (dpl == 4) PM_SCATTER(4);
    else PM_SCATTER(8);
#undef PM_SCATTER
    const int64_t len = 2 * H * D;
    hipLaunchKernelGGL(mca_defer_reduce_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, (const double *)work,
                       (int)groups, stats + H * D, len);
    return (int)hipGetLastError();
}

extern "C" int pm_mca_estep_mstats_defer_f64(const double *scores, int64_t lds, const double *wnorm2, const double *ynorm2,
                                             const double *Y, int64_t ldy, const double *Wrho, const double *Wrm1,
                                             const int32_t *cand, const uint16_t *state_masks, int64_t S,
                                             const pm_mca_params *params_host, int64_t N, int64_t H, int64_t D,
                                             int64_t Hprime, double *logpj, int64_t ldl, double *lse1, double *lseb,
                                             double *q1, int64_t ldq, double *stats, double *defer_rec, double *defer_sc,
                                             void *stream) {
    if ((defer_rec == nullptr) != (defer_sc == nullptr)) return PM_EINVAL;
    if (N == 0) return PM_OK;
    if (!scores || !wnorm2 || !ynorm2 || !Y || !Wrho || !Wrm1 || !cand || !params_host || !logpj || !lse1 || !lseb ||
        !q1 || !stats || N < 0 || H <= 0 || D <= 0 || Hprime <= 0 || S < 0 || lds < H || ldy < D ||
        ldl < 1 + H + S || ldq < H || (S > 0 && !state_masks))
        return PM_EINVAL;
    if (D > 512 || Hprime > 12 || Hprime > H || S > 65535) return PM_ERANGE;
    const int dpl = D <= 64 ? 1 : D <= 128 ? 2 : D <= 256 ? 4 : 8;
    const int hp_tile = Hprime <= 4 ? 4 : Hprime <= 8 ? 8 : 12;
    if ((int64_t)dpl * hp_tile > 48) return PM_ERANGE;  // V[HP][DPL] register tile
    const bool sgn = params_host->signed_w != 0.0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t per_wave = sizeof(double) * ((size_t)hp_tile * 64 * dpl * (sgn ? 2 : 1) + S);
    const size_t shared = sizeof(double) * (PM_POWTAB_LEN + PM_FUSED_RT_LEN + H + 16);
    const int waves = pick_waves(per_wave, shared);
    const size_t shmem = shared + per_wave * waves;
    if (shmem > 150 * 1024) return PM_ERANGE;
    dim3 grid((unsigned)grid_waves(N, waves)), block(64 * waves);
#define PM_ARGS (int)Hprime, grid, block, shmem, s, scores, lds, wnorm2, ynorm2, Y, ldy, Wrho, Wrm1, cand, state_masks, \
                (int)S, *params_host, N, (int)H, (int)D, logpj, ldl, lse1, lseb, q1, ldq, stats, defer_rec, defer_sc
    int rc;
    switch (dpl) {
        case 1: rc = sgn ? launch_fused_hp<1, true>(PM_ARGS) : launch_fused_hp<1, false>(PM_ARGS); break;
        case 2: rc = sgn ? launch_fused_hp<2, true>(PM_ARGS) : launch_fused_hp<2, false>(PM_ARGS); break;
        case 4: rc = sgn ? launch_fused_hp<4, true>(PM_ARGS) : launch_fused_hp<4, false>(PM_ARGS); break;
        default: rc = sgn ? launch_fused_hp<8, true>(PM_ARGS) : launch_fused_hp<8, false>(PM_ARGS); break;
    }
#undef PM_ARGS
    if (rc) return rc;
    mca_fold(stats, H, D, s);
    return (int)hipGetLastError();
}

// The element-wise W update of MCA_ET.M_step (mca_et.py:333-348) from the all-reduced statistics
// [G1 (H,D) | Wp_m (H,D) | Wq_m (H,D) | q1sum (H)]:  Wp = G1 W^2 + Wp_m,  Wq = q1sum_h W^2 + Wq_m,  W_new = Wp / Wq with
// Wq < tiny -> 0 / tiny (the reference's guard against a division by zero) -- one launch instead of eleven tensor operations;
// `wt_clamped` (optional): max(W_new, w_tol), what check_params makes of it at the top of the next step (the selection
// seeded behind the download ranks that).
namespace {
__global__ __launch_bounds__(256) void mca_wupdate_kernel(const double *__restrict__ stats, const double *__restrict__ wt,
                                                          int H, int D, double w_tol, double *__restrict__ wt_new,
                                                          double *__restrict__ wt_clamped) {
    const int64_t HD = (int64_t)H * D;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= HD) return;
    const double tiny = 2.2250738585072014e-308;
    const double w = wt[e], wsq = w * w;
    double wp = stats[e] * wsq + stats[HD + e];
    double wq = stats[3 * HD + e / D] * wsq + stats[2 * HD + e];
    if (wq < tiny) {
        wp = 0.0;
        wq = tiny;
    }
    const double r = wp / wq;
    wt_new[e] = r;
    if (wt_clamped) wt_clamped[e] = fmax(r, w_tol);
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// The per-step tables of W (mca_et.py:218-227, mmca_et.py:250-260): W^T as given, sign(W)|W|^rho, |W|^(rho-1) and |W_h|^2
// -- on the device, so that an EM loop never waits for the host's log / exp over H x D elements and a 3 H D upload between
// the download of the new W and the next E-step (0.2-0.4 ms of idle device per step on the pool's slower hosts).  One
// workgroup per latent; the row norm in a fixed tree (same bits every time).
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void mca_tables_kernel(const double *__restrict__ wt, int D, double rho,
                                                         double *__restrict__ tabs, int64_t plane,
                                                         double *__restrict__ wnorm2) {
    __shared__ double s_part[4];
    const int h = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    for (int d = tid; d < D; d += 256) {
        const double w = wt[(int64_t)h * D + d];
        const double lw = log(fabs(w));
        const double wr = exp(rho * lw);
        tabs[(int64_t)h * D + d] = w;
        tabs[plane + (int64_t)h * D + d] = copysign(wr, w);
        tabs[2 * plane + (int64_t)h * D + d] = exp((rho - 1.0) * lw);
        acc = fma(w, w, acc);
    }
    acc = pm_wave_sum(acc);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) wnorm2[h] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}
}  // namespace

extern "C" int pm_mca_tables_f64(const double *wt, int64_t H, int64_t D, double rho, double *tabs, double *wnorm2,
                                 void *stream) {
    if (!wt || !tabs || !wnorm2 || H <= 0 || D <= 0 || !(rho > 0.0)) return PM_EINVAL;
    if (D > INT32_MAX || H > INT32_MAX) return PM_ERANGE;
    hipLaunchKernelGGL(mca_tables_kernel, dim3((unsigned)H), dim3(256), 0, static_cast<hipStream_t>(stream), wt, (int)D, rho,
                       tabs, H * D, wnorm2);
    return (int)hipGetLastError();
}

extern "C" int pm_mca_w_update_f64(const double *stats, const double *wt, int64_t H, int64_t D, double w_tol, double *wt_new,
                                   double *wt_clamped, void *stream) {
    if (!stats || !wt || !wt_new || H <= 0 || D <= 0) return PM_EINVAL;
    const int64_t HD = H * D;
    if (HD > (int64_t)INT32_MAX * 256) return PM_ERANGE;
    hipLaunchKernelGGL(mca_wupdate_kernel, dim3((unsigned)((HD + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       stats, wt, (int)H, (int)D, w_tol, wt_new, wt_clamped);
    return (int)hipGetLastError();
}

PM_DET_SETTER(mca)
