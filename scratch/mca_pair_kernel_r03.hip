// Round 3's two-wavefronts-per-datapoint variant of mca_estep_fused_kernel, measured slower (8.93 vs 8.03 ms at config 5,
// profiles/r03_mca_pair.txt) and removed from prosper_amd/csrc/mca_kernels.hip in round 4.  Not compiled; kept for the
// record.  It was a section of mca_kernels.hip (uses that file's helpers) plus the launch code at the end of this file.
#ifdef PM_MCA_PAIR
// ---------------------------------------------------------------------------------------------
// The same pass with TWO wavefronts per datapoint (unsigned W, D <= 256): wavefront `half` owns the observed dimensions
// [half 64 DPW, (half + 1) 64 DPW) -- its halves of the candidates' W^rho rows (private LDS), of y and of the Aid
// accumulators V -- so a wavefront carries V[HP][DPW] instead of V[HP][2 DPW]: <= 128 registers and 9.6 KB of LDS per
// wavefront, FOUR wavefronts per SIMD instead of two.  The two halves of a state's squared error meet in LDS: GS states
// are evaluated back to back (their |T|^(1/rho - 1) kept in registers), one exchange + one workgroup barrier per group,
// then both wavefronts form the same weights from the same sums (bit-identical: a + b in the same order) and update
// their V halves.  Wavefront 0 writes the log-joints and the per-datapoint statistics; lb travels through LDS.
// (Round 2 dismissed the split on the count of independent power chains per wavefront; the verdict asked to measure it.)
// MEASURED (round 3, bench state, scratch/mca_ab.sh with -DPM_MCA_PAIR): 8.93 ms against 8.03 ms for one wavefront per
// datapoint (GS = 1, 124 registers, 4 wavefronts per SIMD; GS = 2: 9.24; compiled for 3 per SIMD: 11.3-12.3).  The
// per-state work that does not shrink with the dimensions -- mask handling, the wave reduction, the weight's exponential,
// the V-update branches, plus the address arithmetic of a second wavefront -- is now paid twice: 3.81 G instead of
// 2.54 G VALU wave-instructions per launch (SQ_INSTS_VALU, profiles/r03_mca_pair.txt).  The deeper occupancy raises the
// issue rate from 0.51 to 0.69 of peak and the duplicated work takes more than that back.  Compiled only with
// -DPM_MCA_PAIR.
// ---------------------------------------------------------------------------------------------
template <int DPW, int HP, int GS>
__global__ __launch_bounds__(128, PM_MCA_PAIR_WPE) void mca_estep_fused_pair_kernel(
    const double *__restrict__ scores, int64_t lds, const double *__restrict__ wnorm2, const double *__restrict__ ynorm2,
    const double *__restrict__ Y, int64_t ldy, const double *__restrict__ Wrho, const double *__restrict__ Wrm1,
    const int32_t *__restrict__ cand, const uint16_t *__restrict__ masks, int S, pm_mca_params P, int64_t N, int H, int D,
    int Hp, double *__restrict__ logpj, int64_t ldl, double *__restrict__ lse1, double *__restrict__ lseb,
    double *__restrict__ q1, int64_t ldq, double *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ power tables | q1sum (H) | x (2 buffers x 2 halves x GS) | lb (2) | red (8) | wr: 2 halves x HP x DSW | e (S) ]
    constexpr int DSW = 64 * DPW;
    const int tid = threadIdx.x, lane = tid & 63, half = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *s_tab = reinterpret_cast<double *>(smem);
    double *s_q1sum = s_tab + PM_POWTAB_LEN;
    double *s_x = s_q1sum + H;
    double *s_lb = s_x + 4 * GS;
    double *s_red = s_lb + 2;
    double *s_wr = s_red + 8 + (size_t)half * HP * DSW;
    double *s_e = s_red + 8 + 2 * (size_t)HP * DSW;
    pm_load_powtab(s_tab, tid, blockDim.x);
    for (int h = tid; h < H; h += blockDim.x) s_q1sum[h] = 0.0;
    __syncthreads();

    double *Wp = pm_xcd_copy(stats + (int64_t)H * D, stats + mca_stats_base(H, D), 2 * (int64_t)H * D);
    double *Wq = Wp + (int64_t)H * D;
    double st_pi = 0.0, st_sigma = 0.0, st_ld = 0.0, st_cnt = 0.0;
    const int dbase = half * DSW;

    for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
        const int32_t *cn = cand + n * Hp;
        double y[DPW];
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int d = dbase + lane + 64 * i;
            y[i] = (d < D) ? Y[n * ldy + d] : 0.0;
        }
        for (int j = 0; j < HP; ++j) {
            const bool have = j < Hp;
            const int64_t base = have ? (int64_t)cn[j] * D : 0;
#pragma unroll
            for (int i = 0; i < DPW; ++i) {
                const int d = dbase + lane + 64 * i;
                s_wr[j * DSW + lane + 64 * i] = (have && d < D) ? Wrho[base + d] : 0.0;
            }
        }
        wave_sync_lds();

        double V[HP][DPW];
#pragma unroll
        for (int j = 0; j < HP; ++j)
#pragma unroll
            for (int i = 0; i < DPW; ++i) V[j][i] = 0.0;
        double M = -INFINITY;     // lazily updated reference level of beta f_s (see mca_estep_fused_kernel)

        for (int s0 = 0; s0 < S; s0 += GS) {
            double wb[GS][DPW], part[GS];
            unsigned mk[GS];
#pragma unroll
            for (int q = 0; q < GS; ++q) {
                const int st = s0 + q;
                mk[q] = st < S ? (unsigned)__builtin_amdgcn_readfirstlane((int)masks[st]) : 0u;
                double T[DPW];
#pragma unroll
                for (int i = 0; i < DPW; ++i) T[i] = 0.0;
#pragma unroll
                for (int j = 0; j < HP; ++j)
                    if ((mk[q] >> j) & 1u) {
#pragma unroll
                        for (int i = 0; i < DPW; ++i) T[i] += s_wr[j * DSW + lane + 64 * i];
                    }
                part[q] = 0.0;
#pragma unroll
                for (int i = 0; i < DPW; ++i) {
                    const double r = pm_pow_tab(T[i], P.inv_rho - 1.0, s_tab);      // T = 0: padding, discarded below
                    const double wbar = (T[i] > 0.0) ? T[i] * r : 0.0;
                    const double df = wbar - y[i];
                    part[q] = fma(df, df, part[q]);
                    wb[q][i] = (T[i] > 0.0) ? r : 0.0;
                }
            }
            const int buf = (s0 / GS) & 1;
#pragma unroll
            for (int q = 0; q < GS; ++q) {
                const double p = pm_wave_sum_dpp(part[q]);
                if (lane == 0) s_x[(buf * 2 + half) * GS + q] = p;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < GS; ++q) {
                const int st = s0 + q;
                if (st < S) {                                                   // uniform
                    const double tot = s_x[(buf * 2 + 0) * GS + q] + s_x[(buf * 2 + 1) * GS + q];
                    if (tid == 0) s_e[st] = tot;
                    const double bf = P.beta * (P.pil_bar * (double)__builtin_popcount(mk[q]) + P.pre1 * tot);
                    double w = pm_exp_tab(bf - M, s_tab);
                    if (bf > M + 50.0) {                                        // uniform; the first state, then hardly ever
                        const double sc = exp(M - bf);
#pragma unroll
                        for (int j = 0; j < HP; ++j)
#pragma unroll
                            for (int i = 0; i < DPW; ++i) V[j][i] *= sc;
                        M = bf;
                        w = 1.0;
                    }
#pragma unroll
                    for (int j = 0; j < HP; ++j)
                        if ((mk[q] >> j) & 1u) {
#pragma unroll
                            for (int i = 0; i < DPW; ++i) V[j][i] = fma(w, wb[q][i], V[j][i]);
                        }
                }
            }
        }
        __syncthreads();                                   // every s_e entry is in LDS

        const double yn = ynorm2[n];
        if (half == 0) {
            // log-pseudo-joints, the two log-evidences and the singletons' statistics (as mca_estep_fused_kernel)
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
                s_lb[0] = lb;
            }
            double *qrow = q1 + n * ldq;
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
                    atomicAdd(&s_q1sum[h], q);
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
        __syncthreads();
        const double lbv = s_lb[0];
        const double g = exp(M - lbv);                     // <= 1: every multi-cause beta*f_s is <= lb
        if (S > 0 && g != 0.0) {
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                if (j < Hp) {
                    const int64_t base = (int64_t)cn[j] * D;
#pragma unroll
                    for (int i = 0; i < DPW; ++i) {
                        const int d = dbase + lane + 64 * i;
                        if (d < D) {
                            const double aid = V[j][i] * g * Wrm1[base + d];
                            if (aid != 0.0) {
                                pm_atomic_add(Wp + base + d, aid * y[i]);
                                pm_atomic_add(Wq + base + d, aid);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();       // s_lb, s_e and the row halves are rewritten by the next datapoint
    }

    st_pi = pm_wave_sum(st_pi);
    st_sigma = pm_wave_sum(st_sigma);
    st_ld = pm_wave_sum(st_ld);
    st_cnt = pm_wave_sum(st_cnt);
    if (lane == 0) {
        s_red[half * 4 + 0] = st_pi;
        s_red[half * 4 + 1] = st_sigma;
        s_red[half * 4 + 2] = st_ld;
        s_red[half * 4 + 3] = st_cnt;
    }
    __syncthreads();
    double *g_q1sum = stats + 3 * (int64_t)H * D;
    double *sc = g_q1sum + H;
    if (tid < 4) {
        const double v = s_red[tid] + s_red[4 + tid];
        if (v != 0.0) pm_atomic_add(sc + tid, v);
    }
    for (int h = tid; h < H; h += blockDim.x) {
        const double v = s_q1sum[h];
        if (v != 0.0) pm_atomic_add(g_q1sum + h, v);
    }
}

#endif  // PM_MCA_PAIR

// ---- launch code (inside pm_mca_estep_mstats_f64) ----
#ifdef PM_MCA_PAIR      // (measured slower than one wavefront per datapoint: see the kernel's header; off unless asked for)
    if (!sgn && D <= 256 && hp_tile <= 8) {
        // two wavefronts per datapoint (mca_estep_fused_pair_kernel): one workgroup of 128 threads per datapoint in flight
        constexpr int GS = PM_MCA_PAIR_GS;
        const int dpw = D <= 128 ? 1 : 2, hpv = hp_tile;
        const size_t shmem2 = sizeof(double) * (PM_POWTAB_LEN + H + 4 * GS + 2 + 8 + 2 * (size_t)hpv * 64 * dpw + S);
        int64_t blocks = N;
        const int64_t cap = 256 * 7;                    // seven resident workgroups per CU (21 KB of LDS each)
        if (blocks > cap) blocks = cap;
#define PM_PAIR(DPWV, HPV)                                                                                            \
    do {                                                                                                              \
        if (int e = allow_lds_mca(reinterpret_cast<const void *>(mca_estep_fused_pair_kernel<DPWV, HPV, GS>), shmem2)) \
            return e;                                                                                                 \
        hipLaunchKernelGGL((mca_estep_fused_pair_kernel<DPWV, HPV, GS>), dim3((unsigned)blocks), dim3(128), shmem2, s, \
                           scores, lds, wnorm2, ynorm2, Y, ldy, Wrho, Wrm1, cand, state_masks, (int)S, *params_host, N, \
                           (int)H, (int)D, (int)Hprime, logpj, ldl, lse1, lseb, q1, ldq, stats);                      \
    } while (0)
        if (dpw == 1 && hpv == 4) PM_PAIR(1, 4);
        else if (dpw == 1) PM_PAIR(1, 8);
        else if (hpv == 4) PM_PAIR(2, 4);
        else PM_PAIR(2, 8);
#undef PM_PAIR
        if (int e = (int)hipGetLastError()) return e;
        mca_fold(stats, H, D, s);
        return (int)hipGetLastError();
    }
#endif
