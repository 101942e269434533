// Gaussian (spike-and-slab) Sparse Coding, scalar observation noise (prosper/em/camodels/gsc_et.py,
// class GSC, sigma_sq_type == 'scalar') on gfx950.
//
// One pass per datapoint over the truncated state set computes, from the scores a = W^T y (f64 MFMA
// GEMM), the Gram matrix G = W^T W and psi_sq:
//   component scores + top-H' (gsc_et.py:721-728, 752-809), candidates sorted by latent index
//   every state's un-normalised posterior weight exp(beta * lp)       (gsc_et.py:316-356, 476-522)
//     Lambda = G_aa / s2 + Psi_a^-1,  b = a_a - G_aa mu_a,  |r|^2 = |y|^2 - 2 mu_a.a_a + mu_a^T G_aa mu_a
//     lp = -(logdet Psi_a + logdet Lambda) - |r|^2 / s2 + b^T Lambda^-1 b / s2^2 + sum logit(pi_a)
//     kappa = Lambda^-1 b / s2 + mu_a,  E[z z^T] = kappa kappa^T + Lambda^-1
//   the normalised expectations xpt_s, xpt_sz (N,H) and the SUMS over datapoints of xpt_ss, xpt_szsz
//   (H,H) -- the reference materialises those per datapoint as (N,H,H) (26 GB each at config 4,
//   gsc_et.py:436-438) although its M-step only ever consumes their sums (gsc_et.py:603-610,662-671).
// Weights follow the reference exactly: un-stabilised exp, NaN / underflow clamped to `tiny`, the null
// state's weight not clamped.
//
// Mapping: 16 lanes per datapoint (four datapoints per wavefront), lane j holds latents h = j + 16 i;
// a lane walks multi-cause states s = j + 16 t and solves their g x g systems (g <= GMAX) in registers;
// per-datapoint accumulators live in LDS; reductions over a datapoint are DPP row butterflies.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

constexpr int ROWS = 16;

template <int CTRL>
__device__ __forceinline__ unsigned gdpp32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t gdpp64(uint64_t v) {
    const unsigned lo = gdpp32<CTRL>((unsigned)v), hi = gdpp32<CTRL>((unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ double gdppf(double v) {
    return __longlong_as_double((long long)gdpp64<CTRL>((uint64_t)__double_as_longlong(v)));
}
__device__ __forceinline__ uint64_t g_row_max_u64(uint64_t v) {
    uint64_t o;
    o = gdpp64<0xB1>(v); v = v > o ? v : o;
    o = gdpp64<0x4E>(v); v = v > o ? v : o;
    o = gdpp64<0x141>(v); v = v > o ? v : o;
    o = gdpp64<0x140>(v); v = v > o ? v : o;
    return v;
}
__device__ __forceinline__ double g_row_sum(double v) {
    v += gdppf<0xB1>(v);
    v += gdppf<0x4E>(v);
    v += gdppf<0x141>(v);
    v += gdppf<0x140>(v);
    return v;
}
// sum over the four 16-lane rows of a wave (lanes j, j+16, j+32, j+48); epilogue only
__device__ __forceinline__ double g_col_sum(double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ void g_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint64_t g_order_key(double x) {
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// In-place inverse and log|det| of a symmetric G x G matrix whose leading g x g block is live (the
// rest is the identity): Gauss-Jordan without pivoting, fully unrolled (symmetric positive systems).
template <int G>
__device__ __forceinline__ double sym_inverse(double (&M)[G][G]) {
    double logdet = 0.0;
#pragma unroll
    for (int p = 0; p < G; ++p) {
        const double piv = M[p][p];
        logdet += log(fabs(piv));
        const double ip = 1.0 / piv;
#pragma unroll
        for (int c = 0; c < G; ++c) M[p][c] *= ip;
        M[p][p] = ip;
#pragma unroll
        for (int r = 0; r < G; ++r) {
            if (r == p) continue;
            const double f = M[r][p];
#pragma unroll
            for (int c = 0; c < G; ++c) {
                if (c == p) continue;
                M[r][c] -= f * M[p][c];
            }
            M[r][p] = -f * ip;
        }
    }
    return logdet;
}

struct GscOffsets {
    int off[PM_MAX_HPRIME];
};

// per-latent tables (H doubles each), prepared on the host per EM step
struct GscTables {
    const double *c0;    // nc_h - mu_h^2 G_hh / s2,  nc_h = -(log psi_hh + log lam_h)
    const double *c1;    // 2 mu_h / s2
    const double *gm;    // G_hh mu_h
    const double *il;    // 1 / (lam_h s2^2)
    const double *kl;    // 1 / (lam_h s2)
    const double *ilam;  // 1 / lam_h
    const double *mu;
    const double *lpi;   // log(pi_h) - log(1 - pi_h)
};

template <int VPL, int GMAX>
__global__ __launch_bounds__(256) void gsc_estep_kernel(const double *__restrict__ scores, int64_t lds,
                                                         const double *__restrict__ gram,
                                                         const double *__restrict__ psi,
                                                         const double *__restrict__ ynorm2, GscTables T,
                                                         const uint16_t *__restrict__ masks, int S, double beta,
                                                         double inv_s2, int64_t N, int H, int Hp, int do_select,
                                                         int32_t *__restrict__ cand, double *__restrict__ xpt_s,
                                                         double *__restrict__ xpt_sz, int64_t ldx,
                                                         double *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ 8 tables (H) | colsum_s (H) | colsum_sz (H) | per datapoint: ac (16) cidx(16 as double slots) Gc Pc as asz ass aszsz ]
    double *s_tab = reinterpret_cast<double *>(smem);
    double *s_c0 = s_tab, *s_c1 = s_tab + H, *s_gm = s_tab + 2 * H, *s_il = s_tab + 3 * H, *s_kl = s_tab + 4 * H;
    double *s_ilam = s_tab + 5 * H, *s_mu = s_tab + 6 * H, *s_lpi = s_tab + 7 * H;
    double *s_cs = s_tab + 8 * H, *s_csz = s_tab + 9 * H;
    const int HH = Hp * Hp;
    const int dp_stride = 16 + 4 * HH + 2 * 16;
    double *s_dp = s_tab + 10 * H;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, row = lane >> 4;
    for (int h = tid; h < H; h += 256) {
        s_c0[h] = T.c0[h]; s_c1[h] = T.c1[h]; s_gm[h] = T.gm[h]; s_il[h] = T.il[h]; s_kl[h] = T.kl[h];
        s_ilam[h] = T.ilam[h]; s_mu[h] = T.mu[h]; s_lpi[h] = T.lpi[h];
        s_cs[h] = 0.0; s_csz[h] = 0.0;
    }
    double *s_ac = s_dp + (wave * 4 + row) * dp_stride;   // a at the candidates
    double *s_Gc = s_ac + 16, *s_Pc = s_Gc + HH;
    double *s_ass = s_Pc + HH, *s_aszsz = s_ass + HH;
    double *s_as = s_aszsz + HH, *s_asz = s_as + 16;
    __syncthreads();

    const double tiny = 2.2250738585072014e-308, fmin_ = -1.7976931348623157e308;
    double *g_ss = stats, *g_szsz = stats + (int64_t)H * H;
    // per-lane sums over datapoints: columns of xpt_s / xpt_sz, singleton diagonal of xpt_szsz.
    // (diag of sum xpt_ss needs no accumulator: s_h^2 = s_h, so it equals the column sum of xpt_s)
    double dszsz[VPL], cs[VPL], csz[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) dszsz[i] = cs[i] = csz[i] = 0.0;

    const int64_t groups = (N + ROWS - 1) / ROWS;
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int64_t n = grp * ROWS + wave * 4 + row;
        const bool live = n < N;
        const int64_t nn = live ? n : N - 1;
        const double *arow = scores + nn * lds;
        const double yn = ynorm2[nn];
        double a[VPL], sc[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            a[i] = (h < H) ? arow[h] : 0.0;
            double v = fmin_;
            if (h < H) {  // singleton log-posterior without prior (gsc_et.py:795-805)
                const double bb = a[i] - s_gm[h];
                v = s_c0[h] - yn * inv_s2 + s_c1[h] * a[i] + bb * bb * s_il[h];
                if (v != v || v < fmin_) v = fmin_;
                if (isinf(v)) v = 0.0;
            }
            sc[i] = v;
        }

        // ---- candidates: top-H' scores, then sorted by latent index (gsc_et.py:726-728)
        int myc = 0;
        if (do_select) {
            uint64_t key[VPL];
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int h = j + 16 * i;
                uint64_t k = 0;
                if (h < H) {
                    k = (g_order_key(sc[i]) & ~0x3FFull) | (uint64_t)h;
                    if (k < 0x400ull) k |= 0x400ull;
                }
                key[i] = k;
            }
            uint64_t mine = 0;                                // this lane's own selected latents, bit i
            for (int r = 0; r < Hp; ++r) {
                uint64_t m = key[0];
#pragma unroll
                for (int i = 1; i < VPL; ++i) m = m > key[i] ? m : key[i];
                m = g_row_max_u64(m);
#pragma unroll
                for (int i = 0; i < VPL; ++i)
                    if (key[i] == m) { key[i] = 0; mine |= 1ull << i; }
            }
            // rank of each selected latent among the selected = number of selected latents with a smaller index:
            // prefix over lanes of popcounts is awkward in j + 16 i order, so count directly: latent h = j + 16 i
            // precedes h' = j' + 16 i' iff i < i' or (i == i' and j < j').
            int cnt_i[VPL];  // selected latents in "slot" i across the row
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                double c = (double)((mine >> i) & 1ull);
                cnt_i[i] = (int)(g_row_sum(c) + 0.5);
            }
            int before_slot = 0;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const bool sel = (mine >> i) & 1ull;
                // selected lanes j' < j in the same slot: ballot restricted to this row
                const unsigned long long bal = __ballot(sel);
                const unsigned rowbits = (unsigned)((bal >> (row * 16)) & 0xFFFFull);
                const int lower = __builtin_popcount(rowbits & ((1u << j) - 1u));
                if (sel) s_as[before_slot + lower] = (double)(j + 16 * i);   // s_as reused as index scratch
                before_slot += cnt_i[i];
            }
            g_sync();
            if (j < Hp) myc = (int)s_as[j];
            g_sync();
            if (live && j < Hp) cand[n * Hp + j] = myc;
        } else {
            if (j < Hp) myc = cand[nn * Hp + j];
        }

        // ---- candidate blocks -> LDS, accumulators cleared
        int cpos[PM_MAX_HPRIME];
#pragma unroll
        for (int k = 0; k < PM_MAX_HPRIME; ++k)
            cpos[k] = (k < Hp) ? __builtin_amdgcn_ds_bpermute(((lane & 48) + k) << 2, myc) : 0;
        if (j < Hp) {
            s_ac[j] = arow[myc];
            s_as[j] = 0.0;
            s_asz[j] = 0.0;
        }
        for (int p = j; p < HH; p += 16) {
            const int i = p / Hp, k = p - i * Hp;
            int ci = 0, ck = 0;
#pragma unroll
            for (int q = 0; q < PM_MAX_HPRIME; ++q) {
                ci = (q == i) ? cpos[q] : ci;
                ck = (q == k) ? cpos[q] : ck;
            }
            s_Gc[p] = gram[(int64_t)ci * H + ck];
            s_Pc[p] = psi[(int64_t)ci * H + ck];
            s_ass[p] = 0.0;
            s_aszsz[p] = 0.0;
        }
        g_sync();

        // ---- null state + singletons
        double Z = (j == 0) ? exp(-yn * inv_s2 * beta) : 0.0;
        double xs[VPL], xsz[VPL], qzz[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            double p = 0.0, kap = 0.0;
            if (h < H) {
                // recompute the un-clamped singleton log-posterior (the score clamps are selection-only)
                const double bb = a[i] - s_gm[h];
                const double lp = s_c0[h] - yn * inv_s2 + s_c1[h] * a[i] + bb * bb * s_il[h] + s_lpi[h];
                p = exp(lp * beta);
                if (p != p || p < tiny) p = tiny;
                kap = bb * s_kl[h] + s_mu[h];
                Z += p;
            }
            xs[i] = p;
            xsz[i] = p * kap;
            qzz[i] = p * (kap * kap + ((h < H) ? s_ilam[h] : 0.0));
        }

        // ---- multi-cause states
        for (int s = j; s < S; s += 16) {
            const unsigned mask = masks[s];
            int pos[GMAX];
            int g = 0;
#pragma unroll
            for (int q = 0; q < GMAX; ++q) pos[q] = 0;
            {
                unsigned m = mask;
#pragma unroll
                for (int q = 0; q < GMAX; ++q) {
                    if (m) {
                        pos[q] = __builtin_ctz(m);
                        m &= m - 1;
                        g = q + 1;
                    }
                }
            }
            double Gm[GMAX][GMAX], Lm[GMAX][GMAX], av[GMAX], muv[GMAX];
            double prior = 0.0;
#pragma unroll
            for (int r = 0; r < GMAX; ++r) {
                const bool lr = r < g;
                int cr = 0;
#pragma unroll
                for (int q = 0; q < PM_MAX_HPRIME; ++q) cr = (q == pos[r]) ? cpos[q] : cr;
                av[r] = lr ? s_ac[pos[r]] : 0.0;
                muv[r] = lr ? s_mu[cr] : 0.0;
                prior += lr ? s_lpi[cr] : 0.0;
#pragma unroll
                for (int c = 0; c < GMAX; ++c) {
                    const bool lc = lr && (c < g);
                    Gm[r][c] = lc ? s_Gc[pos[r] * Hp + pos[c]] : 0.0;
                    Lm[r][c] = lc ? s_Pc[pos[r] * Hp + pos[c]] : ((r == c) ? 1.0 : 0.0);
                }
            }
            double C_det = sym_inverse<GMAX>(Lm);              // Lm = Psi_a^-1, log|det Psi_a|
#pragma unroll
            for (int r = 0; r < GMAX; ++r)
#pragma unroll
                for (int c = 0; c < GMAX; ++c)
                    Lm[r][c] = (r < g && c < g) ? Lm[r][c] + Gm[r][c] * inv_s2 : ((r == c) ? 1.0 : 0.0);
            C_det += sym_inverse<GMAX>(Lm);                    // Lm = Lambda^-1, + log|det Lambda|
            double bvec[GMAX], r2 = yn, quad = 0.0;
#pragma unroll
            for (int r = 0; r < GMAX; ++r) {
                double gmu = 0.0;
#pragma unroll
                for (int c = 0; c < GMAX; ++c) gmu += Gm[r][c] * muv[c];
                bvec[r] = av[r] - gmu;
                r2 += muv[r] * (gmu - 2.0 * av[r]);
            }
            double kap[GMAX];
#pragma unroll
            for (int r = 0; r < GMAX; ++r) {
                double lb = 0.0;
#pragma unroll
                for (int c = 0; c < GMAX; ++c) lb += Lm[r][c] * bvec[c];
                quad += bvec[r] * lb;
                kap[r] = lb * inv_s2 + muv[r];
            }
            const double lp = -C_det - r2 * inv_s2 + quad * inv_s2 * inv_s2 + prior;
            double p = exp(lp * beta);
            if (p != p || p < tiny) p = tiny;
            Z += p;
#pragma unroll
            for (int r = 0; r < GMAX; ++r) {
                if (r < g) {
                    atomicAdd(&s_as[pos[r]], p);
                    atomicAdd(&s_asz[pos[r]], p * kap[r]);
#pragma unroll
                    for (int c = 0; c < GMAX; ++c) {
                        if (c < g) {
                            atomicAdd(&s_ass[pos[r] * Hp + pos[c]], p);
                            atomicAdd(&s_aszsz[pos[r] * Hp + pos[c]], p * (kap[r] * kap[c] + Lm[r][c]));
                        }
                    }
                }
            }
        }
        Z = g_row_sum(Z);
        const double nf = 1.0 / (Z + tiny);
        g_sync();

        // ---- expectations of this datapoint; sums over datapoints
        if (live) {
            for (int p = j; p < HH; p += 16) {
                const int i = p / Hp, k = p - i * Hp;
                if (k < i) continue;   // symmetric blocks: upper triangle, mirrored by the host
                int ci = 0, ck = 0;
#pragma unroll
                for (int q = 0; q < PM_MAX_HPRIME; ++q) {
                    ci = (q == i) ? cpos[q] : ci;
                    ck = (q == k) ? cpos[q] : ck;
                }
                // candidates are sorted by index, so ci <= ck for i <= k
                pm_atomic_add(g_ss + (int64_t)ci * H + ck, s_ass[p] * nf);
                pm_atomic_add(g_szsz + (int64_t)ci * H + ck, s_aszsz[p] * nf);
            }
        }
#pragma unroll
        for (int k = 0; k < PM_MAX_HPRIME; ++k) {
            if (k < Hp) {
                const int c = cpos[k];
                if ((c & 15) == j) {
                    const double as = s_as[k], asz = s_asz[k];
#pragma unroll
                    for (int i = 0; i < VPL; ++i)
                        if ((c >> 4) == i) {
                            xs[i] += as;
                            xsz[i] += asz;
                        }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            if (live && h < H) {
                // singles contribute to the diagonals of sum xpt_ss / xpt_szsz (multi-cause diagonal
                // terms went through the block atomics above)
                const double ps = xs[i] * nf;
                xpt_s[n * ldx + h] = ps;
                xpt_sz[n * ldx + h] = xsz[i] * nf;
                cs[i] += ps;
                csz[i] += xsz[i] * nf;
                dszsz[i] += qzz[i] * nf;
            }
        }
        g_sync();
    }

    // flush per-lane column sums / singleton diagonals: lanes of different rows / waves own the same latent, so
    // fold them in LDS first (the parameter tables are dead by now: s_c0 takes the xpt_szsz diagonal) and send
    // ONE global atomic per latent and block -- a per-lane flush puts 256 * VPL atomics per block on H addresses
    // and serialises the whole grid's tail on them.
    __syncthreads();
    for (int h = tid; h < H; h += 256) s_c0[h] = 0.0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int h = j + 16 * i;
        // rows of the wave first (lanes j, j+16, j+32, j+48), then one LDS atomic per wave and latent
        const double c_s = g_col_sum(cs[i]), c_sz = g_col_sum(csz[i]), c_d = g_col_sum(dszsz[i]);
        if (row == 0 && h < H) {
            atomicAdd(&s_cs[h], c_s);
            atomicAdd(&s_csz[h], c_sz);
            atomicAdd(&s_c0[h], c_d);
        }
    }
    __syncthreads();
    double *g_cs = stats + 2 * (int64_t)H * H, *g_csz = g_cs + H, *g_dszsz = g_csz + H;
    for (int h = tid; h < H; h += 256) {
        if (s_cs[h] != 0.0) pm_atomic_add(g_cs + h, s_cs[h]);
        if (s_csz[h] != 0.0) pm_atomic_add(g_csz + h, s_csz[h]);
        if (s_c0[h] != 0.0) pm_atomic_add(g_dszsz + h, s_c0[h]);
    }
}

static int allow_lds_gsc(const void *kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" int64_t pm_gsc_stats_len(int64_t H) { return 2 * H * H + 3 * H; }

extern "C" int pm_gsc_supported(int64_t H, int64_t Hprime, int64_t gamma) {
    if (H <= 0 || H > 512 || Hprime <= 0 || Hprime > PM_MAX_HPRIME || Hprime > H || gamma < 1 || gamma > 4) return 0;
    const size_t shmem = sizeof(double) * (10 * H + ROWS * (48 + 4 * Hprime * Hprime));
    return shmem <= 64 * 1024 ? 1 : 0;
}

extern "C" int pm_gsc_estep_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                                const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                                int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                                int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx,
                                double *stats, void *stream) {
    if (N == 0) return PM_OK;
    if (!scores || !gram || !psi_sq || !ynorm2 || !tables || !cand || !xpt_s || !xpt_sz || !stats || N < 0 || H <= 0 ||
        Hprime <= 0 || S < 0 || lds < H || ldx < H || (S > 0 && !state_masks) || !(sigma_sq > 0.0))
        return PM_EINVAL;
    if (!pm_gsc_supported(H, Hprime, gamma)) return PM_ERANGE;
    GscTables T{tables, tables + H, tables + 2 * H, tables + 3 * H, tables + 4 * H, tables + 5 * H, tables + 6 * H,
                tables + 7 * H};
    const size_t shmem = sizeof(double) * (10 * H + ROWS * (48 + 4 * Hprime * Hprime));
    int64_t groups = (N + ROWS - 1) / ROWS;
    if (groups > 2048) groups = 2048;
    dim3 grid((unsigned)groups), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double inv_s2 = 1.0 / sigma_sq;
#define PM_LAUNCH(V, G)                                                                                             \
    do {                                                                                                            \
        if (int e = allow_lds_gsc(reinterpret_cast<const void *>(gsc_estep_kernel<V, G>), shmem)) return e;          \
        hipLaunchKernelGGL((gsc_estep_kernel<V, G>), grid, block, shmem, s, scores, lds, gram, psi_sq, ynorm2, T,   \
                           state_masks, (int)S, beta, inv_s2, N, (int)H, (int)Hprime, do_select, cand, xpt_s, xpt_sz, \
                           ldx, stats);                                                                             \
    } while (0)
#define PM_BY_G(V)                            \
    do {                                      \
        if (gamma <= 2) PM_LAUNCH(V, 2);      \
        else if (gamma == 3) PM_LAUNCH(V, 3); \
        else PM_LAUNCH(V, 4);                 \
    } while (0)
    if (H <= 16) PM_BY_G(1);
    else if (H <= 32) PM_BY_G(2);
    else if (H <= 64) PM_BY_G(4);
    else if (H <= 128) PM_BY_G(8);
    else if (H <= 256) PM_BY_G(16);
    else PM_BY_G(32);
#undef PM_BY_G
#undef PM_LAUNCH
    return (int)hipGetLastError();
}
