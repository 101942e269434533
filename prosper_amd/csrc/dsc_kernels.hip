// Discrete Sparse Coding (prosper/em/camodels/dsc_et.py) on gfx950: K-ary latents with values
// v_k (one of them 0), linear superposition.  All energies come from the f64 MFMA scores GEMM
// a = Y.W^T and the Gram matrix G = W.W^T:
//
//   singleton (k,h):   e = v_k^2 G_hh - 2 v_k a_h + |y|^2
//   multi-cause state: e = |y|^2 + sum_j v_j (v_j G_jj - 2 a_j) + 2 sum_{j<k} v_j v_k G_jk   (candidate block)
//
//   dsc_select_scores_kernel   R[n,h] = -max_k (pre1 (v_k^2 G_hh - 2 v_k a_h) + log pi_k): the per-latent
//                              best singleton log-joint up to datapoint constants (dsc_et.py:389-400);
//                              ranked by the 16-lane selection kernel (smallest R first = best first)
//   dsc_estep_kernel           log-pseudo-joints + stabilised log-evidence (dsc_et.py:492-585)
//   dsc_mstep_rows_kernel      q = exp(logpj - lse); E[s] rows (for Wp = E[s]^T.Y), E[s s^T] scatter,
//                              expected value counts (pi), sum q e (sigma), sum lse (L) (dsc_et.py:660-735)
//
// One 64-lane wavefront per datapoint, four per workgroup; the state table (S x H' value indices,
// one byte each) and |W_h|^2 sit in LDS.  These kernels are bandwidth/latency-bound row passes; the
// flops of the model are in the two GEMMs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

constexpr int WAVES = 4;
#ifndef PM_DSC_M16_WPE
#define PM_DSC_M16_WPE 4     // wavefronts per SIMD dsc_mstep_rows16_kernel<8, 8> is compiled for: 4 costs 28 spilled registers and
                             // still wins (0.170 ms; 3 per SIMD, no spill: 0.200; scratch/dsc_ab.sh)
#endif

__device__ __forceinline__ void wave_sync_lds_dsc() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void dsc_select_scores_kernel(const double *__restrict__ scores, int64_t lds,
                                                                 const double *__restrict__ gram, pm_dsc_params P,
                                                                 int64_t N, int H, double *__restrict__ R,
                                                                 int64_t ldr) {
    const int64_t total = N * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / H;
        const int h = (int)(i - n * H);
        const double a = scores[n * lds + h];
        const double w2 = gram[(int64_t)h * H + h];
        double best = -INFINITY;
        for (int k = 0; k < P.K; ++k) {
            if (k == P.K0) continue;
            const double v = P.values[k];
            best = fmax(best, P.pre1 * (v * v * w2 - 2.0 * v * a) + P.logpi[k]);
        }
        R[n * ldr + h] = -best;
    }
}

__global__ __launch_bounds__(256) void tsc_select_scores_kernel(const double *__restrict__ scores, int64_t lds,
                                                                 const double *__restrict__ gram, int64_t N, int H,
                                                                 double *__restrict__ R, int64_t ldr) {
    const int64_t total = N * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / H;
        const int h = (int)(i - n * H);
        const double a2 = 2.0 * scores[n * lds + h];
        const double w2 = gram[(int64_t)h * H + h];
        R[n * ldr + h] = -(w2 + a2);
        R[n * ldr + H + h] = -(w2 - a2);
    }
}

// energy of multi-cause state `row` (H' value indices) from the candidate block in LDS.  The state's values are
// gathered first (s_val: the K values in LDS -- indexing the kernel-argument struct per lane compiles to a
// dependent global load per element), then every position is visited with static register indices; zero values
// add exact zeros, so the result equals the reference's sum over the non-zero positions bit for bit.
template <int MAXHP>
__device__ __forceinline__ double state_energy(const uint8_t *row, int Hp, const double *s_val, const double *s_a,
                                               const double *s_G, double yn) {
    double v[MAXHP];
#pragma unroll
    for (int j = 0; j < MAXHP; ++j) v[j] = (j < Hp) ? s_val[row[j]] : 0.0;
    double e = yn;
#pragma unroll
    for (int j = 0; j < MAXHP; ++j) {
        if (j < Hp) {
            double t = v[j] * s_G[j * Hp + j] - 2.0 * s_a[j];
#pragma unroll
            for (int k = 0; k < j; ++k) t += 2.0 * v[k] * s_G[j * Hp + k];
            e += v[j] * t;
        }
    }
    return e;
}

// Builds the energy-term tables of a workgroup (see dsc_estep_kernel): decode tables s_dj / s_dk / s_c1 / s_c2 of the NT
// entries, the six entry indices of every state in s_off.  Returns (to every thread) whether some state has more than
// three non-zero positions -- the tables do not apply then.  Every thread of the workgroup calls this.
__device__ __forceinline__ bool dsc_build_energy_tables(const pm_dsc_params &P, int Hp, int S, int NT, const uint8_t *s_tab,
                                                        const double *s_val, uint8_t *s_off, uint8_t *s_dj, uint8_t *s_dk,
                                                        double *s_c1, double *s_c2) {
    const int tid = threadIdx.x;
    const int Kn = P.K - 1, nU = Hp * Kn;
    for (int e = tid; e < NT; e += blockDim.x) {
        int j = 0, k = 0;
        double c1 = 0.0, c2 = 0.0;
        if (e >= 1 && e <= nU) {
            j = k = (e - 1) / Kn;
            const int c = (e - 1) % Kn;
            const double v = s_val[c < P.K0 ? c : c + 1];
            c1 = v * v;
            c2 = -2.0 * v;
        } else if (e > nU) {
            const int r = e - 1 - nU, pair = r / (Kn * Kn), cc = r % (Kn * Kn);
            k = 1;
            while ((k + 1) * k / 2 <= pair) ++k;           // pair = k (k - 1) / 2 + j, j < k
            j = pair - k * (k - 1) / 2;
            const int ca = cc / Kn, cb = cc % Kn;
            c1 = 2.0 * s_val[ca < P.K0 ? ca : ca + 1] * s_val[cb < P.K0 ? cb : cb + 1];
        }
        s_dj[e] = (uint8_t)j;
        s_dk[e] = (uint8_t)k;
        s_c1[e] = c1;
        s_c2[e] = c2;
    }
    int too_many = 0;
    for (int st = tid; st < S; st += blockDim.x) {
        int pos[3] = {0, 0, 0}, cv[3] = {0, 0, 0}, g = 0;
        for (int j = 0; j < Hp; ++j) {
            const int ki = s_tab[st * Hp + j];
            if (ki != P.K0) {
                if (g < 3) {
                    pos[g] = j;
                    cv[g] = ki < P.K0 ? ki : ki - 1;
                }
                ++g;
            }
        }
        too_many |= g > 3;
        auto U = [&](int a) { return a < g ? 1 + pos[a] * Kn + cv[a] : 0; };
        auto PP = [&](int a, int b) {       // a < b: positions ascend
            return b < g ? 1 + nU + (pos[b] * (pos[b] - 1) / 2 + pos[a]) * Kn * Kn + cv[a] * Kn + cv[b] : 0;
        };
        uint8_t *o = s_off + (size_t)st * 8;
        o[0] = (uint8_t)U(0);
        o[1] = (uint8_t)U(1);
        o[2] = (uint8_t)U(2);
        o[3] = (uint8_t)PP(0, 1);
        o[4] = (uint8_t)PP(0, 2);
        o[5] = (uint8_t)PP(1, 2);
        o[6] = o[7] = 0;
    }
    return __syncthreads_or(too_many);
}

template <int MAXHP>
__global__ __launch_bounds__(64 * WAVES, 4) void dsc_estep_kernel(
    const double *__restrict__ scores, int64_t lds, const double *__restrict__ gram,
    const double *__restrict__ ynorm2, const int32_t *__restrict__ cand, const uint8_t *__restrict__ state_idx, int S,
    const double *__restrict__ prior_g, pm_dsc_params P, int64_t N, int H, int Hp, double *__restrict__ logpj,
    int64_t ldl, double *__restrict__ lse, int stage, int fast_off, int NT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ w2 (H) | per wave: a (Hp) G (Hp*Hp) | state table (S*Hp bytes, padded to 8) | staged: prior (Kt) | per wave: f (Kt) ]
    // staged (when it fits): the log-prior table is read from LDS instead of global memory for every datapoint, and
    // the row of log-joints is kept in LDS for the log-sum-exp pass instead of being read back from global memory
    // behind its own stores (two dependent memory round trips per datapoint less)
    double *s_w2 = reinterpret_cast<double *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    double *s_a = s_w2 + H + wave * (Hp + Hp * Hp);
    double *s_G = s_a + Hp;
    uint8_t *s_tab = reinterpret_cast<uint8_t *>(s_w2 + H + WAVES * (Hp + Hp * Hp));
    __shared__ double s_val[PM_DSC_MAX_K];
    // 2^(j/128) table of pm_exp_tab (pm_common.h): the log-sum-exp's exponentials were libm exp() calls -- ~45 VALU
    // instructions each, 340 of the kernel's ~900 per datapoint (round 3, SQ_INSTS_VALU); table-driven they are 14, and
    // terms below e^-37 of the largest are not evaluated at all
    __shared__ double s_E[128];
    if (tid < 128) s_E[tid] = pm_powtab_dev[256 + tid];
    const double *etab = s_E - 256;                      // pm_exp_tab indexes tab[256 + (k & 127)]
    for (int h = tid; h < H; h += blockDim.x) s_w2[h] = gram[(int64_t)h * H + h];
    for (int i = tid; i < S * Hp; i += blockDim.x) s_tab[i] = state_idx[i];
    if (tid < PM_DSC_MAX_K) s_val[tid] = (tid < P.K) ? P.values[tid] : 0.0;
    __syncthreads();

    const int nss = (P.K - 1) * H;
    const bool tab = P.flags & PM_DSC_TABLE_ONLY;        // TSC: columns = rows of the state table, nothing else
    const int base = tab ? 0 : 1 + nss;
    const int Kt = base + S;
    double *s_prior = reinterpret_cast<double *>(smem + (((size_t)(H + WAVES * (Hp + Hp * Hp)) * 8 + (size_t)S * Hp + 7) & ~size_t(7)));
    double *s_f = s_prior + Kt + (size_t)wave * Kt;
    if (stage) {
        for (int i = tid; i < Kt; i += blockDim.x) s_prior[i] = prior_g[i];
        __syncthreads();
    }
    const double *prior = stage ? s_prior : prior_g;

    // Energy terms by table (round 3).  The energy of a state is yn + sum_j v_j (v_j G_jj - 2 a_j) + 2 sum_{k<j} v_j v_k
    // G_jk over its non-zero positions: with at most three of them (gamma <= 3) it is the sum of three single-position
    // terms U(j, v) and three pair terms P(j, k, v, v') out of a per-datapoint table of NT = 1 + H'(K-1) + C(H',2)(K-1)^2
    // entries (73 at H' = 6, K = 3; entry 0 is an exact zero for the unused slots).  The table costs ~10 instructions
    // per entry and datapoint, a state 6 LDS reads + 6 adds instead of the ~80 instructions of state_energy (which
    // walks all H' positions with zero factors) -- the states were 60 % of this kernel's instructions.
    //   fast tables: [ s_off (S x 8 bytes: six entry indices) | s_dj (NT) s_dk (NT) | s_c1 (NT) | s_c2 (NT) | per wave: T (NT) ]
    //   T[e] = c1[e] G[dj, dk] + c2[e] a[dj]
    // (Tried on top, not kept: fetching the candidate gather -- two dependent round trips -- one datapoint ahead:
    // 0.28 vs 0.21 ms -- not investigated further; loads and stores share the vmcnt counter on gfx950, so the wait for
    // the prefetched values may well have become a wait for the previous datapoint's 477 log-joint stores.)
    bool fast = NT > 0;
    uint8_t *s_off = smem + fast_off;
    const int NTp = (NT + 7) & ~7;
    uint8_t *s_dj = s_off + (size_t)S * 8, *s_dk = s_dj + NTp;
    double *s_c1 = reinterpret_cast<double *>(s_dk + NTp), *s_c2 = s_c1 + NT;
    double *s_T = s_c2 + NT + (size_t)wave * NT;
    if (fast)
        fast = !dsc_build_energy_tables(P, Hp, S, NT, s_tab, s_val, s_off, s_dj, s_dk, s_c1, s_c2);   // (> 3 non-zeros: generic walk)
    for (int64_t n = (int64_t)blockIdx.x * WAVES + wave; n < N; n += (int64_t)gridDim.x * WAVES) {
        const double *arow = scores + n * lds;
        const int32_t *cn = cand + n * Hp;
        const double yn = ynorm2[n];
        const int myc = lane < Hp ? cn[lane] : 0;        // candidate `lane` of this datapoint: ONE load, shuffles after
        if (lane < Hp) s_a[lane] = arow[myc];
        for (int p0 = 0; p0 < Hp * Hp; p0 += 64) {         // uniform trip count: every lane feeds the shuffles
            const int p = p0 + lane;
            const bool ok = p < Hp * Hp;
            const int ci = __shfl(myc, ok ? p / Hp : 0), ck = __shfl(myc, ok ? p % Hp : 0);
            if (ok) s_G[p] = gram[(int64_t)ci * H + ck];
        }
        wave_sync_lds_dsc();
        if (fast) {
            for (int e = lane; e < NT; e += 64) {
                const int j = s_dj[e];
                s_T[e] = fma(s_c1[e], s_G[j * Hp + s_dk[e]], s_c2[e] * s_a[j]);
            }
        }

        double *out = logpj + n * ldl;
        double m = -INFINITY;
        if (!tab && lane == 0) {
            const double f0 = P.ecoef * yn + P.pscale * prior[0];
            out[0] = f0;
            if (stage) s_f[0] = f0;
            m = f0;
        }
        // singletons: column 1 + c*H + h for the c-th non-zero value (dsc_et.py:566-568)
        int c = 0;
        for (int k = 0; k < P.K && !tab; ++k) {
            if (k == P.K0) continue;
            const double v = P.values[k];
            for (int h = lane; h < H; h += 64) {
                const double e = v * v * s_w2[h] - 2.0 * v * arow[h] + yn;
                const double f = P.ecoef * e + P.pscale * prior[1 + c * H + h];
                out[1 + c * H + h] = f;
                if (stage) s_f[1 + c * H + h] = f;
                m = fmax(m, f);
            }
            ++c;
        }
        if (fast) wave_sync_lds_dsc();
        for (int s = lane; s < S; s += 64) {
            double e;
            if (fast) {
                const uint2 o = *reinterpret_cast<const uint2 *>(s_off + (size_t)s * 8);
                e = yn + s_T[o.x & 255u] + s_T[(o.x >> 8) & 255u] + s_T[(o.x >> 16) & 255u] + s_T[o.x >> 24] +
                    s_T[o.y & 255u] + s_T[(o.y >> 8) & 255u];
            } else {
                e = state_energy<MAXHP>(s_tab + s * Hp, Hp, s_val, s_a, s_G, yn);
            }
            const double f = P.ecoef * e + P.pscale * prior[base + s];
            out[base + s] = f;
            if (stage) s_f[base + s] = f;
            m = fmax(m, f);
        }
        m = pm_wave_max(m);
        // second pass over this lane's own values
        const double *src = stage ? s_f : out;
        double sum = 0.0;
        auto add = [&](double d, bool mine) {           // uniform trip counts: every lane reaches the __any
            const bool need = mine && d > -37.0;
            if (__any(need)) sum += need ? pm_exp_tab(d, etab) : 0.0;
        };
        if (!tab) add(lane == 0 ? src[0] - m : 0.0, lane == 0);
        c = 0;
        for (int k = 0; k < P.K && !tab; ++k) {
            if (k == P.K0) continue;
            for (int h0 = 0; h0 < H; h0 += 64) {
                const int h = h0 + lane;
                add(h < H ? src[1 + c * H + h] - m : 0.0, h < H);
            }
            ++c;
        }
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            add(s < S ? src[base + s] - m : 0.0, s < S);
        }
        sum = pm_wave_sum(sum);
        if (lane == 0) lse[n] = m + log(sum);
        wave_sync_lds_dsc();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dsc_estep_kernel with SIXTEEN lanes per datapoint (four datapoints per wavefront, 16 per workgroup).  The one-
// wavefront form is latency-bound -- per datapoint two dependent memory round trips (candidates, then their scores and
// Gram entries) in front of ~500 instructions, and the wave-wide reductions, the gather and the syncs are paid per
// datapoint.  Here four datapoints share every wavefront instruction and four gathers are in flight per wavefront;
// reductions are four DPP steps inside a 16-lane row.  Lane j of a row takes the singleton columns h = j + 16 i and
// the states s = j + 16 i.  The log-joints are not parked in LDS between the two passes of the log-sum-exp (16 rows of
// K doubles would not fit): the second pass recomputes them (6-12 instructions each) and evaluates the exponential.
// LDS: [ w2 (H) | state table | prior (K, when staged) | energy-term tables | per row: a (H') G (H'^2) T (NT) ]
// ---------------------------------------------------------------------------------------------------------------
struct Lay16 {
    int tab, prior, off, dj, dk, c1, c2, rows, row_stride, bytes;
};
__host__ __device__ inline Lay16 dsc_lay16(int H, int Hp, int S, int Kt, int NT, int stage) {
    Lay16 L;
    int o = 8 * H;
    L.tab = o;
    o += (S * Hp + 7) & ~7;
    L.prior = o;
    o += stage ? 8 * Kt : 0;
    L.off = o;
    o += NT ? S * 8 : 0;
    const int NTp = (NT + 7) & ~7;
    L.dj = o;
    L.dk = o + NTp;
    o += 2 * NTp;
    L.c1 = o;
    L.c2 = o + 8 * NT;
    o += 16 * NT;
    L.rows = o;
    L.row_stride = 8 * (Hp + Hp * Hp + NT);
    L.bytes = o + 16 * L.row_stride;
    return L;
}

template <int MAXHP, int VPL>      // VPL: latents per lane, H <= 16 VPL (the scores row of a datapoint is held in registers)
__global__ __launch_bounds__(256, MAXHP <= 8 ? 4 : 2) void dsc_estep16_kernel(
    const double *__restrict__ scores, int64_t lds, const double *__restrict__ gram,
    const double *__restrict__ ynorm2, const int32_t *__restrict__ cand, const uint8_t *__restrict__ state_idx, int S,
    const double *__restrict__ prior_g, pm_dsc_params P, int64_t N, int H, int Hp, double *__restrict__ logpj,
    int64_t ldl, double *__restrict__ lse, int stage, int NT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nss = (P.K - 1) * H;
    const bool tab = P.flags & PM_DSC_TABLE_ONLY;        // TSC: columns = rows of the state table, nothing else
    const int base = tab ? 0 : 1 + nss;
    const int Kt = base + S;
    const Lay16 L = dsc_lay16(H, Hp, S, Kt, NT, stage);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, row = lane >> 4, rowbase = lane & 48;
    double *s_w2 = reinterpret_cast<double *>(smem);
    uint8_t *s_tab = smem + L.tab;
    double *s_prior = reinterpret_cast<double *>(smem + L.prior);
    uint8_t *s_off = smem + L.off, *s_dj = smem + L.dj, *s_dk = smem + L.dk;
    double *s_c1 = reinterpret_cast<double *>(smem + L.c1), *s_c2 = reinterpret_cast<double *>(smem + L.c2);
    double *s_a = reinterpret_cast<double *>(smem + L.rows + (wave * 4 + row) * L.row_stride);
    double *s_G = s_a + Hp, *s_T = s_G + Hp * Hp;
    __shared__ double s_val[PM_DSC_MAX_K];
    __shared__ double s_E[128];
    if (tid < 128) s_E[tid] = pm_powtab_dev[256 + tid];
    const double *etab = s_E - 256;
    for (int h = tid; h < H; h += blockDim.x) s_w2[h] = gram[(int64_t)h * H + h];
    for (int i = tid; i < S * Hp; i += blockDim.x) s_tab[i] = state_idx[i];
    if (tid < PM_DSC_MAX_K) s_val[tid] = (tid < P.K) ? P.values[tid] : 0.0;
    if (stage)
        for (int i = tid; i < Kt; i += blockDim.x) s_prior[i] = prior_g[i];
    __syncthreads();
    const double *prior = stage ? s_prior : prior_g;
    bool fast = NT > 0;
    if (fast) fast = !dsc_build_energy_tables(P, Hp, S, NT, s_tab, s_val, s_off, s_dj, s_dk, s_c1, s_c2);

    for (int64_t n0 = (int64_t)blockIdx.x * 16; n0 < N; n0 += (int64_t)gridDim.x * 16) {
        const int64_t n = n0 + wave * 4 + row;
        const bool live = n < N;
        const int64_t nn = live ? n : N - 1;              // rows past N shadow the last datapoint and store nothing
        const double *arow = scores + nn * lds;
        const double yn = ynorm2[nn];
        const int myc = j < Hp ? cand[nn * Hp + j] : 0;
        // the scores row, requested at once (the singleton loops below would otherwise wait for one load per trip)
        double ar[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) ar[i] = (!tab && j + 16 * i < H) ? arow[j + 16 * i] : 0.0;
        if (j < Hp) s_a[j] = arow[myc];
        for (int p0 = 0; p0 < Hp * Hp; p0 += 16) {        // uniform trip count: every lane feeds the permutes
            const int p = p0 + j;
            const bool ok = p < Hp * Hp;
            const int pi = ok ? p / Hp : 0, pk = ok ? p - pi * Hp : 0;
            const int ci = __builtin_amdgcn_ds_bpermute((rowbase + pi) << 2, myc);
            const int ck = __builtin_amdgcn_ds_bpermute((rowbase + pk) << 2, myc);
            if (ok) s_G[p] = gram[(int64_t)ci * H + ck];
        }
        wave_sync_lds_dsc();
        if (fast) {
            for (int e = j; e < NT; e += 16) {
                const int dj = s_dj[e];
                s_T[e] = fma(s_c1[e], s_G[dj * Hp + s_dk[e]], s_c2[e] * s_a[dj]);
            }
            wave_sync_lds_dsc();
        }
        double *out = logpj + nn * ldl;
        // every column of this lane, in a fixed order: fn(column, log-joint)
        auto visit = [&](auto &&fn) {
            if (!tab && j == 0) fn(0, P.ecoef * yn + P.pscale * prior[0]);
            int c = 0;
            for (int k = 0; k < P.K && !tab; ++k) {
                if (k == P.K0) continue;
                const double v = s_val[k];
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const int h = j + 16 * i;
                    if (h < H) {
                        const double e = v * v * s_w2[h] - 2.0 * v * ar[i] + yn;
                        fn(1 + c * H + h, P.ecoef * e + P.pscale * prior[1 + c * H + h]);
                    }
                }
                ++c;
            }
            for (int st = j; st < S; st += 16) {
                double e;
                if (fast) {
                    const uint2 o = *reinterpret_cast<const uint2 *>(s_off + (size_t)st * 8);
                    e = yn + s_T[o.x & 255u] + s_T[(o.x >> 8) & 255u] + s_T[(o.x >> 16) & 255u] + s_T[o.x >> 24] +
                        s_T[o.y & 255u] + s_T[(o.y >> 8) & 255u];
                } else {
                    e = state_energy<MAXHP>(s_tab + st * Hp, Hp, s_val, s_a, s_G, yn);
                }
                fn(base + st, P.ecoef * e + P.pscale * prior[base + st]);
            }
        };
        double m = -INFINITY;
        visit([&](int col, double f) {
            if (live) out[col] = f;
            m = fmax(m, f);
        });
        m = fmax(m, pm_dpp_f64<0xB1>(m));
        m = fmax(m, pm_dpp_f64<0x4E>(m));
        m = fmax(m, pm_dpp_f64<0x141>(m));
        m = fmax(m, pm_dpp_f64<0x140>(m));
        double sum = 0.0;
        visit([&](int, double f) {
            const double d = f - m;
            sum += d > -37.0 ? pm_exp_tab(d, etab) : 0.0;
        });
        sum += pm_dpp_f64<0xB1>(sum);
        sum += pm_dpp_f64<0x4E>(sum);
        sum += pm_dpp_f64<0x141>(sum);
        sum += pm_dpp_f64<0x140>(sum);
        if (j == 0 && live) lse[n] = m + log(sum);
        wave_sync_lds_dsc();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dsc_estep16_kernel that ALSO produces the M-step's per-datapoint statistics (dsc_et.py:587-774; what
// dsc_mstep_rows16_kernel computes from the stored log-joints): the posterior weights are the exponentials the
// log-sum-exp evaluates anyway -- taken relative to the row maximum, scaled by 1 / sum once per datapoint -- so the
// second pass over 382 MB of log-joints (config-4-sized DSC) and its exponentials go away.  For E_step inside
// CAModel.step with no data truncation ahead (every datapoint is kept); terms below e^-37 of the largest add nothing.
// LDS: the layout of dsc_estep16_kernel, then [ qdiag (H) cnt (8) scal (4) | per row: m (H') B (H'^2) ]; the E[s] row of a
// datapoint stays in its lanes' registers.
// ---------------------------------------------------------------------------------------------------------------
#ifndef PM_DSC_MS_WPE
#define PM_DSC_MS_WPE 3        // wavefronts per SIMD the kernel is compiled for (register budget 512 / WPE)
#endif
template <int MAXHP, int VPL, int KM>      // KM >= K: latent values the per-lane counters are sized for (4 or 8)
__global__ __launch_bounds__(256, (MAXHP <= 8 && VPL <= 8) ? PM_DSC_MS_WPE : 2) void dsc_estep16_ms_kernel(
    const double *__restrict__ scores, int64_t lds, const double *__restrict__ gram,
    const double *__restrict__ ynorm2, const int32_t *__restrict__ cand, const uint8_t *__restrict__ state_idx, int S,
    const double *__restrict__ prior_g, pm_dsc_params P, int64_t N, int H, int D, int Hp, double *__restrict__ logpj,
    int64_t ldl, double *__restrict__ lse, int stage, int NT, double *__restrict__ expect, int64_t lde,
    double *__restrict__ stats, uint16_t *__restrict__ nz_idx, double *__restrict__ nz_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nss = (P.K - 1) * H;
    const bool tab = P.flags & PM_DSC_TABLE_ONLY;        // TSC: columns = rows of the state table, nothing else
    const int base = tab ? 0 : 1 + nss;
    const int Kt = base + S;
    const Lay16 L = dsc_lay16(H, Hp, S, Kt, NT, stage);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, row = lane >> 4, rowbase = lane & 48;
    double *s_w2 = reinterpret_cast<double *>(smem);
    uint8_t *s_tab = smem + L.tab;
    double *s_prior = reinterpret_cast<double *>(smem + L.prior);
    uint8_t *s_off = smem + L.off, *s_dj = smem + L.dj, *s_dk = smem + L.dk;
    double *s_c1 = reinterpret_cast<double *>(smem + L.c1), *s_c2 = reinterpret_cast<double *>(smem + L.c2);
    double *s_a = reinterpret_cast<double *>(smem + L.rows + (wave * 4 + row) * L.row_stride);
    double *s_G = s_a + Hp, *s_T = s_G + Hp * Hp;
    // the statistics' areas behind the E-step layout
    double *s_qdiag = reinterpret_cast<double *>(smem + ((L.bytes + 7) & ~7));
    double *s_cnt = s_qdiag + H, *s_scal = s_cnt + PM_DSC_MAX_K;
    const int per_row = Hp + Hp * Hp;
    double *s_m = s_scal + 4 + (size_t)(wave * 4 + row) * per_row;
    double *s_B = s_m + Hp;
    __shared__ double s_val[PM_DSC_MAX_K];
    __shared__ double s_E[128];
    if (tid < 128) s_E[tid] = pm_powtab_dev[256 + tid];
    const double *etab = s_E - 256;
    for (int h = tid; h < H; h += blockDim.x) s_w2[h] = gram[(int64_t)h * H + h];
    for (int i = tid; i < S * Hp; i += blockDim.x) s_tab[i] = state_idx[i];
    if (tid < PM_DSC_MAX_K) s_val[tid] = (tid < P.K) ? P.values[tid] : 0.0;
    for (int h = tid; h < H + PM_DSC_MAX_K + 4; h += blockDim.x) s_qdiag[h] = 0.0;
    if (stage)
        for (int i = tid; i < Kt; i += blockDim.x) s_prior[i] = prior_g[i];
    __syncthreads();
    const double *prior = stage ? s_prior : prior_g;
    bool fast = NT > 0;
    if (fast) fast = !dsc_build_energy_tables(P, Hp, S, NT, s_tab, s_val, s_off, s_dj, s_dk, s_c1, s_c2);
    double sig = 0.0, fs = 0.0, kept = 0.0, overflow = 0.0;
    double cnt[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) cnt[k] = 0.0;
    double *Wq = stats + (int64_t)H * D;

    // The scores row, |y|^2 and the candidates of the NEXT group of datapoints are requested before this group's outputs are
    // stored (`ar_n`, below): as far as s_waitcnt can tell vector-memory operations complete in issue order, so a row requested
    // behind ~40 stores per lane waits for their acknowledgements too (gsc_kernels.hip: the same change was worth 7 % there).
    double ar_n[VPL], yn_n = 0.0;
    int myc_n = -1;
    auto prefetch = [&](int64_t g0) {
        const int64_t n = g0 + wave * 4 + row;
        const int64_t nn = n < N ? n : N - 1;
        const double *ap = scores + nn * lds;
        yn_n = ynorm2[nn];
        myc_n = j < Hp ? cand[nn * Hp + j] : -1;
#pragma unroll
        for (int i = 0; i < VPL; ++i) ar_n[i] = (!tab && j + 16 * i < H) ? ap[j + 16 * i] : 0.0;
    };
    prefetch((int64_t)blockIdx.x * 16);
    for (int64_t n0 = (int64_t)blockIdx.x * 16; n0 < N; n0 += (int64_t)gridDim.x * 16) {
        const int64_t n = n0 + wave * 4 + row;
        const bool live = n < N;
        const int64_t nn = live ? n : N - 1;              // rows past N shadow the last datapoint and contribute nothing
        const double *arow = scores + nn * lds;
        const double yn = yn_n;
        const int myc = myc_n;
        double ar[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) ar[i] = ar_n[i];
        if (j < Hp) {
            s_a[j] = arow[myc];
            s_m[j] = 0.0;
        }
        for (int p = j; p < Hp * Hp; p += 16) s_B[p] = 0.0;
        for (int p0 = 0; p0 < Hp * Hp; p0 += 16) {        // uniform trip count: every lane feeds the permutes
            const int p = p0 + j;
            const bool ok = p < Hp * Hp;
            const int pi = ok ? p / Hp : 0, pk = ok ? p - pi * Hp : 0;
            const int ci = __builtin_amdgcn_ds_bpermute((rowbase + pi) << 2, myc);
            const int ck = __builtin_amdgcn_ds_bpermute((rowbase + pk) << 2, myc);
            if (ok) s_G[p] = gram[(int64_t)ci * H + ck];
        }
        // positions that are the last occurrence of their latent (all of them unless candidates repeat: TSC)
        bool mine_last = j < Hp;
        for (int k = 1; k < Hp; ++k) {                          // uniform trip count
            const int other = __builtin_amdgcn_ds_bpermute((rowbase + k) << 2, myc);
            if (k > j && other == myc) mine_last = false;
        }
        const unsigned lastmask = (P.flags & PM_DSC_LAST_POSITION)
                                      ? (unsigned)((__ballot(mine_last) >> rowbase) & 0xFFFFull)
                                      : ((1u << Hp) - 1u);
        wave_sync_lds_dsc();
        if (fast) {
            for (int e = j; e < NT; e += 16) {
                const int dj = s_dj[e];
                s_T[e] = fma(s_c1[e], s_G[dj * Hp + s_dk[e]], s_c2[e] * s_a[dj]);
            }
            wave_sync_lds_dsc();
        }
        double *out = logpj + nn * ldl;
        auto state_e = [&](int st) -> double {
            if (fast) {
                const uint2 o = *reinterpret_cast<const uint2 *>(s_off + (size_t)st * 8);
                return yn + s_T[o.x & 255u] + s_T[(o.x >> 8) & 255u] + s_T[(o.x >> 16) & 255u] + s_T[o.x >> 24] +
                       s_T[o.y & 255u] + s_T[(o.y >> 8) & 255u];
            }
            return state_energy<MAXHP>(s_tab + st * Hp, Hp, s_val, s_a, s_G, yn);
        };
        // ---- pass 1: log-joints out, row maximum
        double m = -INFINITY;
        if (!tab && j == 0) {
            const double f0 = P.ecoef * yn + P.pscale * prior[0];
            if (live) out[0] = f0;
            m = f0;
        }
        {
            int c = 0;
            for (int k = 0; k < P.K && !tab; ++k) {
                if (k == P.K0) continue;
                const double v = s_val[k];
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const int h = j + 16 * i;
                    if (h < H) {
                        const double e = v * v * s_w2[h] - 2.0 * v * ar[i] + yn;
                        const double f = P.ecoef * e + P.pscale * prior[1 + c * H + h];
                        if (live) out[1 + c * H + h] = f;
                        m = fmax(m, f);
                    }
                }
                ++c;
            }
        }
        for (int st = j; st < S; st += 16) {
            const double f = P.ecoef * state_e(st) + P.pscale * prior[base + st];
            if (live) out[base + st] = f;
            m = fmax(m, f);
        }
        m = fmax(m, pm_dpp_f64<0xB1>(m));
        m = fmax(m, pm_dpp_f64<0x4E>(m));
        m = fmax(m, pm_dpp_f64<0x141>(m));
        m = fmax(m, pm_dpp_f64<0x140>(m));
        // ---- pass 2: exponentials relative to the maximum; the M-step's sums weighted by them (scaled by 1 / sum below)
        double sum = 0.0, sig_dp = 0.0;
        double cnt_dp[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) cnt_dp[k] = 0.0;
        double rowv[VPL], qd[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) rowv[i] = qd[i] = 0.0;
        if (!tab && j == 0) {
            const double d = (P.ecoef * yn + P.pscale * prior[0]) - m;
            const double ex = d > -37.0 ? pm_exp_tab(d, etab) : 0.0;
            sum += ex;
            sig_dp += ex * yn;
        }
        {
            int c = 0;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                if (tab || k >= P.K || k == P.K0) continue;
                const double v = s_val[k];
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const int h = j + 16 * i;
                    if (h < H) {
                        const double e = v * v * s_w2[h] - 2.0 * v * ar[i] + yn;
                        const double d = (P.ecoef * e + P.pscale * prior[1 + c * H + h]) - m;
                        const double ex = d > -37.0 ? pm_exp_tab(d, etab) : 0.0;
                        sum += ex;
                        rowv[i] += ex * v;
                        qd[i] += ex * v * v;
                        cnt_dp[k] += ex;
                        sig_dp += ex * e;
                    }
                }
                ++c;
            }
        }
        for (int st = j; st < S; st += 16) {
            const double e = state_e(st);
            const double d = (P.ecoef * e + P.pscale * prior[base + st]) - m;
            if (!(d > -37.0)) continue;
            const double ex = pm_exp_tab(d, etab);
            sum += ex;
            sig_dp += ex * e;
            const uint8_t *srow = s_tab + st * Hp;
            int ki[MAXHP];
            double vv[MAXHP];
#pragma unroll
            for (int a = 0; a < MAXHP; ++a) {
                ki[a] = (a < Hp) ? (int)srow[a] : P.K0;
                vv[a] = (a < Hp) ? s_val[ki[a]] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < MAXHP; ++a) {
                if (a < Hp && ki[a] != P.K0) {
#pragma unroll
                    for (int k = 0; k < KM; ++k)
                        if (k == ki[a]) cnt_dp[k] += ex;
                    atomicAdd(&s_m[a], ex * vv[a]);
#pragma unroll
                    for (int k2 = a; k2 < MAXHP; ++k2)
                        if (k2 < Hp && ki[k2] != P.K0) atomicAdd(&s_B[a * Hp + k2], ex * vv[a] * vv[k2]);
                }
            }
        }
        sum += pm_dpp_f64<0xB1>(sum);
        sum += pm_dpp_f64<0x4E>(sum);
        sum += pm_dpp_f64<0x141>(sum);
        sum += pm_dpp_f64<0x140>(sum);
        const double lse_n = m + log(sum);
        const double inv = live ? 1.0 / sum : 0.0;               // (shadow rows: every weight 0)
        if (j == 0 && live) {
            lse[n] = lse_n;
            fs += lse_n;
            kept += 1.0;
        }
        sig += sig_dp * inv;
#pragma unroll
        for (int k = 0; k < KM; ++k) cnt[k] += cnt_dp[k] * inv;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            rowv[i] *= inv;
            if (h < H) {
                const double q2 = qd[i] * inv;
                if (q2 != 0.0) atomicAdd(&s_qdiag[h], PM_Q(q2, 0));
            }
        }
        wave_sync_lds_dsc();
        // the candidates' multi-cause share of E[s]: the lane that holds latent c_a takes s_m[a] (distinct latents)
        for (int a = 0; a < Hp; ++a) {                           // uniform trip count
            const int c = __builtin_amdgcn_ds_bpermute((rowbase + a) << 2, myc);
            const double add = s_m[a] * inv;
            const bool mine = ((lastmask >> a) & 1u) && (c & 15) == j;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (mine && (c >> 4) == i) rowv[i] += add;
        }
        prefetch(n0 + (int64_t)gridDim.x * 16);              // (ahead of the stores below; the scores row is dead here)
        if (live) {
            double *erow = expect + n * lde;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (j + 16 * i < H) erow[j + 16 * i] = rowv[i];
        }
        if (nz_idx) {      // the row's non-zeros as a list too (pm_wp_sparse_f64; format of the BSC statistics pass)
            uint16_t *nzi = nz_idx + nn * PM_BSC_NZ_MAX;
            double *nzv = nz_val + nn * PM_BSC_NZ_MAX;
            uint32_t nzn = 0;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int h = j + 16 * i;
                const double v = h < H ? rowv[i] : 0.0;
                const uint32_t mine = (uint32_t)((__ballot(v != 0.0) >> rowbase) & 0xFFFFull);
                const uint32_t pos = nzn + __builtin_popcount(mine & ((1u << j) - 1u));
                if (v != 0.0 && pos < PM_BSC_NZ_MAX && live) {
                    nzi[pos] = (uint16_t)h;
                    nzv[pos] = v;
                }
                nzn += __builtin_popcount(mine);
            }
            if (live) {
                if ((uint32_t)j >= nzn) nzi[j] = 0xFFFFu;               // (16 lanes = PM_BSC_NZ_MAX slots)
                if (j == 0 && nzn > PM_BSC_NZ_MAX) overflow += 1.0;
            }
        }
        for (int p0 = 0; p0 < Hp * Hp; p0 += 16) {              // uniform trip count: every lane feeds the permutes
            const int p = p0 + j;
            const bool ok = p < Hp * Hp;
            const int a = ok ? p / Hp : 0, k2 = ok ? p - a * Hp : 0;
            const int cj = __builtin_amdgcn_ds_bpermute((rowbase + a) << 2, myc);
            const int ck = __builtin_amdgcn_ds_bpermute((rowbase + k2) << 2, myc);
            if (!ok || k2 < a || !((lastmask >> a) & 1u) || !((lastmask >> k2) & 1u)) continue;
            const double v = s_B[p] * inv;
            if (v == 0.0) continue;
            const int r = cj < ck ? cj : ck, cc = cj < ck ? ck : cj;
            pm_atomic_add(Wq + (int64_t)r * H + cc, PM_Q(v, 0));          // upper triangle (pm_spd_inverse_f64 layout)
        }
        wave_sync_lds_dsc();
    }

    sig = pm_wave_sum(sig);
    fs = pm_wave_sum(fs);
    kept = pm_wave_sum(kept);
    overflow = pm_wave_sum(overflow);
#pragma unroll
    for (int k = 0; k < KM; ++k) cnt[k] = pm_wave_sum(cnt[k]);
    if (lane == 0) {
        atomicAdd(&s_scal[0], PM_Q(sig, 1));
        atomicAdd(&s_scal[1], PM_Q(fs, 2));
        atomicAdd(&s_scal[2], kept);
        if (overflow != 0.0) atomicAdd(&s_scal[3], overflow);
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (cnt[k] != 0.0) atomicAdd(&s_cnt[k], PM_Q(cnt[k], 0));
    }
    __syncthreads();
    double *g_qdiag = stats + (int64_t)H * D + (int64_t)H * H;
    for (int h = tid; h < H + PM_DSC_MAX_K + 4; h += blockDim.x) {
        const double v = s_qdiag[h];
        if (v != 0.0) pm_atomic_add(g_qdiag + h, v);
    }
}

// one wave per datapoint and several dependent memory round trips per datapoint: latency-bound, so waves per SIMD
// are what counts (4 for H' <= 8)
template <int MAXHP>
__global__ __launch_bounds__(64 * WAVES, MAXHP <= 8 ? 4 : 3) void dsc_mstep_rows_kernel(
    const double *__restrict__ logpj, int64_t ldl, const double *__restrict__ lse, double lse_cut,
    const int32_t *__restrict__ cand, const uint8_t *__restrict__ state_idx, int S, const double *__restrict__ prior_g,
    pm_dsc_params P, int64_t N, int H, int D, int Hp, double *__restrict__ expect, int64_t lde,
    double *__restrict__ stats, int stage, const double *__restrict__ cut_dev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (cut_dev) lse_cut = cut_dev[0];       // (round 6: the radix select's result read where it was left, no host round trip)
    // [ qdiag (H) | cnt (8) | scal (4) | per wave: row (H) m (Hp) B (Hp*Hp) | state table ]
    double *s_qdiag = reinterpret_cast<double *>(smem);
    double *s_cnt = s_qdiag + H;
    double *s_scal = s_cnt + PM_DSC_MAX_K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_wave = H + Hp + Hp * Hp;
    double *s_row = s_scal + 4 + wave * per_wave;
    double *s_m = s_row + H;
    double *s_B = s_m + Hp;
    uint8_t *s_tab = reinterpret_cast<uint8_t *>(s_scal + 4 + WAVES * per_wave);
    __shared__ double s_val[PM_DSC_MAX_K];
    __shared__ double s_E[128];                          // (pm_exp_tab's table: see dsc_estep_kernel)
    if (tid < 128) s_E[tid] = pm_powtab_dev[256 + tid];
    const double *etab = s_E - 256;
    for (int h = tid; h < H + PM_DSC_MAX_K + 4; h += blockDim.x) s_qdiag[h] = 0.0;
    for (int i = tid; i < S * Hp; i += blockDim.x) s_tab[i] = state_idx[i];
    if (tid < PM_DSC_MAX_K) s_val[tid] = (tid < P.K) ? P.values[tid] : 0.0;
    __syncthreads();

    const int nss = (P.K - 1) * H;
    const bool tab = P.flags & PM_DSC_TABLE_ONLY;
    const int base = tab ? 0 : 1 + nss;
    // staged (when it fits): the log-prior table in LDS behind the state table, as in the E-step kernel
    double *s_prior = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(s_tab) + (((size_t)S * Hp + 7) & ~size_t(7)));
    if (stage) {
        for (int i = tid; i < base + S; i += blockDim.x) s_prior[i] = prior_g[i];
        __syncthreads();
    }
    const double *prior = stage ? s_prior : prior_g;
    const double inv_ecoef = 1.0 / P.ecoef;
    const double qcut = -60.0;     // multi-cause weights below e^-60 of the evidence add nothing in f64
    double sig = 0.0, fs = 0.0, kept = 0.0;
    double cnt[PM_DSC_MAX_K];
#pragma unroll
    for (int k = 0; k < PM_DSC_MAX_K; ++k) cnt[k] = 0.0;

    for (int64_t n = (int64_t)blockIdx.x * WAVES + wave; n < N; n += (int64_t)gridDim.x * WAVES) {
        double *erow = expect + n * lde;
        const double l = lse[n];
        if (!(l > lse_cut)) {   // truncated: strictly-greater rule of dsc_et.py:832
            for (int h = lane; h < H; h += 64) erow[h] = 0.0;
            continue;
        }
        const double *f = logpj + n * ldl;
        const int32_t *cn = cand + n * Hp;
        if (lane < Hp) s_m[lane] = 0.0;
        for (int p = lane; p < Hp * Hp; p += 64) s_B[p] = 0.0;
        // positions that are the last occurrence of their latent (all of them unless candidates repeat)
        bool mine_last = lane < Hp;
        const int myc = lane < Hp ? cn[lane] : -1;       // ONE load of the candidates; the comparisons are shuffles
        for (int k = 1; k < Hp; ++k) {                     // uniform trip count
            const int other = __shfl(myc, k);
            if (k > lane && other == myc) mine_last = false;
        }
        const unsigned long long lastmask =
            (P.flags & PM_DSC_LAST_POSITION) ? __ballot(mine_last) : ((1ull << Hp) - 1ull);
        if (lane == 0) {
            if (!tab) {
                const double f0 = f[0];
                sig += (f0 - l > qcut ? pm_exp_tab(f0 - l, etab) : 0.0) * ((f0 - P.pscale * prior[0]) * inv_ecoef);
            }
            fs += l;
            kept += 1.0;
        }
        for (int h = lane; h < H; h += 64) {
            double row = 0.0, qd = 0.0;
            int c = 0;
#pragma unroll
            for (int k = 0; k < PM_DSC_MAX_K; ++k) {
                if (tab || k >= P.K || k == P.K0) continue;
                const double fh = f[1 + c * H + h];
                const double dq = fh - l;
                const double q = dq > qcut ? pm_exp_tab(dq, etab) : 0.0;   // (below e^-60 of the evidence: nothing in f64)
                const double v = P.values[k];
                row += q * v;
                qd += q * v * v;
                cnt[k] += q;
                sig += q * ((fh - P.pscale * prior[1 + c * H + h]) * inv_ecoef);
                ++c;
            }
            s_row[h] = row;
            if (qd != 0.0) atomicAdd(&s_qdiag[h], PM_Q(qd, 0));
        }
        wave_sync_lds_dsc();
        for (int s = lane; s < S; s += 64) {
            const double fsv = f[base + s];
            const double dl = fsv - l;
            if (!(dl > qcut)) continue;
            const double q = pm_exp_tab(dl, etab);
            sig += q * ((fsv - P.pscale * prior[base + s]) * inv_ecoef);
            const uint8_t *row = s_tab + s * Hp;
            // the state's value indices and values first (LDS lookups, static register indices), then the scatter
            int ki[MAXHP];
            double vv[MAXHP];
#pragma unroll
            for (int j = 0; j < MAXHP; ++j) {
                ki[j] = (j < Hp) ? (int)row[j] : P.K0;
                vv[j] = (j < Hp) ? s_val[ki[j]] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < MAXHP; ++j) {
                if (j < Hp && ki[j] != P.K0) {
#pragma unroll
                    for (int k = 0; k < PM_DSC_MAX_K; ++k)
                        if (k == ki[j]) cnt[k] += q;
                    atomicAdd(&s_m[j], q * vv[j]);
#pragma unroll
                    for (int k2 = j; k2 < MAXHP; ++k2)
                        if (k2 < Hp && ki[k2] != P.K0) atomicAdd(&s_B[j * Hp + k2], q * vv[j] * vv[k2]);
                }
            }
        }
        wave_sync_lds_dsc();
        if (lane < Hp && ((lastmask >> lane) & 1ull)) s_row[myc] += s_m[lane];   // distinct latents
        wave_sync_lds_dsc();
        for (int h = lane; h < H; h += 64) erow[h] = s_row[h];
        double *Wq = stats + (int64_t)H * D;
        for (int p0 = 0; p0 < Hp * Hp; p0 += 64) {          // uniform trip count: every lane feeds the shuffles
            const int p = p0 + lane;
            const bool ok = p < Hp * Hp;
            const int j = ok ? p / Hp : 0, k2 = ok ? p - j * Hp : 0;
            const int cj = __shfl(myc, j), ck = __shfl(myc, k2);
            if (!ok || k2 < j || !((lastmask >> j) & 1ull) || !((lastmask >> k2) & 1ull)) continue;
            const double v = s_B[p];
            if (v == 0.0) continue;
            const int r = cj < ck ? cj : ck, cc = cj < ck ? ck : cj;
            pm_atomic_add(Wq + (int64_t)r * H + cc, PM_Q(v, 0));      // upper triangle (pm_spd_inverse_f64 layout)
        }
        wave_sync_lds_dsc();
    }

    sig = pm_wave_sum(sig);
    fs = pm_wave_sum(fs);
    kept = pm_wave_sum(kept);
#pragma unroll
    for (int k = 0; k < PM_DSC_MAX_K; ++k) cnt[k] = pm_wave_sum(cnt[k]);
    if (lane == 0) {
        atomicAdd(&s_scal[0], PM_Q(sig, 1));
        atomicAdd(&s_scal[1], PM_Q(fs, 2));
        atomicAdd(&s_scal[2], kept);
#pragma unroll
        for (int k = 0; k < PM_DSC_MAX_K; ++k)
            if (cnt[k] != 0.0) atomicAdd(&s_cnt[k], PM_Q(cnt[k], 0));
    }
    __syncthreads();
    double *g_qdiag = stats + (int64_t)H * D + (int64_t)H * H;
    for (int h = tid; h < H + PM_DSC_MAX_K + 4; h += blockDim.x) {
        const double v = s_qdiag[h];
        if (v != 0.0) pm_atomic_add(g_qdiag + h, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dsc_mstep_rows_kernel with sixteen lanes per datapoint (see dsc_estep16_kernel): the log-joints of a datapoint are
// requested in batches (VPL singleton columns per value, eight state columns at a time) instead of one load per loop
// trip, the candidate bookkeeping and the reductions are per 16-lane row.
// LDS: [ qdiag (H) cnt (8) scal (4) | state table | prior (K, when staged) | per row: E[s] row (H) m (H') B (H'^2) ]
// ---------------------------------------------------------------------------------------------------------------
template <int MAXHP, int VPL>
__global__ __launch_bounds__(256, (MAXHP <= 8 && VPL <= 8) ? PM_DSC_M16_WPE : 2) void dsc_mstep_rows16_kernel(
    const double *__restrict__ logpj, int64_t ldl, const double *__restrict__ lse, double lse_cut,
    const int32_t *__restrict__ cand, const uint8_t *__restrict__ state_idx, int S, const double *__restrict__ prior_g,
    pm_dsc_params P, int64_t N, int H, int D, int Hp, double *__restrict__ expect, int64_t lde,
    double *__restrict__ stats, int stage, uint16_t *__restrict__ nz_idx, double *__restrict__ nz_val,
    const double *__restrict__ cut_dev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (cut_dev) lse_cut = cut_dev[0];       // (round 6: the radix select's result read where it was left, no host round trip)
    double *s_qdiag = reinterpret_cast<double *>(smem);
    double *s_cnt = s_qdiag + H;
    double *s_scal = s_cnt + PM_DSC_MAX_K;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, row = lane >> 4, rowbase = lane & 48;
    double overflow = 0.0;
    const int nss = (P.K - 1) * H;
    const bool tab = P.flags & PM_DSC_TABLE_ONLY;
    const int base = tab ? 0 : 1 + nss;
    const int Kt = base + S;
    uint8_t *s_tab = reinterpret_cast<uint8_t *>(s_scal + 4);
    double *s_prior = reinterpret_cast<double *>(s_tab + (((size_t)S * Hp + 7) & ~size_t(7)));
    const int per_row = H + Hp + Hp * Hp;
    double *s_row = s_prior + (stage ? Kt : 0) + (size_t)(wave * 4 + row) * per_row;
    double *s_m = s_row + H;
    double *s_B = s_m + Hp;
    __shared__ double s_val[PM_DSC_MAX_K];
    __shared__ double s_E[128];
    if (tid < 128) s_E[tid] = pm_powtab_dev[256 + tid];
    const double *etab = s_E - 256;
    for (int h = tid; h < H + PM_DSC_MAX_K + 4; h += blockDim.x) s_qdiag[h] = 0.0;
    for (int i = tid; i < S * Hp; i += blockDim.x) s_tab[i] = state_idx[i];
    if (tid < PM_DSC_MAX_K) s_val[tid] = (tid < P.K) ? P.values[tid] : 0.0;
    if (stage)
        for (int i = tid; i < Kt; i += blockDim.x) s_prior[i] = prior_g[i];
    __syncthreads();
    const double *prior = stage ? s_prior : prior_g;
    const double inv_ecoef = 1.0 / P.ecoef;
    const double qcut = -60.0;     // weights below e^-60 of the evidence add nothing in f64
    double sig = 0.0, fs = 0.0, kept = 0.0;
    double cnt[PM_DSC_MAX_K];
#pragma unroll
    for (int k = 0; k < PM_DSC_MAX_K; ++k) cnt[k] = 0.0;
    double *Wq = stats + (int64_t)H * D;

    for (int64_t n0 = (int64_t)blockIdx.x * 16; n0 < N; n0 += (int64_t)gridDim.x * 16) {
        const int64_t n = n0 + wave * 4 + row;
        const bool live = n < N;
        const int64_t nn = live ? n : N - 1;
        const double l_n = lse[nn];
        const bool use = live && (l_n > lse_cut);             // truncated: strictly-greater rule of dsc_et.py:832
        const double l = use ? l_n : INFINITY;                  // (every weight of a dropped / shadow row is exp(-inf) = 0)
        const double *f = logpj + nn * ldl;
        const int myc = j < Hp ? cand[nn * Hp + j] : -1;
        if (j < Hp) s_m[j] = 0.0;
        for (int p = j; p < Hp * Hp; p += 16) s_B[p] = 0.0;
        // positions that are the last occurrence of their latent (all of them unless candidates repeat)
        bool mine_last = j < Hp;
        for (int k = 1; k < Hp; ++k) {                          // uniform trip count
            const int other = __builtin_amdgcn_ds_bpermute((rowbase + k) << 2, myc);
            if (k > j && other == myc) mine_last = false;
        }
        const unsigned lastmask = (P.flags & PM_DSC_LAST_POSITION)
                                      ? (unsigned)((__ballot(mine_last) >> rowbase) & 0xFFFFull)
                                      : ((1u << Hp) - 1u);
        if (j == 0 && use) {
            if (!tab) {
                const double f0 = f[0];
                sig += (f0 - l > qcut ? pm_exp_tab(f0 - l, etab) : 0.0) * ((f0 - P.pscale * prior[0]) * inv_ecoef);
            }
            fs += l;
            kept += 1.0;
        }
        // singletons: E[s_h] and its second moment over the values (dsc_et.py:660-700)
        {
            double rowv[VPL], qd[VPL];
#pragma unroll
            for (int i = 0; i < VPL; ++i) rowv[i] = qd[i] = 0.0;
            int c = 0;
#pragma unroll
            for (int k = 0; k < PM_DSC_MAX_K; ++k) {
                if (tab || k >= P.K || k == P.K0) continue;
                const double v = s_val[k];
                double fv[VPL];
#pragma unroll
                for (int i = 0; i < VPL; ++i) fv[i] = (j + 16 * i < H) ? f[1 + c * H + j + 16 * i] : -INFINITY;
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const int h = j + 16 * i;
                    if (h < H) {
                        const double dq = fv[i] - l;
                        const double q = dq > qcut ? pm_exp_tab(dq, etab) : 0.0;
                        rowv[i] += q * v;
                        qd[i] += q * v * v;
                        cnt[k] += q;
                        sig += q * ((fv[i] - P.pscale * prior[1 + c * H + h]) * inv_ecoef);
                    }
                }
                ++c;
            }
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int h = j + 16 * i;
                if (h < H) {
                    s_row[h] = rowv[i];
                    if (qd[i] != 0.0) atomicAdd(&s_qdiag[h], PM_Q(qd[i], 0));
                }
            }
        }
        wave_sync_lds_dsc();
        // multi-cause states, eight columns per lane at a time
        for (int s0 = 0; s0 < S; s0 += 128) {
            double fv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int st = s0 + j + 16 * u;
                fv[u] = st < S ? f[base + st] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int st = s0 + j + 16 * u;
                const double dl = fv[u] - l;
                if (!(dl > qcut)) continue;
                const double q = pm_exp_tab(dl, etab);
                sig += q * ((fv[u] - P.pscale * prior[base + st]) * inv_ecoef);
                const uint8_t *srow = s_tab + st * Hp;
                int ki[MAXHP];
                double vv[MAXHP];
#pragma unroll
                for (int a = 0; a < MAXHP; ++a) {
                    ki[a] = (a < Hp) ? (int)srow[a] : P.K0;
                    vv[a] = (a < Hp) ? s_val[ki[a]] : 0.0;
                }
#pragma unroll
                for (int a = 0; a < MAXHP; ++a) {
                    if (a < Hp && ki[a] != P.K0) {
#pragma unroll
                        for (int k = 0; k < PM_DSC_MAX_K; ++k)
                            if (k == ki[a]) cnt[k] += q;
                        atomicAdd(&s_m[a], q * vv[a]);
#pragma unroll
                        for (int k2 = a; k2 < MAXHP; ++k2)
                            if (k2 < Hp && ki[k2] != P.K0) atomicAdd(&s_B[a * Hp + k2], q * vv[a] * vv[k2]);
                    }
                }
            }
        }
        wave_sync_lds_dsc();
        if (j < Hp && ((lastmask >> j) & 1u)) s_row[myc] += s_m[j];   // distinct latents
        wave_sync_lds_dsc();
        if (live) {
            double *erow = expect + n * lde;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (j + 16 * i < H) erow[j + 16 * i] = s_row[j + 16 * i];
        }
        if (nz_idx) {      // the row's non-zeros as a list too (pm_wp_sparse_f64; format of the BSC statistics pass)
            uint16_t *nzi = nz_idx + nn * PM_BSC_NZ_MAX;
            double *nzv = nz_val + nn * PM_BSC_NZ_MAX;
            uint32_t nzn = 0;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int h = j + 16 * i;
                const double v = h < H ? s_row[h] : 0.0;
                const uint32_t mine = (uint32_t)((__ballot(v != 0.0) >> rowbase) & 0xFFFFull);
                const uint32_t pos = nzn + __builtin_popcount(mine & ((1u << j) - 1u));
                if (v != 0.0 && pos < PM_BSC_NZ_MAX && live) {
                    nzi[pos] = (uint16_t)h;
                    nzv[pos] = v;
                }
                nzn += __builtin_popcount(mine);
            }
            if (live) {
                if ((uint32_t)j >= nzn) nzi[j] = 0xFFFFu;               // (16 lanes = PM_BSC_NZ_MAX slots)
                if (j == 0 && nzn > PM_BSC_NZ_MAX) overflow += 1.0;
            }
        }
        for (int p0 = 0; p0 < Hp * Hp; p0 += 16) {              // uniform trip count: every lane feeds the permutes
            const int p = p0 + j;
            const bool ok = p < Hp * Hp;
            const int a = ok ? p / Hp : 0, k2 = ok ? p - a * Hp : 0;
            const int cj = __builtin_amdgcn_ds_bpermute((rowbase + a) << 2, myc);
            const int ck = __builtin_amdgcn_ds_bpermute((rowbase + k2) << 2, myc);
            if (!ok || k2 < a || !((lastmask >> a) & 1u) || !((lastmask >> k2) & 1u)) continue;
            const double v = s_B[p];
            if (v == 0.0) continue;
            const int r = cj < ck ? cj : ck, cc = cj < ck ? ck : cj;
            pm_atomic_add(Wq + (int64_t)r * H + cc, PM_Q(v, 0));          // upper triangle (pm_spd_inverse_f64 layout)
        }
        wave_sync_lds_dsc();
    }

    sig = pm_wave_sum(sig);
    fs = pm_wave_sum(fs);
    kept = pm_wave_sum(kept);
    overflow = pm_wave_sum(overflow);
#pragma unroll
    for (int k = 0; k < PM_DSC_MAX_K; ++k) cnt[k] = pm_wave_sum(cnt[k]);
    if (lane == 0) {
        atomicAdd(&s_scal[0], PM_Q(sig, 1));
        atomicAdd(&s_scal[1], PM_Q(fs, 2));
        atomicAdd(&s_scal[2], kept);
        if (overflow != 0.0) atomicAdd(&s_scal[3], overflow);
#pragma unroll
        for (int k = 0; k < PM_DSC_MAX_K; ++k)
            if (cnt[k] != 0.0) atomicAdd(&s_cnt[k], PM_Q(cnt[k], 0));
    }
    __syncthreads();
    double *g_qdiag = stats + (int64_t)H * D + (int64_t)H * H;
    for (int h = tid; h < H + PM_DSC_MAX_K + 4; h += blockDim.x) {
        const double v = s_qdiag[h];
        if (v != 0.0) pm_atomic_add(g_qdiag + h, v);
    }
}

inline size_t align8(size_t x) { return (x + 7) & ~size_t(7); }

inline int allow_lds_dsc(const void *kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

inline bool bad_params(const pm_dsc_params *P) {
    return !P || P->K < 2 || P->K > PM_DSC_MAX_K || P->K0 < 0 || P->K0 >= P->K || P->values[P->K0] != 0.0;
}

// one round of resident workgroups (the kernels stride over the datapoints): `per_cu` workgroups fit a CU at the
// kernel's register count, and a second, partly filled round would only add a tail
inline unsigned row_grid(int64_t N, int per_cu) {
    const int64_t blocks = (N + WAVES - 1) / WAVES;
    const int64_t cap = 256 * (int64_t)per_cu;
    return (unsigned)(blocks < 1 ? 1 : blocks > cap ? cap : blocks);
}

}  // namespace

extern "C" int64_t pm_dsc_stats_len(int64_t H, int64_t D) { return H * D + H * H + H + PM_DSC_MAX_K + 4; }

extern "C" int pm_dsc_select_scores_f64(const double *scores, int64_t lds, const double *gram,
                                        const pm_dsc_params *params_host, int64_t N, int64_t H, double *R,
                                        int64_t ldr, void *stream) {
    if (N == 0) return PM_OK;
    if (!scores || !gram || !R || N < 0 || H <= 0 || lds < H || ldr < H || bad_params(params_host)) return PM_EINVAL;
    if (H > INT32_MAX) return PM_ERANGE;
    const int64_t total = N * H;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dsc_select_scores_kernel, dim3((unsigned)(blocks > 256 * 16 ? 256 * 16 : blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), scores, lds, gram, *params_host, N, (int)H, R, ldr);
    return (int)hipGetLastError();
}

extern "C" int pm_dsc_estep_f64(const double *scores, int64_t lds, const double *gram, const double *ynorm2,
                                const int32_t *cand, const uint8_t *state_idx, int64_t S, const double *prior,
                                const pm_dsc_params *params_host, int64_t N, int64_t H, int64_t Hprime, double *logpj,
                                int64_t ldl, double *lse, void *stream) {
    if (N == 0) return PM_OK;
    if (!scores || !gram || !ynorm2 || !cand || !prior || !logpj || !lse || N < 0 || H <= 0 || Hprime <= 0 || S < 0 ||
        lds < H || bad_params(params_host) || (S > 0 && !state_idx))
        return PM_EINVAL;
    if (ldl < ((params_host->flags & PM_DSC_TABLE_ONLY) ? S : 1 + (params_host->K - 1) * H + S)) return PM_EINVAL;
    if (Hprime > PM_MAX_HPRIME || Hprime > H || H > 65536) return PM_ERANGE;
    const int64_t Kt = (params_host->flags & PM_DSC_TABLE_ONLY) ? S : 1 + (params_host->K - 1) * H + S;
#ifndef PM_DSC_WAVE64
    {
        // sixteen lanes per datapoint where its LDS layout fits four workgroups per CU
        const int64_t Kn16 = params_host->K - 1;
        int64_t NT16 = 1 + Hprime * Kn16 + Hprime * (Hprime - 1) / 2 * Kn16 * Kn16;
        if (NT16 > 256 || S == 0) NT16 = 0;
        if (S * Hprime < (1 << 20) && Kt < (1 << 20)) {
            int stage16 = 1;
            Lay16 L = dsc_lay16((int)H, (int)Hprime, (int)S, (int)Kt, (int)NT16, stage16);
            if (L.bytes > 40 * 1024) {
                stage16 = 0;
                L = dsc_lay16((int)H, (int)Hprime, (int)S, (int)Kt, (int)NT16, stage16);
            }
            if (L.bytes <= 40 * 1024 && H <= 256) {
                const int64_t blocks16 = (N + 15) / 16;
                const unsigned grid16 = (unsigned)(blocks16 < 256 * 4 ? blocks16 : 256 * 4);
#define PM_LAUNCH16(M, V)                                                                                              \
    do {                                                                                                               \
        if (int e = allow_lds_dsc(reinterpret_cast<const void *>(dsc_estep16_kernel<M, V>), (size_t)L.bytes)) return e; \
        hipLaunchKernelGGL((dsc_estep16_kernel<M, V>), dim3(grid16), dim3(256), (size_t)L.bytes,                       \
                           static_cast<hipStream_t>(stream), scores, lds, gram, ynorm2, cand, state_idx, (int)S, prior, \
                           *params_host, N, (int)H, (int)Hprime, logpj, ldl, lse, stage16, (int)NT16);                 \
    } while (0)
                if (Hprime <= 8 && H <= 128) PM_LAUNCH16(8, 8);
                else if (Hprime <= 8) PM_LAUNCH16(8, 16);
                else if (H <= 128) PM_LAUNCH16(PM_MAX_HPRIME, 8);
                else PM_LAUNCH16(PM_MAX_HPRIME, 16);
#undef PM_LAUNCH16
                return (int)hipGetLastError();
            }
        }
    }
#endif
    size_t shmem = sizeof(double) * (H + WAVES * (Hprime + Hprime * Hprime)) + align8((size_t)S * Hprime);
    if (shmem > 150 * 1024) return PM_ERANGE;
    const size_t staged = shmem + sizeof(double) * (size_t)(WAVES + 1) * (size_t)Kt;
    const int stage = staged <= 30 * 1024 ? 1 : 0;       // keep five workgroups per CU
    if (stage) shmem = staged;
    // the energy-term tables (see the kernel): where they fit beside the rest
    const int64_t Kn = params_host->K - 1;
    int64_t NT = 1 + Hprime * Kn + Hprime * (Hprime - 1) / 2 * Kn * Kn;
    const size_t fast_off = align8(shmem);
    const size_t fast_bytes = (size_t)S * 8 + 2 * (size_t)((NT + 7) & ~7) + sizeof(double) * (size_t)NT * (2 + WAVES);
    if (NT > 256 || S == 0 || fast_off + fast_bytes > 40 * 1024) NT = 0;
    else shmem = fast_off + fast_bytes;
#define PM_LAUNCH(M)                                                                                                 \
    do {                                                                                                             \
        if (int e = allow_lds_dsc(reinterpret_cast<const void *>(dsc_estep_kernel<M>), shmem)) return e;             \
        hipLaunchKernelGGL(dsc_estep_kernel<M>, dim3(row_grid(N, M <= 8 ? 5 : 4)), dim3(64 * WAVES), shmem,            \
                           static_cast<hipStream_t>(stream), scores, lds, gram, ynorm2, cand, state_idx, (int)S, prior, \
                           *params_host, N, (int)H, (int)Hprime, logpj, ldl, lse, stage, (int)fast_off, (int)NT);   \
    } while (0)
    if (Hprime <= 8) PM_LAUNCH(8);
    else PM_LAUNCH(PM_MAX_HPRIME);
#undef PM_LAUNCH
    return (int)hipGetLastError();
}

// E-step + M-step row statistics in one pass (dsc_estep16_ms_kernel): its LDS layout, or 0 where it does not apply
static size_t dsc_estep_ms_lds(int64_t H, int64_t Hprime, int64_t S, int64_t Kt, int64_t Kn, int *stage16, int *nt16) {
#ifdef PM_DSC_WAVE64
    return 0;
#else
    if (!(H <= 256 && S * Hprime < (1 << 20) && Kt < (1 << 20))) return 0;
    int64_t NT16 = 1 + Hprime * Kn + Hprime * (Hprime - 1) / 2 * Kn * Kn;
    if (NT16 > 256 || S == 0) NT16 = 0;
    *stage16 = 1;
    Lay16 L = dsc_lay16((int)H, (int)Hprime, (int)S, (int)Kt, (int)NT16, 1);
    if (L.bytes > 40 * 1024) {
        *stage16 = 0;
        L = dsc_lay16((int)H, (int)Hprime, (int)S, (int)Kt, (int)NT16, 0);
    }
    if (L.bytes > 40 * 1024) return 0;
    *nt16 = (int)NT16;
    const size_t total = (size_t)((L.bytes + 7) & ~7) + sizeof(double) * (size_t)(H + PM_DSC_MAX_K + 4) +
                         sizeof(double) * 16 * (size_t)(Hprime + Hprime * Hprime);
    return total <= 64 * 1024 ? total : 0;
#endif
}

extern "C" int pm_dsc_estep_mstats_supported(int64_t H, int64_t Hprime, int64_t S, int64_t K, int flags) {
    if (H <= 0 || Hprime <= 0 || Hprime > PM_MAX_HPRIME || Hprime > H || S < 0 || K < 2 || K > PM_DSC_MAX_K) return 0;
    int st = 0, nt = 0;
    return dsc_estep_ms_lds(H, Hprime, S, (flags & PM_DSC_TABLE_ONLY) ? S : 1 + (K - 1) * H + S, K - 1, &st, &nt) ? 1 : 0;
}

extern "C" int pm_dsc_estep_mstats_f64(const double *scores, int64_t lds, const double *gram, const double *ynorm2,
                                       const int32_t *cand, const uint8_t *state_idx, int64_t S, const double *prior,
                                       const pm_dsc_params *params_host, int64_t N, int64_t H, int64_t D, int64_t Hprime,
                                       double *logpj, int64_t ldl, double *lse, double *expect, int64_t lde,
                                       double *stats, uint16_t *nz_idx, double *nz_val, void *stream) {
    if ((nz_idx == nullptr) != (nz_val == nullptr)) return PM_EINVAL;
    if (N == 0) return PM_OK;
    if (!scores || !gram || !ynorm2 || !cand || !prior || !logpj || !lse || !expect || !stats || N < 0 || H <= 0 || D <= 0 ||
        Hprime <= 0 || S < 0 || lds < H || lde < H || bad_params(params_host) || (S > 0 && !state_idx) ||
        params_host->ecoef == 0.0)
        return PM_EINVAL;
    const int64_t Kt = (params_host->flags & PM_DSC_TABLE_ONLY) ? S : 1 + (params_host->K - 1) * H + S;
    if (ldl < Kt) return PM_EINVAL;
    if (Hprime > PM_MAX_HPRIME || Hprime > H) return PM_ERANGE;
    int stage16 = 0, NT16 = 0;
    const size_t shmem = dsc_estep_ms_lds(H, Hprime, S, Kt, params_host->K - 1, &stage16, &NT16);
    if (!shmem) return PM_ERANGE;
    const int64_t blocks16 = (N + 15) / 16;
    const int per_cu = (Hprime <= 8 && H <= 128) ? PM_DSC_MS_WPE : 2;
    const unsigned grid16 = (unsigned)(blocks16 < 256 * per_cu ? blocks16 : 256 * per_cu);
#define PM_LAUNCH16K(M, V, KMV)                                                                                        \
    do {                                                                                                               \
        if (int e = allow_lds_dsc(reinterpret_cast<const void *>(dsc_estep16_ms_kernel<M, V, KMV>), shmem)) return e;  \
        hipLaunchKernelGGL((dsc_estep16_ms_kernel<M, V, KMV>), dim3(grid16), dim3(256), shmem,                         \
                           static_cast<hipStream_t>(stream), scores, lds, gram, ynorm2, cand, state_idx, (int)S, prior, \
                           *params_host, N, (int)H, (int)D, (int)Hprime, logpj, ldl, lse, stage16, NT16, expect, lde,   \
                           stats, nz_idx, nz_val);                                                                     \
    } while (0)
#define PM_LAUNCH16(M, V)                        \
    do {                                         \
        if (params_host->K <= 4) {               \
            PM_LAUNCH16K(M, V, 4);               \
        } else {                                 \
            PM_LAUNCH16K(M, V, PM_DSC_MAX_K);    \
        }                                        \
    } while (0)
    if (Hprime <= 8 && H <= 128) PM_LAUNCH16(8, 8);
    else if (Hprime <= 8) PM_LAUNCH16(8, 16);
    else if (H <= 128) PM_LAUNCH16(PM_MAX_HPRIME, 8);
    else PM_LAUNCH16(PM_MAX_HPRIME, 16);
#undef PM_LAUNCH16
#undef PM_LAUNCH16K
    return (int)hipGetLastError();
}

// the launch geometry of dsc_mstep_rows16_kernel, or 0 bytes where it does not apply
static size_t dsc_rows16_lds(int64_t H, int64_t Hprime, int64_t S, int64_t Kt, int *stage16) {
#ifdef PM_DSC_WAVE64
    return 0;
#else
    if (!(H <= 256 && S * Hprime < (1 << 20) && Kt < (1 << 20))) return 0;
    const size_t fixed = sizeof(double) * (H + PM_DSC_MAX_K + 4) + align8((size_t)S * Hprime);
    const size_t rows16 = sizeof(double) * 16 * (size_t)(H + Hprime + Hprime * Hprime);
    *stage16 = 1;
    size_t sh16 = fixed + sizeof(double) * (size_t)Kt + rows16;
    if (sh16 > 40 * 1024) {
        *stage16 = 0;
        sh16 = fixed + rows16;
    }
    return sh16 <= 40 * 1024 ? sh16 : 0;
#endif
}

extern "C" int pm_dsc_rows16_supported(int64_t H, int64_t Hprime, int64_t S, int64_t K, int flags) {
    if (H <= 0 || Hprime <= 0 || Hprime > PM_MAX_HPRIME || S < 0 || K < 2 || K > PM_DSC_MAX_K) return 0;
    int st = 0;
    return dsc_rows16_lds(H, Hprime, S, (flags & PM_DSC_TABLE_ONLY) ? S : 1 + (K - 1) * H + S, &st) ? 1 : 0;
}

extern "C" int pm_dsc_mstep_rows_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                                     const int32_t *cand, const uint8_t *state_idx, int64_t S, const double *prior,
                                     const pm_dsc_params *params_host, int64_t N, int64_t H, int64_t D, int64_t Hprime,
                                     double *expect, int64_t lde, double *stats, void *stream) {
    return pm_dsc_mstep_rows_nz_f64(logpj, ldl, lse, lse_cut, cand, state_idx, S, prior, params_host, N, H, D, Hprime,
                                    expect, lde, stats, nullptr, nullptr, stream);
}

extern "C" int pm_dsc_mstep_rows_nz_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                                        const int32_t *cand, const uint8_t *state_idx, int64_t S, const double *prior,
                                        const pm_dsc_params *params_host, int64_t N, int64_t H, int64_t D,
                                        int64_t Hprime, double *expect, int64_t lde, double *stats, uint16_t *nz_idx,
                                        double *nz_val, void *stream) {
    return pm_dsc_mstep_rows_cutp_f64(logpj, ldl, lse, lse_cut, nullptr, cand, state_idx, S, prior, params_host, N, H, D, Hprime,
                                      expect, lde, stats, nz_idx, nz_val, stream);
}

extern "C" int pm_dsc_mstep_rows_cutp_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                                          const double *cut_dev, const int32_t *cand, const uint8_t *state_idx, int64_t S,
                                          const double *prior, const pm_dsc_params *params_host, int64_t N, int64_t H,
                                          int64_t D, int64_t Hprime, double *expect, int64_t lde, double *stats,
                                          uint16_t *nz_idx, double *nz_val, void *stream) {
    if ((nz_idx == nullptr) != (nz_val == nullptr)) return PM_EINVAL;
    if (N == 0) return PM_OK;
    if (!logpj || !lse || !cand || !prior || !expect || !stats || N < 0 || H <= 0 || D <= 0 || Hprime <= 0 || S < 0 ||
        lde < H || bad_params(params_host) || (S > 0 && !state_idx))
        return PM_EINVAL;
    if (ldl < ((params_host->flags & PM_DSC_TABLE_ONLY) ? S : 1 + (params_host->K - 1) * H + S) ||
        params_host->ecoef == 0.0)
        return PM_EINVAL;
    if (Hprime > PM_MAX_HPRIME || Hprime > H || H > 65536) return PM_ERANGE;
    const int64_t Kt = (params_host->flags & PM_DSC_TABLE_ONLY) ? S : 1 + (params_host->K - 1) * H + S;
    {
        // sixteen lanes per datapoint where the layout fits four workgroups per CU
        int stage16 = 0;
        const size_t sh16 = dsc_rows16_lds(H, Hprime, S, Kt, &stage16);
        if (sh16) {
            const int64_t blocks16 = (N + 15) / 16;
            const unsigned grid16 = (unsigned)(blocks16 < 256 * 4 ? blocks16 : 256 * 4);
#define PM_LAUNCH16(M, V)                                                                                              \
    do {                                                                                                               \
        if (int e = allow_lds_dsc(reinterpret_cast<const void *>(dsc_mstep_rows16_kernel<M, V>), sh16)) return e;      \
        hipLaunchKernelGGL((dsc_mstep_rows16_kernel<M, V>), dim3(grid16), dim3(256), sh16,                             \
                           static_cast<hipStream_t>(stream), logpj, ldl, lse, lse_cut, cand, state_idx, (int)S, prior, \
                           *params_host, N, (int)H, (int)D, (int)Hprime, expect, lde, stats, stage16, nz_idx, nz_val,  \
                           cut_dev);                                                                                   \
    } while (0)
            if (Hprime <= 8 && H <= 128) PM_LAUNCH16(8, 8);
            else if (Hprime <= 8) PM_LAUNCH16(8, 16);
            else if (H <= 128) PM_LAUNCH16(PM_MAX_HPRIME, 8);
            else PM_LAUNCH16(PM_MAX_HPRIME, 16);
#undef PM_LAUNCH16
            return (int)hipGetLastError();
        }
    }
    if (nz_idx) return PM_ERANGE;       // (lists come from the sixteen-lane kernel only: pm_dsc_rows16_supported)
    size_t shmem = sizeof(double) * (H + PM_DSC_MAX_K + 4 + WAVES * (H + Hprime + Hprime * Hprime)) +
                   align8((size_t)S * Hprime);
    if (shmem > 150 * 1024) return PM_ERANGE;
    const int stage = shmem + sizeof(double) * (size_t)Kt <= 36 * 1024 ? 1 : 0;     // four workgroups per CU stay
    if (stage) shmem += sizeof(double) * (size_t)Kt;
#define PM_LAUNCH(M)                                                                                                 \
    do {                                                                                                             \
        if (int e = allow_lds_dsc(reinterpret_cast<const void *>(dsc_mstep_rows_kernel<M>), shmem)) return e;        \
        hipLaunchKernelGGL(dsc_mstep_rows_kernel<M>, dim3(row_grid(N, M <= 8 ? 4 : 3)), dim3(64 * WAVES), shmem,       \
                           static_cast<hipStream_t>(stream), logpj, ldl, lse, lse_cut, cand, state_idx, (int)S, prior, \
                           *params_host, N, (int)H, (int)D, (int)Hprime, expect, lde, stats, stage, cut_dev);        \
    } while (0)
    if (Hprime <= 8) PM_LAUNCH(8);
    else PM_LAUNCH(PM_MAX_HPRIME);
#undef PM_LAUNCH
    return (int)hipGetLastError();
}

extern "C" int pm_tsc_select_scores_f64(const double *scores, int64_t lds, const double *gram, int64_t N, int64_t H,
                                        double *R, int64_t ldr, void *stream) {
    if (N == 0) return PM_OK;
    if (!scores || !gram || !R || N < 0 || H <= 0 || lds < H || ldr < 2 * H) return PM_EINVAL;
    if (H > INT32_MAX / 2) return PM_ERANGE;
    const int64_t blocks = (N * H + 255) / 256;
    hipLaunchKernelGGL(tsc_select_scores_kernel, dim3((unsigned)(blocks > 256 * 16 ? 256 * 16 : blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), scores, lds, gram, N, (int)H, R, ldr);
    return (int)hipGetLastError();
}

PM_DET_SETTER(dsc)
