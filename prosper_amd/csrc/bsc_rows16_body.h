// Binary Sparse Coding, 16 lanes per datapoint: the per-datapoint pass of select_Hprimes (bsc_et.py:98-115)
// and E_step (bsc_et.py:119-192), shared by
//   bsc_select_estep16_kernel (bsc_rows16.hip)  scores read back from HBM
//   bsc_estep_fused_kernel    (bsc_fused.hip)   scores taken from the MFMA accumulators of the scores GEMM
//
// A 64-lane wavefront is four DPP rows of 16 lanes; each row owns one datapoint, so every reduction over a
// datapoint's latents / states (top-H', max, sum) is a 4-step DPP butterfly inside the row.  Lane j of a row
// holds latents h = j + 16 i (i < VPL) -- which is also how v_mfma_f64_16x16x4_f64 leaves a 16-row block of
// scores in its accumulators (column = lane & 15, row = (lane >> 4) + 4 reg).
//
// Multi-cause state energies are built incrementally by state size (pairs, triples, ...):
//   e'(s) = e'(s minus its highest candidate k) + d_k + 2 sum_{i in s, i<k} G[c_i, c_k],
//   d_k = G[c_k,c_k] - 2 a_{c_k},   e(s) = |y|^2 + e'(s)
// with the parent's index precomputed on the host; a state costs |s|+1 LDS reads instead of
// |s|(|s|+3)/2.  Posterior terms below exp(-37) (< 1e-16 of the largest) are skipped wave-wide.
#ifndef PM_BSC_ROWS16_BODY_H
#define PM_BSC_ROWS16_BODY_H

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace pm_rows16 {

constexpr int ROWS = 16;          // datapoints per 256-thread workgroup (4 wavefronts x 4 DPP rows)
constexpr double NEGLIGIBLE = -37.0;

// ---- DPP helpers (all-reduce butterflies inside a 16-lane row) -------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned dpp32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v) {
    const unsigned lo = dpp32<CTRL>((unsigned)v), hi = dpp32<CTRL>((unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ double dppf64(double v) {
    return __longlong_as_double((long long)dpp64<CTRL>((uint64_t)__double_as_longlong(v)));
}
// xor 1, xor 2 (quad_perm), reverse within 8 (row_half_mirror), reverse within 16 (row_mirror)
#define PM_ROW_BUTTERFLY(OP, T, F)          \
    v = OP(v, F<0xB1>(v));                  \
    v = OP(v, F<0x4E>(v));                  \
    v = OP(v, F<0x141>(v));                 \
    v = OP(v, F<0x140>(v));
__device__ __forceinline__ double fadd(double a, double b) { return a + b; }
__device__ __forceinline__ double row_max_f64(double v) {
    PM_ROW_BUTTERFLY(fmax, double, dppf64)
    return v;
}
__device__ __forceinline__ double row_sum_f64(double v) {
    PM_ROW_BUTTERFLY(fadd, double, dppf64)
    return v;
}

__device__ __forceinline__ void wave_lds_sync16() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct SizeOffsets {  // multi-cause states of size g occupy [off[g-2], off[g-1]) ; g = 2..gamma
    int off[PM_MAX_HPRIME];
};

// kernel-uniform arguments of a pass
struct RowParams {
    const double *gram, *ynorm2, *wmu, *ymu;
    int S, gamma;
    pm_bsc_estep_params P;
    int64_t N;
    int H, Hp, mode;   // mode bit 0: select (compute + write candidates); else candidates are read from `cand`
                       //      bit 1: E-step (write logpj / lse)
                       //      bit 2: rank smallest first; bit 3: rank the scores as they are; bit 4: rank the
                       //             squared distance |W_h|^2 - 2 a_h (MMCA, mmca_et.py:119-120)
    int32_t *cand;
    double *logpj;
    int64_t ldl;
    double *lse;
};

// LDS areas: workgroup tables + this DPP row's datapoint scratch
struct RowLds {
    const double *w2;     // (H)  |W_h|^2 (+ 2 W_h.mu)
    const double *sw;     // (H)  1 / |W_h|
    const uint32_t *tab;  // (S)  state mask | parent << 16
    double *d;            // (16) d_k of the candidates
    double *G;            // (Hp*Hp) Gram block of the candidates
    double *e;            // (S)  multi-cause energies, then their log-joints
};

// Score of latent c (any c < 16 VPL) of this DPP row's datapoint, fetched from the lane that holds it:
// lane (c & 15) of the row, register c >> 4.  Every lane of the wavefront must call this (ds_bpermute).
template <int VPL>
__device__ __forceinline__ double row_lookup(const double (&a)[VPL], int lane, int c) {
    const int src = ((lane & 48) + (c & 15)) << 2;
    const int want = c >> 4;
    double out = 0.0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const long long b = __double_as_longlong(a[i]);
        const int lo = __builtin_amdgcn_ds_bpermute(src, (int)b);
        const int hi = __builtin_amdgcn_ds_bpermute(src, (int)(b >> 32));
        if (want == i) out = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    }
    return out;
}

// select_Hprimes for one datapoint per DPP row (bsc_et.py:98-115).  a[i] = score of latent h = j + 16 i
// (j = lane & 15); n = this row's datapoint (rows with n >= N shadow the last datapoint and write nothing).
// Returns, in lane j < Hp, the latent at candidate position j -- selected here (mode bit 0) or read from A.cand.
template <int VPL>
__device__ __forceinline__ int row_select(const double (&a)[VPL], const RowParams &A, const RowLds &L, int lane,
                                          int64_t n) {
    const int j = lane & 15;
    const int H = A.H, Hp = A.Hp;
    const bool live = n < A.N;               // uniform per DPP row
    const int64_t nn = live ? n : A.N - 1;
    int myc = 0;
    if (!(A.mode & 1)) {
        if (j < Hp) myc = A.cand[nn * Hp + j];
        return myc;
    }
    // ---------------- top-H' of a / |W_h| / |y| (ascending, best last) -------------------
    const double sy = 1.0 / sqrt(A.ynorm2[nn]);
    const bool smallest = A.mode & 4, raw = A.mode & 8, dist = A.mode & 16;
    // Ranking keys are DOUBLES whose low 10 mantissa bits carry the latent index (v_max_f64 is one
    // instruction, a 64-bit integer maximum three; the keys keep 42 mantissa bits either way).  Ties resolve
    // as a stable argsort would: largest-first keeps the larger index last-best, smallest-first the smaller
    // index first -- the index code counts up for non-negative keys and down for negative ones, whose
    // magnitude grows with the low bits.  NaN ranks below every number, +-inf are clamped to the largest
    // finite magnitudes (their low bits must stay free), -inf itself marks "taken".
    double key[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int h = j + 16 * i;
        double kx = -INFINITY;
        if (h < H) {
            double x = raw ? a[i] : dist ? L.w2[h] - 2.0 * a[i] : a[i] * L.sw[h] * sy;
            if (smallest) x = -x;
            uint64_t b = (uint64_t)__double_as_longlong(x);
            if (x != x) b = 0xFFEFFFFFFFFFFC00ull;
            else if ((b & 0x7FF0000000000000ull) == 0x7FF0000000000000ull)
                b = (b & 0x8000000000000000ull) | 0x7FEFFFFFFFFFF800ull;
            const uint64_t code = (uint64_t)(smallest ? 0x3FF - h : h);
            b = (b & ~0x3FFull) | ((b >> 63) ? 0x3FFull - code : code);
            kx = __longlong_as_double((long long)b);
        }
        key[i] = kx;
    }
    for (int r = 0; r < Hp; ++r) {
        double m = key[0];
#pragma unroll
        for (int i = 1; i < VPL; ++i) m = __builtin_fmax(m, key[i]);
        m = row_max_f64(m);
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (key[i] == m) key[i] = -INFINITY;
        const uint64_t mb = (uint64_t)__double_as_longlong(m);
        const int code = (int)(mb & 0x3FFull);
        const int win = (mb >> 63) ? 0x3FF - code : code;
        if (j == (smallest ? r : Hp - 1 - r)) myc = smallest ? 0x3FF - win : win;
    }
    if (live && j < Hp) A.cand[n * Hp + j] = myc;
    return myc;
}

// E_step for one datapoint per DPP row (bsc_et.py:119-192): log-joints of the null state, the H singletons and the
// multi-cause states over the candidates `myc` (lane j < Hp holds position j), and their log-sum-exp.
// `arow`: the datapoint's scores as an indexable row in global memory; with FROM_LANES it is ignored and the
// candidates' scores are fetched from the lanes that hold them.  a[] is overwritten (singleton log-joints).  `so` is taken by reference
// so that its dynamic indexing stays a scalar load from the kernel-argument segment (a copy would live in scratch).
template <int VPL, bool FROM_LANES>
__device__ __forceinline__ void row_estep(double (&a)[VPL], const double *arow, int myc, const RowParams &A,
                                          const SizeOffsets &so, const RowLds &L, int lane, int64_t n) {
    const int j = lane & 15;
    const int H = A.H, Hp = A.Hp, S = A.S;
    const bool live = n < A.N;               // uniform per DPP row
    const int64_t nn = live ? n : A.N - 1;
    const double ppil = A.P.prior_scale * A.P.pil_bar;
    double yn = A.ynorm2[nn];

    // ---------------- candidate block: d_k and G[c_i,c_k] -> LDS ------------------------
    if (A.ymu) yn = yn - 2.0 * A.ymu[nn] + A.P.mu_sqnorm;
    {
        const int c = (j < Hp) ? myc : 0;
        const double sc = FROM_LANES ? row_lookup<VPL>(a, lane, c) : arow[c];
        if (j < Hp) {
            const double ac = sc - (A.wmu ? A.wmu[c] : 0.0);
            L.d[j] = A.gram[(int64_t)c * H + c] - 2.0 * ac;
        }
    }
    for (int p0 = 0; p0 < Hp * Hp; p0 += 16) {            // uniform trip count: every lane feeds the bpermutes
        const int p = p0 + j;
        const bool valid = p < Hp * Hp;
        const int i = valid ? p / Hp : 0, k = valid ? p - i * Hp : 0;
        // candidates i and k of this datapoint, from the lanes of its DPP row that hold them
        const int ci = __builtin_amdgcn_ds_bpermute(((lane & 48) + i) << 2, myc);
        const int ck = __builtin_amdgcn_ds_bpermute(((lane & 48) + k) << 2, myc);
        if (valid) L.G[p] = A.gram[(int64_t)ci * H + ck];
    }
    wave_lds_sync16();

    // ---------------- multi-cause energies by size --------------------------------------
    for (int g = 2; g <= A.gamma; ++g) {
        for (int s = so.off[g - 2] + j; s < so.off[g - 1]; s += 16) {
            const uint32_t t = L.tab[s];
            const unsigned mask = t & 0xFFFFu;
            const int k = 31 - __builtin_clz(mask);  // highest candidate position of the state
            unsigned rest = mask & ~(1u << k);
            double e = L.d[k];
            if (g == 2) {
                const int i = __builtin_ctz(rest);
                e += L.d[i] + 2.0 * L.G[i * Hp + k];
            } else {
                e += L.e[t >> 16];
                double off = 0.0;
                while (rest) {
                    const int i = __builtin_ctz(rest);
                    rest &= rest - 1;
                    off += L.G[i * Hp + k];
                }
                e += 2.0 * off;
            }
            L.e[s] = e;
        }
        wave_lds_sync16();
    }

    // ---------------- log-pseudo-joints ---------------------------------------------------
    double *out = A.logpj + nn * A.ldl;
    double mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {  // a[i] becomes the singleton log-joint of latent h
        const int h = j + 16 * i;
        double f = -INFINITY;
        if (h < H) {
            const double e = L.w2[h] - 2.0 * a[i] + yn;
            f = ppil + A.P.ecoef * e;
            if (live) out[1 + h] = f;
        }
        a[i] = f;
        mx = fmax(mx, f);
    }
    const double f0 = A.P.ecoef * yn;
    if (j == 0) {
        if (live) out[0] = f0;
        mx = fmax(mx, f0);
    }
    for (int s = j; s < S; s += 16) {
        const unsigned mask = L.tab[s] & 0xFFFFu;
        const double f = ppil * (double)__builtin_popcount(mask) + A.P.ecoef * (yn + L.e[s]);
        if (live) out[1 + H + s] = f;
        L.e[s] = f;  // kept for the log-sum-exp pass (same lane re-reads it)
        mx = fmax(mx, f);
    }
    if (!A.lse) {
        wave_lds_sync16();
        return;
    }
    mx = row_max_f64(mx);
    double sum = (j == 0) ? exp(f0 - mx) : 0.0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const double dlt = a[i] - mx;
        const bool need = dlt > NEGLIGIBLE;
        if (__any(need)) sum += need ? exp(dlt) : 0.0;
    }
    for (int s0 = 0; s0 < S; s0 += 16) {  // uniform trip count
        const int s = s0 + j;
        const double dlt = (s < S) ? L.e[s] - mx : -INFINITY;
        const bool need = dlt > NEGLIGIBLE;
        if (__any(need)) sum += need ? exp(dlt) : 0.0;
    }
    sum = row_sum_f64(sum);
    if (live && j == 0) A.lse[n] = mx + log(sum);
    wave_lds_sync16();  // per-datapoint LDS areas are reused by the next pass
}

}  // namespace pm_rows16

#endif  // PM_BSC_ROWS16_BODY_H
