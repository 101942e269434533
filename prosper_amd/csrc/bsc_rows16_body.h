// Binary Sparse Coding, 16 lanes per datapoint: the per-datapoint pass of select_Hprimes (bsc_et.py:98-115)
// and E_step (bsc_et.py:119-192), shared by
//   bsc_select_estep16_kernel (bsc_rows16.hip)  scores read back from HBM
//   bsc_estep_fused_kernel    (bsc_fused.hip)   scores taken from the MFMA accumulators of the scores GEMM
//
// A 64-lane wavefront is four rows of 16 lanes; each row owns one datapoint, so every reduction over a datapoint's
// latents / states (top-H', max, sum) stays inside the row.  Lane j of a row holds latents h = j + 16 i (i < VPL) --
// which is also how v_mfma_f64_16x16x4_f64 leaves a 16-row block of scores in its accumulators (column = lane & 15,
// row = (lane >> 4) + 4 reg).
//
// What shaped this code (measured on MI355X, scratch/coissue*.hip): on a SIMD, VALU instructions and f64 MFMAs do not
// overlap -- v_mfma_f64_16x16x4_f64 holds the vector ALU for its 64 cycles, and every VALU instruction of either
// wavefront on that SIMD adds ~5.5 cycles on top.  LDS instructions (ds_read / ds_write / ds_swizzle / ds_bpermute),
// scalar instructions and memory latency cost the matrix pipe nothing.  Inside the fused kernel the row passes are
// therefore priced by their VALU instruction COUNT, and the work is moved off the vector ALU wherever possible:
//   * top-H': every lane sorts its keys once (Batcher network, v_max_f64 / v_min_f64) and parks the sorted list in
//     LDS; a round is then "read my head, row maximum through ds_swizzle, advance my pointer if I won" -- 7 VALU
//     instructions instead of the ~85 of a max-tree + knock-out sweep over 16 keys per lane;
//   * multi-cause state energies are table driven: e(s) = P[parent] + P[d_k] + 2 (P[g0] + P[g1] + P[g2]) with the five
//     LDS offsets of every state precomputed once per workgroup (states of up to 4 causes; larger ones walk their
//     mask) -- no per-state bit scans, no data-dependent loops;
//   * row reductions go through ds_swizzle (LDS crossbar) instead of DPP moves;
//   * exp / log are lean polynomial versions for the argument ranges that occur ([-37, 0] and [1, K]).
//
// Multi-cause state energies are built incrementally by state size (pairs, triples, ...):
//   e'(s) = e'(s minus its highest candidate k) + d_k + 2 sum_{i in s, i<k} G[c_i, c_k],
//   d_k = G[c_k,c_k] - 2 a_{c_k},   e(s) = |y|^2 + e'(s)
// with the parent's index precomputed on the host.  Posterior terms below exp(-37) (< 1e-16 of the largest) are
// skipped wave-wide.
#ifndef PM_BSC_ROWS16_BODY_H
#define PM_BSC_ROWS16_BODY_H

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace pm_rows16 {

constexpr int ROWS = 16;          // datapoints per 256-thread workgroup (4 wavefronts x 4 rows of 16 lanes)
constexpr double NEGLIGIBLE = -37.0;
constexpr int LIST_ROWS = 17;     // sorted keys parked per lane: up to 16 + one -inf sentinel

// v_max_f64 / v_min_f64 as single instructions: fmax()/fmin() first canonicalise operands the compiler cannot prove
// quiet (anything loaded or bit-cast), three instructions instead of one -- and every vector instruction here costs
// the matrix pipe ~5.5 cycles.  Operands are never signalling NaNs.
__device__ __forceinline__ double vmax64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmin64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- reductions inside a 16-lane row through the LDS crossbar (ds_swizzle, bit mode: lane ^ XOR) ------------------
template <int XOR>
__device__ __forceinline__ double swz_xor_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_swizzle((int)b, (XOR << 10) | 0x1F);
    const int hi = __builtin_amdgcn_ds_swizzle((int)(b >> 32), (XOR << 10) | 0x1F);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double row_max_f64(double v) {
    v = vmax64(v, swz_xor_f64<1>(v));
    v = vmax64(v, swz_xor_f64<2>(v));
    v = vmax64(v, swz_xor_f64<4>(v));
    v = vmax64(v, swz_xor_f64<8>(v));
    return v;
}
__device__ __forceinline__ double row_sum_f64(double v) {
    v += swz_xor_f64<1>(v);
    v += swz_xor_f64<2>(v);
    v += swz_xor_f64<4>(v);
    v += swz_xor_f64<8>(v);
    return v;
}

// LDS traffic of ONE wavefront is processed in issue order, so a ds_write is visible to a later ds_read of another
// lane of the same wavefront without waiting for anything: only the compiler has to keep the order (a workgroup-scope
// fence would also drain vmcnt, i.e. wait for every global store of the pass).
__device__ __forceinline__ void wave_lds_sync16() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// exp(x) for x in [-745, 0] (used on [-37, 0]): n = rint(x log2 e), r = x - n ln 2, degree-13 Taylor, ldexp.
__device__ __forceinline__ double exp_neg(double x) {
    x = vmax64(x, -745.0);     // lanes whose result is discarded may hold anything: keep the int conversion defined
    const double n = rint(x * 1.4426950408889634);
    double t = fma(-n, 6.9314718036912382e-01, x);
    t = fma(-n, 1.9082149292705877e-10, t);
    double q = 1.0 / 6227020800.0;
    q = fma(q, t, 1.0 / 479001600.0);
    q = fma(q, t, 1.0 / 39916800.0);
    q = fma(q, t, 1.0 / 3628800.0);
    q = fma(q, t, 1.0 / 362880.0);
    q = fma(q, t, 1.0 / 40320.0);
    q = fma(q, t, 1.0 / 5040.0);
    q = fma(q, t, 1.0 / 720.0);
    q = fma(q, t, 1.0 / 120.0);
    q = fma(q, t, 1.0 / 24.0);
    q = fma(q, t, 1.0 / 6.0);
    q = fma(q, t, 0.5);
    q = fma(q, t, 1.0);
    q = fma(q, t, 1.0);
    return ldexp(q, (int)n);
}

// log(x) for finite x >= 1 (used on [1, number of states]): x = m 2^e, m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1),
// log m = 2 s sum_k s^(2k)/(2k+1) (k <= 10).  Relative error ~2e-16.
__device__ __forceinline__ double log_ge1(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double s = num * r;
    s = fma(fma(-den, s, num), r, s);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = p * z;                                            // log m = 2s (1 + p)
    const double ed = (double)e;
    const double two_s = s + s;
    const double lg_lo = fma(two_s, p, ed * 1.9082149292705877e-10);
    return fma(ed, 6.9314718036912382e-01, two_s + lg_lo);
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
    // M-step statistics produced in the same pass (fused kernel, no data truncation ahead): E[s] rows and the upper
    // triangle of sum_n E[s s^T] over the multi-cause states (bsc_et.py:334-366); null when not wanted
    double *expect;
    int64_t lde;
    double *wq;
    int cand_mod;      // != 0: candidates are stored modulo this (TSC: one-cause state 0 .. 2H-1 -> its latent, tsc_et.py:210)
};

// per-lane partial sums of the scalar M-step statistics over the passes of a wavefront
struct MAcc {
    double sig, fs, cnt;   // sum_nk q e (bsc_et.py:395-415), sum_n log-evidence, datapoints
};

// ---- LDS layout of a workgroup ------------------------------------------------------------------------------------
//   [ w2 (HT) | sw (HT) | ew (HT) | mus (HT) | 16 datapoint areas of `area` doubles | tab (S u32) | st (S x 8 u16) |
//     ik (Hp*Hp u16) ]
// HT = latents rounded up to 16.  A datapoint area:
//   P    = [ zero | d (16) | G (Hp*Hp) | e (S) ]   byte-addressed by the state table; during selection the same bytes
//                                                  hold the lanes' sorted key lists (LIST_ROWS x 16 doubles)
//   win  = (16)   the winners of the selection rounds
//   row  = (HT)   the datapoint's scores as an indexable row (fused kernel only: rowbuf = 1)
struct Layout {
    int HT, area, p_len;          // doubles
    int off_dp, off_tab, off_st, off_ik;   // bytes from the start of the workgroup's LDS
    int bytes;
};
__host__ __device__ inline Layout make_layout(int H, int Hp, int S, int rowbuf) {
    Layout L;
    L.HT = (H + 15) / 16 * 16;
    const int p = 1 + 16 + Hp * Hp + S, lists = LIST_ROWS * 16;
    L.p_len = p > lists ? p : lists;
    L.area = L.p_len + 16 + (rowbuf ? L.HT : 0);
    L.off_dp = 4 * L.HT * 8;
    L.off_tab = L.off_dp + ROWS * L.area * 8;
    L.off_st = (L.off_tab + 4 * S + 15) / 16 * 16;
    L.off_ik = L.off_st + 16 * S;
    L.bytes = (L.off_ik + 2 * Hp * Hp + 15) / 16 * 16;
    return L;
}

// workgroup tables + this row-of-16-lanes' datapoint area
struct RowLds {
    const double *w2;     // (HT) |W_h|^2 (+ 2 W_h.mu)
    const double *sw;     // (HT) 1 / |W_h|
    const double *ew;     // (HT) ecoef |W_h|^2 (+ 2 ecoef W_h.mu) + prior: the singleton log-joint up to -2 ecoef a_h + ecoef |y|^2
    double *mus;          // (HT) sum of E[s_h] over the workgroup's datapoints (M-statistics mode)
    const uint32_t *tab;  // (S)  state mask | parent << 16
    const uint16_t *st;   // (S x 8) byte offsets into P: parent term, d_k, three Gram terms (unused ones -> zero slot)
    const uint16_t *ik;   // (Hp*Hp) i | k << 8 of the Gram block's entry p = i Hp + k
    double *P;            // [ zero | d | G | e ] / sorted key lists
    double *win;          // (16)
    double *row;          // (HT), fused kernel only
};

__device__ __forceinline__ RowLds row_lds(unsigned char *smem, const Layout &L, int dp /* 0..15 */) {
    double *base = reinterpret_cast<double *>(smem);
    double *area = reinterpret_cast<double *>(smem + L.off_dp) + dp * L.area;
    return RowLds{base, base + L.HT, base + 2 * L.HT, base + 3 * L.HT, reinterpret_cast<const uint32_t *>(smem + L.off_tab),
                  reinterpret_cast<const uint16_t *>(smem + L.off_st), reinterpret_cast<const uint16_t *>(smem + L.off_ik),
                  area, area + L.p_len, area + L.p_len + 16};
}

// Fill the workgroup tables (all 256 threads; the caller synchronises afterwards).  g_h = G[h,h] and wmu_h = (W.mu)_h
// of latent h = min(tid, H-1) and tab_s = mask | parent << 16 of state s = min(tid, S-1) are handed in (loaded by
// the caller, possibly long before); latents and states beyond 256 are read here.
__device__ __forceinline__ void build_tables(unsigned char *smem, const Layout &L, int tid, double g_h, double wmu_h,
                                             uint32_t tab_s, double ecoef, double ppil,
                                             const double *__restrict__ gram,
                                             const double *__restrict__ wmu, int H,
                                             const uint16_t *__restrict__ masks, const uint16_t *__restrict__ parents,
                                             int S, int Hp) {
    double *w2 = reinterpret_cast<double *>(smem);
    double *sw = w2 + L.HT, *ew = w2 + 2 * L.HT;
    if (tid < L.HT) {
        const double w = g_h + 2.0 * wmu_h;
        w2[tid] = w;
        sw[tid] = 1.0 / sqrt(g_h);   // ranking uses a * (1/|W_h|): keys keep 42 mantissa bits anyway
        ew[tid] = fma(ecoef, w, ppil);
        w2[3 * L.HT + tid] = 0.0;    // mus
    }
    for (int h = 256 + tid; h < L.HT; h += 256) {
        const int hc = h < H ? h : H - 1;
        const double g = gram[(int64_t)hc * H + hc];
        const double w = g + (wmu ? 2.0 * wmu[hc] : 0.0);
        w2[h] = w;
        sw[h] = 1.0 / sqrt(g);
        ew[h] = fma(ecoef, w, ppil);
        w2[3 * L.HT + h] = 0.0;
    }
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem + L.off_tab);
    uint16_t *st = reinterpret_cast<uint16_t *>(smem + L.off_st);
    const int o_d = 8, o_G = 8 * 17, o_e = 8 * (17 + Hp * Hp);
    for (int s = tid; s < S; s += 256) {
        const uint32_t t = (s < 256) ? tab_s : ((uint32_t)masks[s] | ((uint32_t)parents[s] << 16));
        const unsigned mask = t & 0xFFFFu, par = t >> 16;
        tab[s] = t;
        const int k = 31 - __builtin_clz(mask);
        unsigned rest = mask & ~(1u << k);
        const int g = __builtin_popcount(mask);
        unsigned e0 = 0, e2 = 0, e3 = 0, e4 = 0;
        const unsigned e1 = o_d + 8 * k;
        if (g <= 4) {
            const int b0 = __builtin_ctz(rest);
            e0 = (g == 2) ? (unsigned)(o_d + 8 * b0) : (unsigned)(o_e + 8 * par);
            e2 = o_G + 8 * (b0 * Hp + k);
            rest &= rest - 1;
            if (rest) {
                e3 = o_G + 8 * (__builtin_ctz(rest) * Hp + k);
                rest &= rest - 1;
            }
            if (rest) e4 = o_G + 8 * (__builtin_ctz(rest) * Hp + k);
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(st + s * 8);
        dst[0] = e0 | (e1 << 16);
        dst[1] = e2 | (e3 << 16);
        dst[2] = e4;
        dst[3] = 0;
    }
    uint16_t *ik = reinterpret_cast<uint16_t *>(smem + L.off_ik);
    for (int p = tid; p < Hp * Hp; p += 256) {
        const int i = p / Hp;
        ik[p] = (uint16_t)(i | ((p - i * Hp) << 8));
    }
}

// Batcher's odd-even merge sort on registers, descending: k[0] >= k[1] >= ...  (n a power of two; all indices are
// compile-time constants after unrolling).
template <int n>
__device__ __forceinline__ void sort_desc(double (&k)[n]) {
#pragma unroll
    for (int p = 1; p < n; p <<= 1) {
#pragma unroll
        for (int q = p; q >= 1; q >>= 1) {
#pragma unroll
            for (int j = q % p; j + q < n; j += 2 * q) {
#pragma unroll
                for (int i = 0; i < q; ++i) {
                    if (i + j + q < n && (i + j) / (2 * p) == (i + j + q) / (2 * p)) {
                        const double hi = vmax64(k[i + j], k[i + j + q]);
                        const double lo = vmin64(k[i + j], k[i + j + q]);
                        k[i + j] = hi;
                        k[i + j + q] = lo;
                    }
                }
            }
        }
    }
}

// select_Hprimes for one datapoint per row of 16 lanes (bsc_et.py:98-115).  a[i] = score of latent h = j + 16 i
// (j = lane & 15); n = this row's datapoint (rows with n >= N shadow the last datapoint and write nothing).
// Returns, in lane j < Hp, the latent at candidate position j -- selected here (mode bit 0) or read from A.cand.
// SEL >= 0: the ranking mode (A.mode >> 2) is a compile-time constant (0 = Binary Sparse Coding's own); SEL < 0: read
// from A.mode.  FULL: H == 16 VPL, every lane register holds a latent.
template <int VPL, int SEL = -1, bool FULL = false>
__device__ __forceinline__ int row_select(const double (&a)[VPL], const RowParams &A, const RowLds &L, int lane,
                                          int64_t n) {
    const int j = lane & 15;
    const int H = A.H, Hp = A.Hp;
    const bool live = n < A.N;               // uniform per row
    const int64_t nn = live ? n : A.N - 1;
    int myc = 0;
    if (!(A.mode & 1)) {
        if (j < Hp) myc = A.cand[nn * Hp + j];
        return myc;
    }
    // ---------------- top-H' of a / |W_h| (ascending, best last) ------------------------------------------
    // (the reference divides by |y| too -- a positive factor per datapoint, the ranking is the same)
    const int flags = SEL >= 0 ? SEL : (A.mode >> 2);
    const bool smallest = flags & 1, raw = flags & 2, dist = flags & 4;
    // Ranking keys are DOUBLES whose low 10 mantissa bits carry the latent index (the keys keep 42 mantissa bits).
    // Ties resolve as a stable argsort would: largest-first keeps the larger index last-best, smallest-first the
    // smaller index first -- the index code counts up for non-negative keys and down for negative ones, whose
    // magnitude grows with the low bits.  NaN ranks below every number, +-inf are clamped to the largest finite
    // magnitudes (their low bits must stay free), -inf itself marks "no latent".
    double key[VPL];
    bool odd = false;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int h = j + 16 * i;
        const int hc = (FULL || h < H) ? h : H - 1;
        double x = raw ? a[i] : dist ? L.w2[hc] - 2.0 * a[i] : a[i] * L.sw[hc];
        if (smallest) x = -x;
        odd |= __builtin_amdgcn_class(x, 0x207);         // NaN, -inf, +inf
        const uint64_t b = (uint64_t)__double_as_longlong(x);
        const uint32_t flip = (uint32_t)((int32_t)(b >> 32) >> 31) & 0x3FFu;
        const uint32_t code = (uint32_t)(smallest ? 0x3FF - h : h) ^ flip;
        const uint64_t kb = (b & ~0x3FFull) | code;
        key[i] = (FULL || h < H) ? __longlong_as_double((long long)kb) : -INFINITY;
    }
    if (__any(odd)) {   // rare: redo the keys of non-finite scores
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            const int hc = h < H ? h : H - 1;
            double x = raw ? a[i] : dist ? L.w2[hc] - 2.0 * a[i] : a[i] * L.sw[hc];
            if (smallest) x = -x;
            uint64_t b = (uint64_t)__double_as_longlong(x);
            if (x != x) b = 0xFFEFFFFFFFFFFC00ull;
            else if ((b & 0x7FF0000000000000ull) == 0x7FF0000000000000ull)
                b = (b & 0x8000000000000000ull) | 0x7FEFFFFFFFFFF800ull;
            const uint64_t code = (uint64_t)(smallest ? 0x3FF - h : h);
            b = (b & ~0x3FFull) | ((b >> 63) ? 0x3FFull - code : code);
            if (h < H) key[i] = __longlong_as_double((long long)b);
        }
    }
    // every lane sorts its keys (descending) and parks the list in LDS, list entry t of lane j at P[t * 16 + j]
    sort_desc<VPL>(key);
    constexpr int LR = VPL < 16 ? VPL : 16;
    double *list = L.P + j;
#pragma unroll
    for (int t = 0; t < LR; ++t) list[t * 16] = key[t];
    list[LR * 16] = -INFINITY;
    wave_lds_sync16();
    // a round: my best remaining key against the row's; the winner advances to its next key
    int ptr = 0;
    for (int r = 0; r < Hp; ++r) {
        const double head = list[ptr * 16];
        const double m = row_max_f64(head);
        L.win[r] = m;                        // the same value from all 16 lanes
        ptr += (head == m) ? 1 : 0;
    }
    wave_lds_sync16();
    if (j < Hp) {
        const uint64_t mb = (uint64_t)__double_as_longlong(L.win[smallest ? j : Hp - 1 - j]);
        const int code = (int)(mb & 0x3FFull);
        const int win = (mb >> 63) ? 0x3FF - code : code;
        myc = smallest ? 0x3FF - win : win;
        if (A.cand_mod) myc = myc % A.cand_mod;
        if (live) A.cand[n * Hp + j] = myc;
    }
    wave_lds_sync16();   // the lists' bytes become P
    return myc;
}

// E_step for one datapoint per row of 16 lanes (bsc_et.py:119-192): log-joints of the null state, the H singletons and
// the multi-cause states over the candidates `myc` (lane j < Hp holds position j), and their log-sum-exp.
// `arow`: the datapoint's scores as an indexable row (global memory, or L.row in LDS).  a[] is overwritten (singleton
// log-joints).  `so` is taken by reference so that its dynamic indexing stays a scalar load from the kernel-argument
// segment (a copy would live in scratch).
// What a pass fetches from global memory once its candidates are known: |y|^2, G[c,c] and the score offset (W.mu)_c of the
// candidate this lane holds, and this lane's share of the candidates' Gram block (entries p = 16 it + j of Hp x Hp).
// Split from the arithmetic so that a caller with several passes in hand can have all of their loads in flight at once.
template <int GI>
struct RowFetch {
    double yn, gdiag, wmuc;
    double G[GI];
};
template <int GI>
__device__ __forceinline__ RowFetch<GI> row_fetch(int myc, const RowParams &A, const RowLds &L, int lane, int64_t n) {
    const int j = lane & 15;
    const int H = A.H, Hp = A.Hp;
    const int64_t nn = n < A.N ? n : A.N - 1;
    RowFetch<GI> F;
    F.yn = A.ynorm2[nn];
    if (A.ymu) F.yn = F.yn - 2.0 * A.ymu[nn] + A.P.mu_sqnorm;
    const int c = (j < Hp) ? myc : 0;
    F.gdiag = A.gram[(int64_t)c * H + c];
    F.wmuc = A.wmu ? A.wmu[c] : 0.0;
#pragma unroll
    for (int it = 0; it < GI; ++it) {                   // every lane feeds the bpermutes
        const int p = 16 * it + j;
        const bool valid = p < Hp * Hp;
        const unsigned ik = valid ? L.ik[p] : 0u;
        // candidates i and k of this datapoint, from the lanes of its row that hold them
        const int ci = __builtin_amdgcn_ds_bpermute(((lane & 48) + (int)(ik & 0xFF)) << 2, myc);
        const int ck = __builtin_amdgcn_ds_bpermute(((lane & 48) + (int)(ik >> 8)) << 2, myc);
        F.G[it] = (16 * it < Hp * Hp) ? A.gram[(int64_t)ci * H + ck] : 0.0;
    }
    return F;
}

template <int VPL, bool FULL, bool MSTATS>
__device__ __forceinline__ void row_estep_compute(double (&a)[VPL], double yn, int myc, const RowParams &A,
                                                  const SizeOffsets &so, const RowLds &L, int lane, int64_t n, MAcc *acc);

// E_step with fetched values: ac = the score of this lane's candidate (lanes j < Hp).  GI * 16 >= Hp * Hp.
template <int VPL, bool FULL, int GI, bool MSTATS = false>
__device__ __forceinline__ void row_estep_fetched(double (&a)[VPL], double ac, int myc, const RowFetch<GI> &F,
                                                  const RowParams &A, const SizeOffsets &so, const RowLds &L, int lane,
                                                  int64_t n, MAcc *acc = nullptr) {
    const int j = lane & 15;
    const int Hp = A.Hp;
    double *Pd = L.P + 1, *PG = L.P + 17;
    if (j == 0) L.P[0] = 0.0;
    if (j < Hp) Pd[j] = F.gdiag - 2.0 * (ac - F.wmuc);
#pragma unroll
    for (int it = 0; it < GI; ++it) {
        const int p = 16 * it + j;
        if (p < Hp * Hp) PG[p] = F.G[it];
    }
    wave_lds_sync16();
    row_estep_compute<VPL, FULL, MSTATS>(a, F.yn, myc, A, so, L, lane, n, acc);
}

template <int VPL, bool FULL = false>
__device__ __forceinline__ void row_estep(double (&a)[VPL], const double *arow, int myc, const RowParams &A,
                                          const SizeOffsets &so, const RowLds &L, int lane, int64_t n) {
    const int j = lane & 15;
    const int H = A.H, Hp = A.Hp;
    const bool live = n < A.N;               // uniform per row
    const int64_t nn = live ? n : A.N - 1;
    double yn = A.ynorm2[nn];
    if (A.ymu) yn = yn - 2.0 * A.ymu[nn] + A.P.mu_sqnorm;

    // ---------------- candidate block: d_k and G[c_i,c_k] -> LDS ------------------------
    double *Pd = L.P + 1, *PG = L.P + 17;
    if (j == 0) L.P[0] = 0.0;
    if (j < Hp) {
        const int c = myc;
        const double ac = arow[c] - (A.wmu ? A.wmu[c] : 0.0);
        Pd[j] = A.gram[(int64_t)c * H + c] - 2.0 * ac;
    }
    for (int p0 = 0; p0 < Hp * Hp; p0 += 16) {            // uniform trip count: every lane feeds the bpermutes
        const int p = p0 + j;
        const bool valid = p < Hp * Hp;
        const unsigned ik = valid ? L.ik[p] : 0u;
        // candidates i and k of this datapoint, from the lanes of its row that hold them
        const int ci = __builtin_amdgcn_ds_bpermute(((lane & 48) + (int)(ik & 0xFF)) << 2, myc);
        const int ck = __builtin_amdgcn_ds_bpermute(((lane & 48) + (int)(ik >> 8)) << 2, myc);
        if (valid) PG[p] = A.gram[(int64_t)ci * H + ck];
    }
    wave_lds_sync16();
    row_estep_compute<VPL, FULL, false>(a, yn, myc, A, so, L, lane, n, nullptr);
}

// The arithmetic of the E-step: P = [ zero | d | G ] of this datapoint is in LDS.  MSTATS: also the per-datapoint part
// of the M-step (bsc_et.py:271-272, 334-366, 395-415) for a datapoint that is certainly kept -- posterior weights from
// the exponentials the log-sum-exp has just evaluated, E[s] row (through L.row), second-moment block of the candidates
// scattered into A.wq, column sums into L.mus, scalar sums into *acc.  Needs L.row.
template <int VPL, bool FULL, bool MSTATS>
__device__ __forceinline__ void row_estep_compute(double (&a)[VPL], double yn, int myc, const RowParams &A,
                                                  const SizeOffsets &so, const RowLds &L, int lane, int64_t n, MAcc *acc) {
    const int j = lane & 15;
    const int H = A.H, Hp = A.Hp;
    const bool live = n < A.N;               // uniform per row
    const int64_t nn = live ? n : A.N - 1;
    const double ppil = A.P.prior_scale * A.P.pil_bar, ecoef = A.P.ecoef;
    double *Pd = L.P + 1, *PG = L.P + 17, *Pe = L.P + 17 + Hp * Hp;
    const unsigned char *Pb = reinterpret_cast<const unsigned char *>(L.P);

    // ---------------- multi-cause energies and log-joints, by state size ------------------------------
    double *out = A.logpj + nn * A.ldl;
    double mx = -INFINITY;
    for (int g = 2; g <= A.gamma; ++g) {
        const double pg = ppil * (double)g;
        const int s1 = so.off[g - 1];
        if (g <= 4) {
            for (int s = so.off[g - 2] + j; s < s1; s += 16) {
                const uint16_t *t = L.st + s * 8;
                const double e = (*reinterpret_cast<const double *>(Pb + t[0]) + *reinterpret_cast<const double *>(Pb + t[1])) +
                                 2.0 * ((*reinterpret_cast<const double *>(Pb + t[2]) +
                                         *reinterpret_cast<const double *>(Pb + t[3])) +
                                        *reinterpret_cast<const double *>(Pb + t[4]));
                Pe[s] = e;
                const double f = fma(ecoef, yn + e, pg);
                if (live) out[1 + H + s] = f;
                mx = vmax64(mx, f);
            }
        } else {
            for (int s = so.off[g - 2] + j; s < s1; s += 16) {
                const uint32_t t = L.tab[s];
                const unsigned mask = t & 0xFFFFu;
                const int k = 31 - __builtin_clz(mask);  // highest candidate position of the state
                unsigned rest = mask & ~(1u << k);
                double off = 0.0;
                while (rest) {
                    const int i = __builtin_ctz(rest);
                    rest &= rest - 1;
                    off += PG[i * Hp + k];
                }
                const double e = (Pe[t >> 16] + Pd[k]) + 2.0 * off;
                Pe[s] = e;
                const double f = fma(ecoef, yn + e, pg);
                if (live) out[1 + H + s] = f;
                mx = vmax64(mx, f);
            }
        }
        wave_lds_sync16();
    }

    // ---------------- null state and singletons ---------------------------------------------------
    // f_h = prior + ecoef (|W_h|^2 - 2 a_h + |y|^2), with ecoef |W_h|^2 + prior tabulated per workgroup
    const double f0 = ecoef * yn, m2e = -2.0 * ecoef;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {  // a[i] becomes the singleton log-joint of latent h
        const int h = j + 16 * i;
        double f = -1.0e300;     // a register without a latent: never the maximum, never counted (finite: 0 x f stays 0)
        if (FULL || h < H) {
            f = fma(m2e, a[i], L.ew[h]) + f0;
            if (live) out[1 + h] = f;
        }
        a[i] = f;
        mx = vmax64(mx, f);
    }
    if (j == 0) {
        if (live) out[0] = f0;
        mx = vmax64(mx, f0);
    }
    if (!A.lse) return;

    // ---------------- log-sum-exp: only terms within exp(-37) of the largest are evaluated ------------
    mx = row_max_f64(mx);
    const double thr = mx + NEGLIGIBLE;
    double sum = 0.0, qe = 0.0;          // qe: sum of exp(f - mx) (f - prior) = ecoef sum of exp(.) e
    {
        const bool need = (j == 0) && f0 > thr;
        if (__any(need)) {
            sum = need ? exp_neg(f0 - mx) : 0.0;
            if (MSTATS) qe = sum * f0;
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {      // MSTATS: a[i] becomes exp(f - mx) of a term that counts, else 0
        const bool need = a[i] > thr;
        double ex = 0.0;
        if (__any(need)) {
            ex = need ? exp_neg(a[i] - mx) : 0.0;
            sum += ex;
            if (MSTATS) qe = need ? fma(ex, a[i] - ppil, qe) : qe;     // (a padded latent holds -inf: 0 x -inf)
        }
        if (MSTATS) a[i] = ex;
    }
    // multi-cause states: the test runs on the energies (f > thr  <=>  e < te_g for ecoef < 0, one compare per state);
    // the log-joint is rebuilt -- with the arithmetic that produced the stored value -- only where an exp is due
    const double inv_ecoef = 1.0 / ecoef;
    double *m2 = PG;                     // MSTATS: the Gram block's bytes become the candidates' second-moment block
    if (MSTATS) {
        for (int p = j; p < Hp * Hp; p += 16) m2[p] = 0.0;
        wave_lds_sync16();
    }
    for (int g = 2; g <= A.gamma; ++g) {
        const double pg = ppil * (double)g;
        const double te = (thr - pg) * inv_ecoef - yn;
        const int s0 = so.off[g - 2], s1 = so.off[g - 1];
        for (int sb = s0; sb < s1; sb += 16) {  // uniform trip count
            const int s = sb + j;
            const double e = Pe[s < s1 ? s : s1 - 1];
            const bool need = (s < s1) && (ecoef < 0.0 ? e < te : e > te);
            if (__any(need)) {
                const double f = fma(ecoef, yn + e, pg);
                const double ex = (need && f > thr) ? exp_neg(f - mx) : 0.0;
                sum += ex;
                if (MSTATS && ex != 0.0) {
                    qe = fma(ex, f - pg, qe);
                    unsigned mi = L.tab[s] & 0xFFFFu;
                    while (mi) {             // E[s_i s_k] += weight for every pair i <= k of the state (LDS atomics)
                        const int i = __builtin_ctz(mi);
                        mi &= mi - 1;
                        atomicAdd(&m2[i * Hp + i], ex);
                        unsigned mk = mi;
                        while (mk) {
                            const int k = __builtin_ctz(mk);
                            mk &= mk - 1;
                            atomicAdd(&m2[i * Hp + k], ex);
                        }
                    }
                }
            }
        }
    }
    sum = row_sum_f64(sum);
    const double lse_n = mx + log_ge1(sum);
    if (live && j == 0) A.lse[n] = lse_n;
    if (MSTATS) {
        const double inv = 1.0 / sum;
        if (live) {
            acc->sig += qe * inv * inv_ecoef;
            if (j == 0) {
                acc->fs += lse_n;
                acc->cnt += 1.0;
            }
        }
        // E[s] row: singleton weights through the LDS row, the candidates' multi-cause weights added there
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (FULL || j + 16 * i < H) L.row[j + 16 * i] = a[i] * inv;
        wave_lds_sync16();
        if (j < Hp) {
            const double m1 = m2[j * Hp + j] * inv;
            if (m1 != 0.0) L.row[myc] += m1;
        }
        // the candidates' second-moment block -> upper triangle of Wq
        for (int p0 = 0; p0 < Hp * Hp; p0 += 16) {          // uniform trip count: every lane feeds the bpermutes
            const int p = p0 + j;
            const bool valid = p < Hp * Hp;
            const unsigned ik = valid ? L.ik[p] : 0u;
            const int ci = __builtin_amdgcn_ds_bpermute(((lane & 48) + (int)(ik & 0xFF)) << 2, myc);
            const int ck = __builtin_amdgcn_ds_bpermute(((lane & 48) + (int)(ik >> 8)) << 2, myc);
            const double v = valid ? m2[p] * inv : 0.0;
            if (live && v != 0.0) {
                const int lo = ci < ck ? ci : ck, hi = ci < ck ? ck : ci;
                pm_atomic_add(A.wq + (int64_t)lo * H + hi, PM_Q(v, 0));
            }
        }
        wave_lds_sync16();
        double *erow = A.expect + nn * A.lde;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            if (FULL || h < H) {
                const double v = L.row[h];
                if (live) {
                    erow[h] = v;
                    if (__any(v != 0.0)) {
                        if (v != 0.0) atomicAdd(&L.mus[h], PM_Q(v, 0));
                    }
                }
            }
        }
    }
    wave_lds_sync16();  // the datapoint area is reused by the next pass
}

}  // namespace pm_rows16

#endif  // PM_BSC_ROWS16_BODY_H
