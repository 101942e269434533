// Device helpers shared by the gfx950 kernels: 64-lane wavefront reductions and f64 atomics.
#ifndef PM_COMMON_H
#define PM_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PM_WAVE 64

// global_atomic_add_f64 (no return): agent scope, relaxed.  Built with -munsafe-fp-atomics so
// this lowers to the hardware instruction, not a compare-and-swap loop.
__device__ __forceinline__ void pm_atomic_add(double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- deterministic build (-DPM_DETERMINISTIC -> libprosper_hip_det.so; `model.deterministic = True`) --------------------------
// The statistics of an M-step are sums over datapoints formed with f64 atomics (LDS and global): the order in which the
// addends arrive changes from run to run and with it the rounding of every partial sum -- two identical EM loops drift apart
// in the last bits (and, where a parameter is an element-wise ratio of two such sums, visibly: MCA).  Rounding each addend, before
// it is added, to a multiple of q = 2^-52 M (M >= any partial sum of its accumulator, a power of two the host derives from
// rigorous bounds on the data and parameters) makes every addition EXACT -- multiples of q below 2^53 q are closed under f64
// addition -- so the result no longer depends on the order: same bits every run, on any schedule.  Price: two adds per atomic and
// an error of at most half a quantum per addend (n 2^-53 M in the worst case, ~sqrt(n) 2^-53 M typically -- the size of the
// rounding error the plain sum makes on its LARGEST entries, spread over all of them).  The default build compiles PM_Q to
// nothing: its kernels are bit-for-bit the ones that were tuned.
//   PM_Q(v, c)   v rounded to category c's quantum (pm_det_M[c] = 1.5 * 2^e: the add / subtract trick)
//   each translation unit with accumulators has its own pm_det_M and a setter the host calls before its launches (from host or
//   device memory -- pm_gsc_det_quanta_f64 leaves quanta on the device)
#ifdef PM_DETERMINISTIC
// (__constant__: loads from the constant address space are invariant for the compiler -- it keeps a quantum in a scalar register
// across the atomics of a loop instead of re-reading a global that, for all it knows, the atomic has just changed: the sparse
// product ran 20 % slower in this build for that alone.  Written between launches only: by the setter, or by
// gsc_det_quanta_kernel through the address prosper_det_addr_<unit> returns.)
static __constant__ double pm_det_M[8];
#define PM_Q(v, c) (((v) + pm_det_M[c]) - pm_det_M[c])
#define PM_DET_SETTER(unit)                                                                                      \
    extern "C" int prosper_det_set_##unit(const double *M8, void *stream) {                                     \
        void *sym = nullptr;                                                                                     \
        if (hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(pm_det_M)); e != hipSuccess) return (int)e;      \
        return (int)hipMemcpyAsync(sym, M8, 8 * sizeof(double), hipMemcpyDefault, static_cast<hipStream_t>(stream)); \
    }                                                                                                            \
    extern "C" double *prosper_det_addr_##unit(void) {     /* the unit's eight quanta, for a kernel that derives them */ \
        void *sym = nullptr;                                                                                     \
        return hipGetSymbolAddress(&sym, HIP_SYMBOL(pm_det_M)) == hipSuccess ? static_cast<double *>(sym) : nullptr; \
    }
#else
#define PM_Q(v, c) (v)
#define PM_DET_SETTER(unit) \
    extern "C" int prosper_det_set_##unit(const double *, void *) { return -2; } \
    extern "C" double *prosper_det_addr_##unit(void) { return nullptr; }
#endif

// MI355X has 8 XCDs with one L2 each.  f64 atomics from all of them on the same few thousand lines bounce those
// lines between the L2s; hot (H,H) / (H,D) accumulators are therefore kept once per XCD -- copy 0 in the
// documented slot, copies 1..7 in a scratch tail of the statistics buffer -- and folded by pm_fold_copies_kernel.
#define PM_XCD_COPIES 8
__device__ __forceinline__ int pm_xcc_id() {
    return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & (PM_XCD_COPIES - 1));   // HW_REG_XCC_ID[3:0]
}
// this XCD's copy of an accumulator region of `len` doubles: `slot` for XCD 0, else tail + (xcc - 1) * len
__device__ __forceinline__ double *pm_xcd_copy(double *slot, double *tail, int64_t len) {
    const int x = pm_xcc_id();
    return x == 0 ? slot : tail + (int64_t)(x - 1) * len;
}
// slot[0 .. len) += the seven tail copies, which are cleared (a later call accumulating into the same buffer
// starts clean)
static __global__ __launch_bounds__(256) void pm_fold_copies_kernel(double *__restrict__ slot,
                                                                     double *__restrict__ tail, int64_t len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    double a = slot[i];
#pragma unroll
    for (int c = 0; c < PM_XCD_COPIES - 1; ++c) {
        a += tail[(int64_t)c * len + i];
        tail[(int64_t)c * len + i] = 0.0;
    }
    slot[i] = a;
}

__device__ __forceinline__ double pm_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, PM_WAVE);
    return v;
}

__device__ __forceinline__ double pm_wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, PM_WAVE));
    return v;
}

// Wave-wide sum without the LDS crossbar: DPP butterflies inside each 16-lane row (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror), then the four row totals through v_readlane.  Result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double pm_dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double pm_readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double pm_wave_sum_dpp(double v) {
    v += pm_dpp_f64<0xB1>(v);
    v += pm_dpp_f64<0x4E>(v);
    v += pm_dpp_f64<0x141>(v);
    v += pm_dpp_f64<0x140>(v);
    return (pm_readlane_f64(v, 0) + pm_readlane_f64(v, 16)) + (pm_readlane_f64(v, 32) + pm_readlane_f64(v, 48));
}

// Wave-wide argmax of (value, index): larger value wins, ties go to the larger index.
__device__ __forceinline__ void pm_wave_argmax(double &v, int &idx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off, PM_WAVE);
        const int oi = __shfl_xor(idx, off, PM_WAVE);
        const bool take = (ov > v) || (ov == v && oi > idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    }
}

// x^c for finite x >= 0 and modest |c log x| (MCA: Wbar = T^(1/rho), mca_et.py:170): exp(c log x) without
// libm's double-double log and special-case handling (144 VALU instructions there, ~50 here).
//   log:  x = m 2^e, m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1), log m = 2 s sum_k s^(2k)/(2k+1) (k <= 10)
//   exp:  y = n ln2 + r, |r| <= ln2/2, degree-13 Taylor polynomial, ldexp
// Relative error <= ~3e-16 for results within [1e-100, 1e100]; x == 0 gives 0 (c > 0).
__device__ __forceinline__ double pm_pow_pos(double x, double c) {
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
    // log x = e ln2_hi + (2s + (2s p + e ln2_lo))
    const double lg_lo = fma(two_s, p, ed * 1.9082149292705877e-10);
    const double lg = fma(ed, 6.9314718036912382e-01, two_s + lg_lo);
    const double y = c * lg;
    const double n = rint(y * 1.4426950408889634);
    double t = fma(-n, 6.9314718036912382e-01, y);
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
    const double res = ldexp(q, (int)n);
    return x == 0.0 ? 0.0 : res;
}

// x^c for NORMAL x > 0 with |c log2 x| < 1000, table-driven (pm_powtab.h, generated by gen_powtab.py): 36 VALU slots
// and two LDS lookups instead of pm_pow_pos's ~80 slots (its ten- and thirteen-term series plus seven quarter-rate
// conversions); the MCA row kernels spend half their time in this function (S * D powers per datapoint).
//   log2 x = e + L_i + log2(1 + d),  d = m r_i - 1, |d| < 2^-8  (i = top 7 mantissa bits, r_i ~ 1/m_i in 24 bits,
//            L_i = -log2 r_i), degree-6 polynomial; carried as a double-double (s, t)
//   2^y    = 2^N E_j 2^f,  y = N + j/128 + f, |f| <= 2^-8, degree-5 polynomial; 2^N by an exponent-field add
// Relative error <= 2.3e-16 against a 200-bit reference over x in [e^-30, e^30], c in {1/rho, 1/rho - 1} (the series
// version reaches 2e-15 there: its log is a single double).  x = 0 returns a finite value the callers discard.
// `tab`: the workgroup's LDS copy of pm_powtab_dev (pm_load_powtab).
#include "pm_powtab.h"
__device__ const double pm_powtab_dev[PM_POWTAB_LEN] = {PM_POWTAB_VALUES};

__device__ __forceinline__ void pm_load_powtab(double *s_tab, int tid, int nthreads) {
    for (int i = tid; i < PM_POWTAB_LEN; i += nthreads) s_tab[i] = pm_powtab_dev[i];
}

__device__ __forceinline__ double pm_pow_tab(double x, double c, const double *tab) {
    typedef double pm_d2 __attribute__((ext_vector_type(2)));
    const unsigned hi = (unsigned)__double2hiint(x), lo = (unsigned)__double2loint(x);
    const unsigned idx = (hi >> 13) & 127u;
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), (int)lo);       // [1, 2)
    const pm_d2 rl = reinterpret_cast<const pm_d2 *>(tab)[idx];
    // e = biased exponent - 1023 as a double without v_cvt: (2^52 + 2^31 + (e ^ 2^31)) - (2^52 + 2^31)
    const double e = __hiloint2double(0x43300000, (int)(((hi >> 20) - 1023u) ^ 0x80000000u)) - 4503601774854144.0;
    const double d = fma(m, rl.x, -1.0);
    double p = PM_POW_A6;
    p = fma(p, d, PM_POW_A5);
    p = fma(p, d, PM_POW_A4);
    p = fma(p, d, PM_POW_A3);
    p = fma(p, d, PM_POW_A2);
    p = fma(p, d, PM_POW_A1);
    const double s = e + rl.y;                          // |e| >= 1 > L_i or e == 0: (e - s) + L_i is the exact error
    const double t = fma(d, p, (e - s) + rl.y);
    const double yh = c * s;
    const double yl = fma(c, t, fma(c, s, -yh));
    const double sh = fma(yh, 128.0, 6755399441055744.0);          // 1.5 * 2^52: the integer lands in the low word
    const int k = __double2loint(sh);
    const double f = fma(sh - 6755399441055744.0, -0.0078125, yh) + yl;
    double q = PM_POW_B5;
    q = fma(q, f, PM_POW_B4);
    q = fma(q, f, PM_POW_B3);
    q = fma(q, f, PM_POW_B2);
    q = fma(q, f, PM_POW_B1);
    const double E = tab[256 + (k & 127)];
    const double r = fma(E, f * q, E);
    return __hiloint2double(__double2hiint(r) + ((k >> 7) << 20), __double2loint(r));
}

// x^(1/21 - 1) = x^(-20/21) for NORMAL x > 0 -- the power of MCA's multi-cause states at rho = 21, i.e. at every annealing
// temperature T <= 1.05 (mca_et.py:142-175: rho = 1 / (1 - 1 / max(T, 1.05))), the steady state of a run -- without a
// logarithm or an exponential:  x = 2^(21 q + r) m,  x' = 2^r m,  y = x'^(-1/21) from a table seed
//   y0 = 2^(-r/21) r_i^(1/21) (1 - d / 21),  d = m r_i - 1 < 2^-7   (r_i: pm_pow_tab's reciprocal table; error 1.5e-6)
// then  x'^(-20/21) = y0^20 (1 - res)^(-20/21),  res = 1 - x' y0^21  (|res| < 4e-5: three series terms, the fourth is 1e-18)
// with y0^20, y0^21 by repeated squaring: 15 f64 + 11 integer instructions and two LDS lookups against pm_pow_tab's 36 + 2,
// and a dependent chain of 14 instead of ~20.  The rounding errors of the power chain cancel between y0^20 and the
// residual: relative error <= 3e-16 against an 80-bit reference over [e^-190, e^40] (pm_pow_tab: 2.3e-16).
// `rt`: the workgroup's LDS table from pm_load_root21 -- 128 pairs (r_i, r_i^(1/21)), then 2^(-r/21), r = 0 .. 20.
#define PM_ROOT21_LEN (256 + 21)
__device__ __forceinline__ void pm_load_root21(double *rt, const double *powtab, int tid, int nthreads) {
    for (int i = tid; i < 128; i += nthreads) {
        const double ri = powtab[2 * i];
        rt[2 * i] = ri;
        rt[2 * i + 1] = pow(ri, 1.0 / 21.0);
    }
    for (int r = tid; r < 21; r += nthreads) rt[256 + r] = exp2(-(double)r / 21.0);
}
#ifndef PM_POW_HWSEED
#define PM_POW_HWSEED 1     // the roots' seed from v_log_f32 / v_exp_f32 (round 5: no LDS lookup in the chain, 27 fewer registers in the MCA
                            // pass: 7.0-7.17 -> 6.69 ms, 2.5e-16); 0: the round-4 table seed (two LDS lookups), kept for A/B builds
#endif
__device__ __forceinline__ double pm_pow_m20_21(double x, const double *rt) {
#if PM_POW_HWSEED
    // seed y0 ~ x'^(-1/21) = exp2(-log2(x') / 21) in f32 (x' = 2^r m in [1, 2^21): relative error ~1e-7, res = 1 - x' y0^21
    // ~2e-6 -- inside the three series terms' reach); no table, no LDS access in the chain
    const unsigned hi = (unsigned)__double2hiint(x), lo = (unsigned)__double2loint(x);
    const unsigned eu = (hi >> 20) + 27u;
    const unsigned q = (eu * 3121u) >> 16;
    const unsigned r = eu - 21u * q;
    const unsigned mant = hi & 0x000FFFFFu;
    const double xp = __hiloint2double((int)(mant | ((1023u + r) << 20)), (int)lo);    // 2^r m
    const double y0 = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)xp) * (-1.0f / 21.0f));
    const double y2 = y0 * y0, y4 = y2 * y2, y8 = y4 * y4, y16 = y8 * y8;
    const double y20 = y16 * y4;
    const double res = fma(-xp, y20 * y0, 1.0);
    double t = fma(res, 20.0 * 41.0 * 62.0 / (6.0 * 9261.0), 20.0 * 41.0 / (2.0 * 441.0));
    t = fma(t, res, 20.0 / 21.0);
    const double v = fma(y20, t * res, y20);
    (void)rt;
    return __hiloint2double(__double2hiint(v) + (int)((1000u - 20u * q) << 20), __double2loint(v));
#else
    typedef double pm_d2 __attribute__((ext_vector_type(2)));
    const unsigned hi = (unsigned)__double2hiint(x), lo = (unsigned)__double2loint(x);
    const unsigned idx = (hi >> 13) & 127u;
    const pm_d2 rs = reinterpret_cast<const pm_d2 *>(rt)[idx];
    const unsigned eu = (hi >> 20) + 27u;                    // biased exponent - 1023 + 1050 in [28, 2073]
    const unsigned q = (eu * 3121u) >> 16;                   // eu / 21 (exact on that range)
    const unsigned r = eu - 21u * q;
    const unsigned mant = hi & 0x000FFFFFu;
    const double m = __hiloint2double((int)(mant | 0x3FF00000u), (int)lo);             // [1, 2)
    const double xp = __hiloint2double((int)(mant | ((1023u + r) << 20)), (int)lo);    // 2^r m
    const double d = fma(m, rs.x, -1.0);
    const double y0 = (rs.y * rt[256 + r]) * fma(d, -1.0 / 21.0, 1.0);
    const double y2 = y0 * y0, y4 = y2 * y2, y8 = y4 * y4, y16 = y8 * y8;
    const double y20 = y16 * y4;
    const double res = fma(-xp, y20 * y0, 1.0);
    double t = fma(res, 20.0 * 41.0 * 62.0 / (6.0 * 9261.0), 20.0 * 41.0 / (2.0 * 441.0));
    t = fma(t, res, 20.0 / 21.0);
    const double v = fma(y20, t * res, y20);
    // * 2^(-20 (q - 50))
    return __hiloint2double(__double2hiint(v) + (int)((1000u - 20u * q) << 20), __double2loint(v));
#endif
}

// x^(1/6 - 1) = x^(-5/6) for NORMAL x > 0: the same construction for rho = 6 -- MMCA's power at every temperature T <= 1.2
// (mmca_et.py:37, 127-199: rho = 1 / (1 - 1 / max(T, 1.2))), the steady state of its runs.  x = 2^(6 q + r) m, x' = 2^r m,
// y0 = 2^(-r/6) r_i^(1/6) (1 - d / 6) (error 7 d^2 / 72 < 6e-6), x'^(-5/6) = y0^5 (1 - res)^(-5/6), res = 1 - x' y0^6 (|res| < 4e-5:
// three series terms, the fourth is 1e-18).  12 f64 + 11 integer instructions.  `rt` from pm_load_root6 (layout of pm_load_root21).
__device__ __forceinline__ void pm_load_root6(double *rt, const double *powtab, int tid, int nthreads) {
    for (int i = tid; i < 128; i += nthreads) {
        const double ri = powtab[2 * i];
        rt[2 * i] = ri;
        rt[2 * i + 1] = pow(ri, 1.0 / 6.0);
    }
    for (int r = tid; r < 6; r += nthreads) rt[256 + r] = exp2(-(double)r / 6.0);
}
__device__ __forceinline__ double pm_pow_m5_6(double x, const double *rt) {
#if PM_POW_HWSEED
    const unsigned hi = (unsigned)__double2hiint(x), lo = (unsigned)__double2loint(x);
    const unsigned eu = (hi >> 20) + 27u;
    const unsigned q = (eu * 10923u) >> 16;
    const unsigned r = eu - 6u * q;
    const unsigned mant = hi & 0x000FFFFFu;
    const double xp = __hiloint2double((int)(mant | ((1023u + r) << 20)), (int)lo);    // 2^r m
    const double y0 = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)xp) * (-1.0f / 6.0f));
    const double y2 = y0 * y0, y4 = y2 * y2;
    const double y5 = y4 * y0;
    const double res = fma(-xp, y5 * y0, 1.0);
    double t = fma(res, 5.0 * 11.0 * 17.0 / (6.0 * 216.0), 5.0 * 11.0 / (2.0 * 36.0));
    t = fma(t, res, 5.0 / 6.0);
    const double v = fma(y5, t * res, y5);
    (void)rt;
    return __hiloint2double(__double2hiint(v) + (int)((875u - 5u * q) << 20), __double2loint(v));
#else
    typedef double pm_d2 __attribute__((ext_vector_type(2)));
    const unsigned hi = (unsigned)__double2hiint(x), lo = (unsigned)__double2loint(x);
    const unsigned idx = (hi >> 13) & 127u;
    const pm_d2 rs = reinterpret_cast<const pm_d2 *>(rt)[idx];
    const unsigned eu = (hi >> 20) + 27u;                    // biased exponent - 1023 + 1050 (= 6 * 175) in [28, 2073]
    const unsigned q = (eu * 10923u) >> 16;                  // eu / 6 (exact on that range)
    const unsigned r = eu - 6u * q;
    const unsigned mant = hi & 0x000FFFFFu;
    const double m = __hiloint2double((int)(mant | 0x3FF00000u), (int)lo);             // [1, 2)
    const double xp = __hiloint2double((int)(mant | ((1023u + r) << 20)), (int)lo);    // 2^r m
    const double d = fma(m, rs.x, -1.0);
    const double y0 = (rs.y * rt[256 + r]) * fma(d, -1.0 / 6.0, 1.0);
    const double y2 = y0 * y0, y4 = y2 * y2;
    const double y5 = y4 * y0;
    const double res = fma(-xp, y5 * y0, 1.0);
    double t = fma(res, 5.0 * 11.0 * 17.0 / (6.0 * 216.0), 5.0 * 11.0 / (2.0 * 36.0));
    t = fma(t, res, 5.0 / 6.0);
    const double v = fma(y5, t * res, y5);
    // * 2^(-5 (q - 175))
    return __hiloint2double(__double2hiint(v) + (int)((875u - 5u * q) << 20), __double2loint(v));
#endif
}

// x^c for NORMAL x > 0 and an exponent c that is UNIFORM over the launch (round 6: MCA / MMCA's state power at any
// temperature, c = 1 / rho - 1 with rho = 1 / (1 - 1 / T) -- every step of an annealing ramp; pm_pow_tab's 36 slots + two
// dependent lookups made those passes 18 % slower than the rho = 21 ones).  With c fixed nothing needs a logarithm:
//   x = 2^(E - 1023) m,  m = (1 + d) / r_i  (i = top 7 mantissa bits, r_i ~ 1 / m_i: pm_pow_tab's table, |d| < 2^-8)
//   x^c = 2^(c (64 E1 - 1023)) * 2^(c E0) * r_i^(-c) * (1 + d)^c,   E = 64 E1 + E0
// -- three factors from tables of 32 + 64 + 128 entries that the workgroup builds for ITS c when it starts (pm_load_upow:
// libm powers, once), and the binomial series of (1 + d)^c to degree 6 (the seventh term is below 2^-56) with coefficients
// held beside the tables.  ~15 VALU slots and three independent LDS reads; relative error <= 5e-16 against an 80-bit
// reference (scratch/pow_uni_check.hip): four correctly rounded factors and the series' rounding.
// |c| <= 1 (the factor tables must not overflow).
// Layout: `pairs` = 128 x (r_i, r_i^-c) -- the place of pm_pow_tab's (r_i, L_i) pairs in the workgroup's copy of the power
// table, whose E_j part pm_exp_tab keeps using --, `ab` = [A (32) | B (64) | b1 .. b6 | pad]: PM_UPOW_AB_LEN doubles.
#define PM_UPOW_AB_LEN (32 + 64 + 8)
__device__ __forceinline__ void pm_load_upow(double *pairs, double *ab, const double *powtab, double c, int tid, int nthreads) {
    for (int i = tid; i < 128; i += nthreads) {
        const double ri = powtab[2 * i];
        pairs[2 * i] = ri;
        pairs[2 * i + 1] = pow(ri, -c);
    }
    // (2^(c n) with the product c n carried exactly: hi + lo, 2^hi (1 + lo ln 2) -- the rounding of c n alone would cost
    // |c n| 2^-53 ~ 1e-13)
    for (int i = tid; i < 96; i += nthreads) {
        const double n = i < 32 ? (double)(64 * i - 1023) : (double)(i - 32);
        const double eh = c * n, el = fma(c, n, -eh);
        ab[i] = exp2(eh) * fma(el, 0.6931471805599453, 1.0);
    }
    if (tid < 8) {
        double b = 1.0;      // b_k = c (c - 1) ... (c - k + 1) / k!
        for (int k = 1; k <= tid + 1; ++k) b *= (c - (double)(k - 1)) / (double)k;
        ab[96 + tid] = tid < 6 ? b : 0.0;
    }
}
__device__ __forceinline__ double pm_pow_uni(double x, const double *pairs, const double *ab) {
    typedef double pm_d2 __attribute__((ext_vector_type(2)));
    const unsigned hi = (unsigned)__double2hiint(x), lo = (unsigned)__double2loint(x);
    const unsigned idx = (hi >> 13) & 127u;
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), (int)lo);       // [1, 2)
    const pm_d2 rl = reinterpret_cast<const pm_d2 *>(pairs)[idx];
    const double a = ab[hi >> 26];
    const double b = ab[32 + ((hi >> 20) & 63u)];
    const double d = fma(m, rl.x, -1.0);
    double p = ab[101];
    p = fma(p, d, ab[100]);
    p = fma(p, d, ab[99]);
    p = fma(p, d, ab[98]);
    p = fma(p, d, ab[97]);
    p = fma(p, d, ab[96]);
    const double t = rl.y * (a * b);
    return fma(t * d, p, t);
}

// e^x for x <= ~700 from the same tables: x = k ln2/128 + r (two-part ln2/128, |r| <= ln2/256), e^r by a degree-5
// polynomial, 2^(k/128) = 2^N E_j.  Arguments below -708 return ~1e-308 (callers only scale by it).  14 VALU slots and one
// LDS lookup; relative error <= 2.3e-16.
__device__ __forceinline__ double pm_exp_tab(double x, const double *tab) {
    x = x < -708.0 ? -708.0 : x;
    const double sh = fma(x, 184.6649652337873, 6755399441055744.0);       // 128 / ln 2
    const int k = __double2loint(sh);
    const double kf = sh - 6755399441055744.0;
    double r = fma(kf, -0.00541521234663378, x);                            // ln2/128, high part (32 significant bits)
    r = fma(kf, -1.4907929134926466e-12, r);                                // ... low part
    double q = 1.0 / 120.0;
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    const double E = tab[256 + (k & 127)];
    const double v = fma(E, r * q, E);
    return __hiloint2double(__double2hiint(v) + ((k >> 7) << 20), __double2loint(v));
}

// Packed BSC statistics buffer: [ Wp (H*D) | Wq (H*H) | qdiag (H) | mus (H) | scalars ]
__host__ __device__ inline int64_t pm_bsc_stats_offset_wq_dev(int64_t H, int64_t D) { return H * D; }
__host__ __device__ inline int64_t pm_bsc_stats_offset_qdiag_dev(int64_t H, int64_t D) { return H * D + H * H; }
__host__ __device__ inline int64_t pm_bsc_stats_offset_mus_dev(int64_t H, int64_t D) { return H * D + H * H + H; }
__host__ __device__ inline int64_t pm_bsc_stats_offset_scalars_dev(int64_t H, int64_t D) { return H * D + H * H + 2 * H; }

#endif  // PM_COMMON_H
