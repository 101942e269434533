// Binary Sparse Coding: scores GEMM + select_Hprimes + E_step (+ the M-step's row statistics) in ONE kernel
// (bsc_et.py:98-192, 334-366, 395-415), config 2's shape class: H in (128, 256], H' in 5 .. 8 (round 6; H' = 8 until then), gamma in {3, 4}.
//
// A workgroup of SIXTEEN wavefronts owns 128 datapoints x 256 latents (one workgroup per CU): wavefront (rg, half) =
// (wave & 7, wave >> 3) accumulates datapoints 16 rg .. 16 rg + 15 against latents 128 half .. 128 half + 127 -- 64
// accumulator registers, <= 128 in total, four wavefronts per SIMD.  K-loop: an LDS-DMA ring (tile16_scores); then four
// "lean" row passes, each over 32 datapoints: the scores cross from the MFMA layout to one half-wavefront per datapoint
// through LDS, which owns it from ranking to log-evidence.  A ragged last round of the shard runs in the TAIL
// instantiation (16-row workgroups of 8 wavefronts that split K four ways).  DESIGN.md section 4.1 has the measurements
// that shaped this; the variants that lost them (8-wavefront tiles with and without lean passes, 32 x 64 blocks per
// wavefront, the old LDS swizzle, register-staged TAIL loads, ...) are kept in scratch/bsc_fused8_r03_variants.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "prosper_hip.h"
#include "pm_common.h"
#include "bsc_rows16_body.h"

namespace pm_fused8 {

using namespace pm_rows16;

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int DK = 8;          // K columns per ring stage
constexpr int HT = 256;        // latent rows of a stage (H in (128, 256], rows beyond H shadow row H - 1)
constexpr int NJ = 8;          // 16-latent column blocks per wavefront
constexpr int THREADS = 512;   // the TAIL workgroup (the main one has 1024)

__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Diagnostic builds only (scratch/f8_bench.hip, scratch/f8_variants.sh), never in the shipped library:
//   -DPM_F8_STAMPS   per-workgroup timeline (s_memrealtime, 100 MHz) and shader clock over the K-loop
//   -DPM_F8_ABL=1    K-loop only (no row passes)          =2  row passes behind a K-loop of 8 steps
//               =3   K-loop without its hand-over barrier (wrong results; what the barrier costs)
#ifndef PM_F8_ABL
#define PM_F8_ABL 0
#endif
//   -DPM_F8_SKIP=mask  timing-only ablations of the lean row passes (wrong results): 1 no ranking (keys, sort, pop
//               rounds), 2 no multi-cause states, 4 no log-joint stores, 8 no log-sum-exp, 16 no barriers in the passes,
//               32 (M-statistics) no E[s] row stores, 64 no Wq / mus atomics
#ifndef PM_F8_SKIP
#define PM_F8_SKIP 0
#endif
#define F8_STORE(p, v) (*reinterpret_cast<double *>(p) = (v))
#ifdef PM_F8_STAMPS
__device__ unsigned long long pm_f8_stamps[8192][8];
__device__ unsigned long long pm_f8_estamps[16][32];       // workgroup 0's wavefronts: phase stamps inside the row passes
#define F8_ESTAMP(slot)                                                                           \
    do {                                                                                          \
        if (blockIdx.x == 300 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 16)              \
            pm_f8_estamps[threadIdx.x >> 6][slot] = __builtin_amdgcn_s_memrealtime();             \
    } while (0)
#define F8_STAMP(slot)                                                                  \
    do {                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 8192)                                      \
            pm_f8_stamps[blockIdx.x][slot] = __builtin_amdgcn_s_memrealtime();          \
    } while (0)
#else
#define F8_STAMP(slot)
#define F8_ESTAMP(slot)
#endif

// workgroup barrier that publishes LDS writes but does NOT wait for global stores (a workgroup-scope fence would drain
// vmcnt, i.e. every log-joint store of the pass)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// reductions over the 32 lanes of a datapoint in the by-datapoint mapping (ds_swizzle bit mode covers 32 lanes)
__device__ __forceinline__ double half_max_f64(double v) {
    v = vmax64(v, swz_xor_f64<1>(v));
    v = vmax64(v, swz_xor_f64<2>(v));
    v = vmax64(v, swz_xor_f64<4>(v));
    v = vmax64(v, swz_xor_f64<8>(v));
    v = vmax64(v, swz_xor_f64<16>(v));
    return v;
}
// The same maximum with the four levels inside a 16-lane row as DPP moves (register-file latency) and only the level
// across the two rows through the LDS crossbar: 13 VALU instructions instead of 5, a fifth of the dependent latency --
// what the pop rounds of the 16-wavefront kernel want (its row passes run with the matrix pipe idle: latency, not the
// instruction count, is their price).
__device__ __forceinline__ double half_max_dpp(double v) {
    v = vmax64(v, pm_dpp_f64<0xB1>(v));
    v = vmax64(v, pm_dpp_f64<0x4E>(v));
    v = vmax64(v, pm_dpp_f64<0x141>(v));
    v = vmax64(v, pm_dpp_f64<0x140>(v));
    v = vmax64(v, swz_xor_f64<16>(v));
    return v;
}
__device__ __forceinline__ double half_sum_f64(double v) {
    v += swz_xor_f64<1>(v);
    v += swz_xor_f64<2>(v);
    v += swz_xor_f64<4>(v);
    v += swz_xor_f64<8>(v);
    v += swz_xor_f64<16>(v);
    return v;
}

// The scores block of one wavefront of the 16-wavefront workgroup (128 datapoints x 256 latents; row group wave & 7,
// latent half wave >> 3): acc[i][r] = <y_n, W_h>, n = m0 + 16 rg + (lane >> 4) + 4 r, h = 128 half + (lane & 15) + 16 i.
template <int STAGES>
__device__ __forceinline__ void tile16_scores(d4 (&acc)[NJ], double *sm, const double *__restrict__ Y, int64_t ldy,
                                             const double *__restrict__ Wt, int64_t ldw, int D, int64_t N, int H,
                                             int64_t m0, int lane, int wave) {
    const int rg = wave & 7, half = wave >> 3;
    constexpr int AROWS = 128, STAGE = (AROWS + HT) * DK;       // doubles per ring stage (24 KB)
    // ---------------- K-loop: LDS-DMA ring, one barrier per K-step ------------------------------------------------
    // DMA sources: wavefront `wave` moves latent block `wave`; the first eight wavefronts also move the datapoint rows of
    // their row group (2 resp. 1 DMA instructions per K-step: the counted waits are per wavefront).
    // A stage is [rows][8 doubles]; pair p (16 B) of row R sits in slot p ^ PI(R >> 2) of its row, PI = {0, 3, 2, 1}:
    // ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 --
    // and with this permutation every group's 16 fragment reads fall into 16 different bank quads (the plain
    // p ^ (R >> 2) of the 4-wavefront kernel is two-way conflicted: SQ_LDS_BANK_CONFLICT = half the LDS cycles).
    const int dr = lane >> 2, dj = (lane & 3) ^ ((4 - (lane >> 4)) & 3);
    const char *sbase[2];
    uint32_t soff[2];
    {
        int64_t r0 = m0 + 16 * rg;
        r0 = r0 < N ? r0 : N - 1;
        int64_t ra = r0 + dr;
        ra = ra < N ? ra : N - 1;
        sbase[0] = reinterpret_cast<const char *>(Y + r0 * ldy);
        soff[0] = (uint32_t)((ra - r0) * ldy * 8 + 16 * dj);
        {
            int b0 = 16 * wave;
            b0 = b0 < H ? b0 : H - 1;
            int rb = b0 + dr;
            rb = rb < H ? rb : H - 1;
            sbase[1] = reinterpret_cast<const char *>(Wt + (int64_t)b0 * ldw);
            soff[1] = (uint32_t)((int64_t)(rb - b0) * ldw * 8 + 16 * dj);
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) double *)(sm));
    auto dma1 = [&](unsigned dst, uint32_t voff, const char *base) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(base)
                     : "memory");
    };
    // (non-temporal loads of the datapoint rows, which are read once, measured +3 % time: DESIGN.md 4.1)

    // fragment reads: pair p of row R sits at R*8 + ((p ^ ((R>>2)&3)) << 1) (the slot the DMA wrote)
    const int frow = lane & 15, fk = lane >> 4;
    const int sw = (4 - (frow >> 2)) & 3;
    const int a_off = (rg * 16 + frow) * DK + ((fk ^ sw) << 1);
    const int b_off = AROWS * DK + (half * 128 + frow) * DK + ((fk ^ sw) << 1);
    auto read_a = [&](int stage) { return *reinterpret_cast<const d2 *>(sm + stage * STAGE + a_off); };
    auto read_b = [&](int stage, int g, d2 (&f)[4]) {
        const double *sb = sm + stage * STAGE + b_off + g * 4 * 16 * DK;
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const d2 *>(sb + q * 16 * DK);
    };

#pragma unroll
    for (int i = 0; i < NJ; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};

#if PM_F8_ABL == 2
    const int nk = 8;
#else
    const int nk = D / DK;   // host guarantees D % DK == 0, D >= DK
#endif
    auto kloop = [&](auto withA) {
        constexpr bool WA = decltype(withA)::value;
        constexpr int L = WA ? 2 : 1;          // DMA instructions per K-step of this wavefront
        auto dma = [&](int kt, int stage) {
            const unsigned dst = lds0 + (unsigned)stage * (unsigned)(STAGE * 8);
            const int64_t k0 = (int64_t)kt * (DK * 8);
            if (WA) dma1(dst + (unsigned)rg * 1024u, soff[0], sbase[0] + k0);
            dma1(dst + (8u + (unsigned)wave) * 1024u, soff[1], sbase[1] + k0);
        };
#pragma unroll
        for (int t = 0; t < STAGES; ++t)
            if (t < nk) dma(t, t);
        {
            const int behind = (nk < STAGES ? nk : STAGES) - 1;   // K-steps issued beyond step 0
            if (behind >= 3) wait_vmcnt<3 * L>();
            else if (behind == 2) wait_vmcnt<2 * L>();
            else if (behind == 1) wait_vmcnt<L>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();

        d2 fa[2];
        d2 fb[2][4];
        fa[0] = read_a(0);
        read_b(0, 0, fb[0]);
        auto kstep = [&](int t, int stage, int nstage, int par) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (g == 0) {
                    read_b(stage, 1, fb[1]);
                } else if (t + 1 < nk) {
                    // my reads of this stage are done (lgkmcnt) and my share of K-step t+1 has landed (vmcnt); after the
                    // barrier that holds for every wavefront: stage t may be refilled, t+1 may be read
                    int ahead = nk - t - 2;
                    ahead = ahead < STAGES - 2 ? ahead : STAGES - 2;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (ahead >= 2) wait_vmcnt<2 * L>();
                    else if (ahead == 1) wait_vmcnt<L>();
                    else wait_vmcnt<0>();
#if PM_F8_ABL != 3
                    __builtin_amdgcn_s_barrier();
#endif
                    const d2 an = read_a(nstage);
                    if (par) fa[0] = an;
                    else fa[1] = an;
                    read_b(nstage, 0, fb[0]);
                    if (t + STAGES < nk) dma(t + STAGES, stage);
                }
                const d2 af = par ? fa[1] : fa[0];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[4 * g + q] = mfma16(af.x, fb[g][q].x, acc[4 * g + q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[4 * g + q] = mfma16(af.y, fb[g][q].y, acc[4 * g + q]);
            }
        };
        int t = 0;
        if (STAGES % 2 == 0) {
            for (; t + STAGES <= nk; t += STAGES) {
#pragma unroll
                for (int u = 0; u < STAGES; ++u) kstep(t + u, u, (u + 1) % STAGES, u & 1);
            }
        }
        for (int stage = t % STAGES; t < nk; ++t) {
            const int nstage = (stage + 1 == STAGES) ? 0 : stage + 1;
            kstep(t, stage, nstage, t & 1);
            stage = nstage;
        }
    };
    if (half == 0) kloop(std::true_type{});
    else kloop(std::false_type{});

}

// =====================================================================================================================
// The lean row passes ("s" = specialised): H' and gamma are compile-time, so the truncated state set's structure is too
// (C(H',g) states of size g, contiguous by size: generate_state_matrix, camodels/__init__.py:21-47) and every loop below
// is straight-line code with immediate LDS / global offsets.  What the counters said about the first version of this
// file (profiles/r03_*): in the fused kernel every VALU instruction of a row pass costs ~7 cycles of a SIMD that could
// issue MFMAs, and nothing else of a row pass costs anything -- so the passes are organised to minimise VALU
// INSTRUCTIONS, not latency:
//   * each pass first transposes its 16 datapoints x 256 scores through LDS (8 ds_write + 8 ds_read per lane, no VALU):
//     afterwards ONE half-wavefront (32 lanes, lane j holds latents j + 32 i) owns a datapoint from ranking to
//     log-evidence -- no duplicated pop rounds, no merge, no partial maxima / sums across wavefronts, two barriers per
//     pass (scores written | scores read), both cheap to reach;
//   * the state table's five operand offsets per state become LDS addresses with one add each; results go out through
//     immediate offsets off one address register per datapoint;
//   * the log of the log-sum-exp is taken once per tile for all passes.
template <int HP, int GAMMA>
struct StateSet {
    static constexpr int cnt(int g) {
        int r = 1;
        for (int i = 0; i < g; ++i) r = r * (HP - i) / (i + 1);
        return r;
    }
    static constexpr int off(int g) {       // index of the first state of size g (sizes 2 .. GAMMA)
        int o = 0;
        for (int q = 2; q < g; ++q) o += cnt(q);
        return o;
    }
    static constexpr int S = off(GAMMA + 1);
    static constexpr int iters(int g) { return (cnt(g) + 31) / 32; }
};

// LDS map of the lean passes (bytes from the workgroup's LDS base; aliases the ring)
constexpr int T_SW = 0, T_EW = 2048, T_MUS = 4096, T_ST = 6144;
constexpr int S_MAX8 = 160;                                   // states the tables are sized for
constexpr int T_TAB = T_ST + 16 * S_MAX8, T_EXP = T_TAB + 4 * S_MAX8;        // 8704, 9344
constexpr int T_EXPC = T_EXP + 1024, T_AREAS = T_EXPC + 384;                 // 10368, 10752 (EXPC: 7 constants; 48 doubles at the end)
constexpr int A_ROW = 0, A_P = 2048, A_WIN = 4096, A_MISC = 4160, AREA_BYTES = 4288;
constexpr int LEAN_LDS_BYTES = T_AREAS + 16 * AREA_BYTES;     // 79360
static_assert(LEAN_LDS_BYTES <= 80 * 1024, "two workgroups per CU");
// (16-wavefront workgroup only: behind the areas, a 36-entry block per datapoint slot where the statistics pass gathers
// E[s_i s_k] of the candidate pairs before they go to Wq -- see PAIRLDS in the kernel)
constexpr int T_PAIRS16 = T_AREAS + 32 * AREA_BYTES, PAIR_ENTRIES = 36;
constexpr int LEAN16_LDS_BYTES = T_PAIRS16 + 32 * PAIR_ENTRIES * 8;   // 157184: the 16-wavefront workgroup, one per CU
static_assert(LEAN16_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// e^x for x in [-708, 0] from LDS tables: E[j] = 2^(j/128) (128 doubles) and C = {128/ln2, 1.5 2^52, -ln2/128 hi, lo,
// 1/120, 1/24, 1/6}.  x = k ln2/128 + r, e^r by a degree-5 polynomial, 2^(k/128) = 2^N E_j (pm_exp_tab of pm_common.h
// with every constant read from LDS: 14 VALU instructions -- a literal f64 operand costs two v_mov each, and in the
// fused kernel VALU instructions are the currency).  Relative error <= 2.3e-16.
__device__ __forceinline__ double exp_lds(double x, const double *E, const double *C) {
    const double c1 = C[1];
    const double sh = fma(x, C[0], c1);
    const int k = __double2loint(sh);
    const double kf = sh - c1;
    double r = fma(kf, C[2], x);
    r = fma(kf, C[3], r);
    double q = C[4];
    q = fma(q, r, C[5]);
    q = fma(q, r, C[6]);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    const double Ej = E[k & 127];
    const double v = fma(Ej, r * q, Ej);
    return __hiloint2double(__double2hiint(v) + ((k >> 7) << 20), __double2loint(v));
}

// The ragged last round (TAIL kernels): a workgroup owns 16 datapoints x 256 latents and splits K four ways --
// wavefront (kq, half) = (wave & 3, wave >> 2) accumulates latents 128 half .. over K-steps [kq nk/4, (kq+1) nk/4).  A
// whole-rounds launch leaves N mod 32768 rows (3392 of 200 000 at config 2), for which 128-row tiles would run 27
// workgroups on 27 of 256 CUs for a full tile time; 16-row tiles make 212 workgroups of a quarter of the K-loop each.
constexpr int TAIL_ROWS = 16, TAIL_DK = 8;
// No operand is shared between wavefronts, so every wavefront streams its own through a PRIVATE two-stage LDS ring
// filled by LDS-DMA (a CU pulls ~25-30 GB/s from L2 into registers with vector loads -- the first version, 0.082 ms --,
// 2-3 times that into LDS): stage = [A chunk | 8 latent chunks] x 1 KB = 9 KB, 2 stages x 8 wavefronts = 144 KB (one
// workgroup per CU).  No barrier in the loop: nothing is shared.
constexpr int TAIL_STAGE_BYTES = 9 * 1024, TAIL_RING_BYTES = 8 * 2 * TAIL_STAGE_BYTES;
__device__ __forceinline__ void tail8_scores_dma(d4 (&acc)[NJ], double *sm, const double *__restrict__ Y, int64_t ldy,
                                                 const double *__restrict__ Wt, int64_t ldw, int D, int64_t N, int H,
                                                 int64_t m0, int lane, int wave) {
    const int kq = wave & 3, half = wave >> 2;
    const int nk = D / DK;
    const int t0 = (kq * nk) / 4, t1 = ((kq + 1) * nk) / 4;
    const int dr = lane >> 2, dj = (lane & 3) ^ ((4 - (lane >> 4)) & 3);
    int64_t r0 = m0 < N ? m0 : N - 1;
    int64_t ra = m0 + dr;
    ra = ra < N ? ra : N - 1;
    const char *baseA = reinterpret_cast<const char *>(Y + r0 * ldy);
    const uint32_t voffA = (uint32_t)((ra - r0) * ldy * 8 + 16 * dj);
    const char *baseB = reinterpret_cast<const char *>(Wt);
    uint32_t voffB[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        int rb = half * 128 + 16 * i + dr;
        rb = rb < H ? rb : H - 1;
        voffB[i] = (uint32_t)rb * (uint32_t)ldw * 8u + 16u * (uint32_t)dj;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) double *)(sm)) + (unsigned)wave * (2u * TAIL_STAGE_BYTES);
    auto dma1 = [&](unsigned dst, uint32_t voff, const char *base) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(base)
                     : "memory");
    };
    auto dma = [&](int t, int stage) {
        const unsigned dst = lds0 + (unsigned)stage * TAIL_STAGE_BYTES;
        const int64_t k0 = (int64_t)t * (DK * 8);
        dma1(dst, voffA, baseA + k0);
#pragma unroll
        for (int i = 0; i < NJ; ++i) dma1(dst + (1u + i) * 1024u, voffB[i], baseB + k0);
    };
    const int frow = lane & 15, fk = lane >> 4;
    const int sw = (4 - (frow >> 2)) & 3;
    const double *mine = sm + (size_t)wave * (2 * TAIL_STAGE_BYTES / 8) + frow * DK + ((fk ^ sw) << 1);
#pragma unroll
    for (int i = 0; i < NJ; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    if (t0 < t1) dma(t0, 0);
    if (t0 + 1 < t1) dma(t0 + 1, 1);
    for (int t = t0; t < t1; ++t) {
        const int stage = (t - t0) & 1;
        if (t + 1 < t1) wait_vmcnt<9>();        // this step's nine pieces have landed (the next step's may be in flight)
        else wait_vmcnt<0>();
        const double *st = mine + stage * (TAIL_STAGE_BYTES / 8);
        const d2 a = *reinterpret_cast<const d2 *>(st);
        d2 b[NJ];
#pragma unroll
        for (int i = 0; i < NJ; ++i) b[i] = *reinterpret_cast<const d2 *>(st + (1 + i) * 128);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + 2 < t1) dma(t + 2, stage);      // (my reads of this stage are in registers)
#pragma unroll
        for (int i = 0; i < NJ; ++i) acc[i] = mfma16(a.x, b[i].x, acc[i]);
#pragma unroll
        for (int i = 0; i < NJ; ++i) acc[i] = mfma16(a.y, b[i].y, acc[i]);
    }
}

// W16 (the shipped main launch): a workgroup of SIXTEEN wavefronts owns 128 datapoints, one per CU (4 wavefronts per
// SIMD as with two 8-wavefront workgroups).  Measured on the stamps (profiles/r03_*): with two independent workgroups per
// CU one ends up in its row passes while the other K-loops -- the row passes, throttled to one VALU slot per MFMA, then
// take 118 us instead of the 21 us they need alone, and the lone K-loop 144 us instead of the 109 us two K-loops take per
// tile when they share a SIMD.  One workgroup keeps all 16 wavefronts in the same phase: an MFMA-bound K-loop over
// 128 rows (W fetched once per 128 rows), then row passes at full vector rate.
template <int STAGES, int HP, int GAMMA, bool FULL, bool MSTATS, bool TAIL>
__global__ __launch_bounds__(TAIL ? THREADS : 1024, TAIL ? 2 : 4) void bsc_estep_fused8s_kernel(
    const double *__restrict__ Y, int64_t ldy, const double *__restrict__ Wt, int64_t ldw, int D,
    const double *__restrict__ gram, const double *__restrict__ ynorm2, const double *__restrict__ wmu,
    const double *__restrict__ ymu, const uint16_t *__restrict__ masks, const uint16_t *__restrict__ parents,
    pm_bsc_estep_params P, int64_t N, int H, int mode, int32_t *__restrict__ cand, double *__restrict__ logpj,
    int64_t ldl, double *__restrict__ lse, double *__restrict__ expect, int64_t lde, double *__restrict__ stats,
    int Dstats, int64_t row0, uint16_t *__restrict__ nz_idx, double *__restrict__ nz_val,
    double *__restrict__ defer) {
    using SS = StateSet<HP, GAMMA>;
    constexpr bool W16 = !TAIL;                       // the main launch: sixteen wavefronts, 128 datapoints
    constexpr int TILE_ROWS = TAIL ? TAIL_ROWS : 128, NPASS = TAIL ? 1 : 4;
    constexpr int NWAVES = W16 ? 16 : 8, RGMASK = W16 ? 7 : 3, HSHIFT = W16 ? 3 : 2;
    constexpr int S = SS::S;
    // (round 6: H' = 5 .. 8.  The LAYOUT stays that of eight candidate positions -- the lane <-> Gram entry mapping, the 8 x 8
    // block in P, the 36-entry pair block --, positions H' .. 7 hold latent 0 and no state refers to them; the selection pops
    // H' winners and the state set is that of H' positions.)
    constexpr int PH = 8;
    constexpr int O_D = 8, O_G = 8 * 9, O_E = 8 * (9 + PH * PH);          // byte offsets inside P = [zero | d | G | e]
    static_assert(9 + PH * PH + S <= 256 && S <= S_MAX8, "P = [zero | d (8) | G | e] must fit the 2 KB list area");
    static_assert(HP >= 4 && HP <= PH && GAMMA <= HP, "H' between 4 and the eight positions of the layout");
    extern __shared__ __attribute__((aligned(1024))) double sm[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & RGMASK, half = wave >> HSHIFT;
    const int64_t m0 = row0 + (int64_t)blockIdx.x * TILE_ROWS;
    F8_STAMP(0);
#ifdef PM_F8_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
    if (tid == 0 && blockIdx.x < 8192)
        pm_f8_stamps[blockIdx.x][7] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) |
                                      __builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
#endif
    d4 acc[NJ];
    if (TAIL) {
        tail8_scores_dma(acc, sm, Y, ldy, Wt, ldw, D, N, H, m0, lane, wave);
        lds_barrier();                   // the rings become the reduction scratch
        // the four K-quarters of a latent half summed in a fixed order, (q0 + q2) + (q1 + q3), through 64 KB of LDS
        double *red = sm + (size_t)((wave & 1) + 2 * half) * (NJ * 4 * 64) + lane;      // [slot][i][r][lane]
        if (rg >= 2) {
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(i * 4 + r) * 64] = acc[i][r];
        }
        lds_barrier();
        if (rg < 2) {
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] += red[(i * 4 + r) * 64];
        }
        lds_barrier();
        double *red2 = sm + (size_t)(1 + 2 * half) * (NJ * 4 * 64) + lane;
        if (rg == 1) {
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) red2[(i * 4 + r) * 64] = acc[i][r];
        }
        lds_barrier();
        if (rg == 0) {
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] += red2[(i * 4 + r) * 64];
        }
    } else {
        tile16_scores<STAGES>(acc, sm, Y, ldy, Wt, ldw, D, N, H, m0, lane, wave);
    }
    F8_STAMP(1);
#ifdef PM_F8_STAMPS
    if (tid == 0 && blockIdx.x < 8192) pm_f8_stamps[blockIdx.x][6] = __builtin_amdgcn_s_memtime() - clk0;
#endif
#if PM_F8_ABL == 1
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) s += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
        if (s == 1.2345e300) lse[0] = s;
        return;
    }
#endif
    lds_barrier();                       // every wavefront is done with the ring (TAIL: with the reduction scratch)
    unsigned char *smem = reinterpret_cast<unsigned char *>(sm);
    const double ppil = P.prior_scale * P.pil_bar, ecoef = P.ecoef;
    const double m2e = -2.0 * ecoef;
    const bool sel = mode & 1, est = mode & 2;

    // ---- workgroup tables ----
    if (tid < HT) {
        const int hc = tid < H ? tid : H - 1;
        const double g = gram[(int64_t)hc * H + hc];
        const double w = g + (wmu ? 2.0 * wmu[hc] : 0.0);
        reinterpret_cast<double *>(smem + T_SW)[tid] = 1.0 / sqrt(g);
        reinterpret_cast<double *>(smem + T_EW)[tid] = fma(ecoef, w, ppil);
        if (MSTATS) reinterpret_cast<double *>(smem + T_MUS)[tid] = 0.0;
    } else if (tid - HT < S) {
        const int s = tid - HT;
        const uint32_t t = (uint32_t)masks[s] | ((uint32_t)parents[s] << 16);
        const unsigned mask = t & 0xFFFFu, par = t >> 16;
        reinterpret_cast<uint32_t *>(smem + T_TAB)[s] = t;
        const int k = 31 - __builtin_clz(mask);
        unsigned rest = mask & ~(1u << k);
        const int g = __builtin_popcount(mask);
        const int b0 = __builtin_ctz(rest);
        const unsigned e0 = (g == 2) ? (unsigned)(O_D + 8 * b0) : (unsigned)(O_E + 8 * par);
        const unsigned e1 = O_D + 8 * k;
        const unsigned e2 = O_G + 8 * (b0 * PH + k);
        unsigned e3 = 0, e4 = 0;
        rest &= rest - 1;
        if (rest) {
            e3 = O_G + 8 * (__builtin_ctz(rest) * PH + k);
            rest &= rest - 1;
        }
        if (rest) e4 = O_G + 8 * (__builtin_ctz(rest) * PH + k);
        uint32_t *dst = reinterpret_cast<uint32_t *>(smem + T_ST + 16 * s);
        dst[0] = e0 | (e1 << 16);
        dst[1] = e2 | (e3 << 16);
        dst[2] = e4;
        dst[3] = 0;
    }
    if (tid >= 384 && tid < 512) {
        const int j = tid - 384;                                      // 128 threads: the exponential's tables
        reinterpret_cast<double *>(smem + T_EXP)[j] = pm_powtab_dev[256 + j];
        if (j < 7) {
            const double c = j == 0 ? 184.6649652337873 : j == 1 ? 6755399441055744.0 : j == 2 ? -0.00541521234663378
                           : j == 3 ? -1.4907929134926466e-12 : j == 4 ? 1.0 / 120.0 : j == 5 ? 1.0 / 24.0 : 1.0 / 6.0;
            reinterpret_cast<double *>(smem + T_EXPC)[j] = c;
        }
    }
    const double *expE = reinterpret_cast<const double *>(smem + T_EXP);
    const double *expC = reinterpret_cast<const double *>(smem + T_EXPC);

    // ---- roles ----
    // accumulating role (MFMA layout): row fk of this wavefront = datapoint 16 rg + fk + 4 r in element r; its scores go
    // to the area of slot 4 rg + fk, entries 128 half + j16 + 16 i
    const int j16 = lane & 15, fk = lane >> 4;
    double *rowW = reinterpret_cast<double *>(smem + T_AREAS + (rg * 4 + fk) * AREA_BYTES + A_ROW) + half * 128 + j16;
    // owner role: lanes 32 dsel .. 32 dsel + 31 own the datapoint of slot 4 rg + 2 half + dsel, lane j32 holds latents
    // j32 + 32 i
    const int j32 = lane & 31, dsel = lane >> 5;
    const int fkM = 2 * half + dsel;
    // (TAIL: the workgroup's 16 datapoints are the 16 slots, one pass; slot 2 wave + dsel is this half-wavefront's)
    const int slot = TAIL ? 2 * wave + dsel : rg * 4 + fkM;
    unsigned char *area = smem + T_AREAS + slot * AREA_BYTES;
    const double *rowR = reinterpret_cast<const double *>(area + A_ROW) + j32;
    double *Pm = reinterpret_cast<double *>(area + A_P);
    unsigned char *Pb = area + A_P;
    double *listA = Pm + j32;
    double *win = reinterpret_cast<double *>(area + A_WIN);
    int *cl = reinterpret_cast<int *>(area + A_MISC);
    // (round 6: the TAIL workgroup gathers its pair blocks in LDS as well -- behind its 16 areas; its ring is larger)
    constexpr bool PAIRLDS = MSTATS;
    constexpr int T_PAIRS = W16 ? T_PAIRS16 : LEAN_LDS_BYTES;
    static_assert(LEAN_LDS_BYTES + 16 * PAIR_ENTRIES * 8 <= TAIL_RING_BYTES, "the TAIL workgroup's pair blocks fit its LDS");
    double *Bp = reinterpret_cast<double *>(smem + T_PAIRS) + (PAIRLDS ? slot : 0) * PAIR_ENTRIES;
    // lane j32 flushes pair entry j32 = k (k + 1) / 2 + i (entries 32..35 = (4..7, 7) go with lanes 0..3)
    const int pair_k = (j32 >= 1) + (j32 >= 3) + (j32 >= 6) + (j32 >= 10) + (j32 >= 15) + (j32 >= 21) + (j32 >= 28);
    const int pair_i = j32 - pair_k * (pair_k + 1) / 2;
    double *mxs = reinterpret_cast<double *>(area + A_MISC) + 8, *sms = mxs + 4;
    const uint32_t *stA = reinterpret_cast<const uint32_t *>(smem + T_ST) + 4 * j32;
    double *PeA = reinterpret_cast<double *>(Pb + O_E) + j32;
    const double *swA = reinterpret_cast<const double *>(smem + T_SW) + j32;
    const double *ewA = reinterpret_cast<const double *>(smem + T_EW) + j32;

    // Global addresses: a wave-uniform base per output (the tile's first row) + a 32-bit byte offset per lane.  Rows
    // beyond N shadow the shard's last row: they recompute and rewrite exactly its values.
    const int rows_left = (int)(N - m0 < TILE_ROWS ? N - m0 : TILE_ROWS);        // >= 1
    const char *yn_t = reinterpret_cast<const char *>(ynorm2 + m0);
    const char *ymu_t = ymu ? reinterpret_cast<const char *>(ymu + m0) : nullptr;
    char *cand_t = reinterpret_cast<char *>(cand + m0 * HP);
    char *out_t = reinterpret_cast<char *>(logpj + m0 * ldl);
    const char *gram_b = reinterpret_cast<const char *>(gram);
    const uint32_t ldl8 = (uint32_t)ldl * 8u, H8 = (uint32_t)H * 8u;
    // M-step statistics (MSTATS): E[s] rows, the candidates' second-moment block -> Wq, column sums, scalars
    char *exp_t = MSTATS ? reinterpret_cast<char *>(expect + m0 * lde) : nullptr;
    double *wq = MSTATS ? stats + pm_bsc_stats_offset_wq_dev(H, Dstats) : nullptr;
    double *t_mus = reinterpret_cast<double *>(smem + T_MUS);
    const uint32_t *t_tab = reinterpret_cast<const uint32_t *>(smem + T_TAB);
    double m_sig = 0.0, m_fs = 0.0, m_cnt = 0.0;        // per-lane partial sums of the scalar statistics
    // DEFERRED statistics (round 6; a data-truncation step, bsc_et.py:247-258: which datapoints count is known only once
    // every log-evidence of every rank is): the pass accumulates NOTHING -- it leaves each datapoint's pair block and
    // sum q e as a record of PM_BSC_DEFER_LD doubles beside its non-zero list, and bsc_defer_apply_kernel adds the records of
    // the datapoints above the cut afterwards (320 B per datapoint instead of a second pass over 3.3 KB of log-joints).
    const bool dfr = MSTATS && defer != nullptr;
    char *rec_t = dfr ? reinterpret_cast<char *>(defer + m0 * PM_BSC_DEFER_LD) : nullptr;

    // scores of pass 0 -> LDS (TAIL: the wavefronts holding the K-sums write all 16 datapoints' rows)
    if (TAIL) {
        if (rg == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    reinterpret_cast<double *>(smem + T_AREAS + (fk + 4 * r) * AREA_BYTES + A_ROW)[half * 128 + j16 + 16 * i] =
                        acc[i][r];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NJ; ++i) rowW[16 * i] = acc[i][0];
    }
    F8_STAMP(2);

#pragma unroll
    for (int r = 0; r < NPASS; ++r) {
        int lrow = TAIL ? slot : 16 * rg + fkM + 4 * r;
        const bool live = lrow < rows_left;
        lrow = live ? lrow : rows_left - 1;
        double yn = *reinterpret_cast<const double *>(yn_t + (uint32_t)lrow * 8u);
        if (ymu_t) yn = yn - 2.0 * *reinterpret_cast<const double *>(ymu_t + (uint32_t)lrow * 8u) + P.mu_sqnorm;
        F8_ESTAMP(r * 8 + 0);
        if (!(PM_F8_SKIP & 16) || r == 0) lds_barrier();      // the scores of pass r (and, r = 0, the tables) are in LDS
        F8_ESTAMP(r * 8 + 1);
        double a[NJ];
#pragma unroll
        for (int i = 0; i < NJ; ++i) a[i] = rowR[32 * i];

        // ---------------- select_Hprimes (bsc_et.py:98-115) ---------------------------------------------------------
        int myc = 0;
        if (PM_F8_SKIP & 1) {
            myc = (j32 * 29 + (int)a[0]) & 255;
        } else if (sel) {
            double key[NJ];
            // scores are finite unless the datapoint or W is not
            bool odd = !(yn < 1.0e150);
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                const int h = j32 + 32 * i;
                const double x = a[i] * swA[32 * i];
                const uint32_t lo = (uint32_t)__double2loint(x), hi = (uint32_t)__double2hiint(x);
                const uint32_t flip = (uint32_t)((int32_t)hi >> 31);
                const uint32_t klo = (lo & ~0x3FFu) | (((uint32_t)h ^ flip) & 0x3FFu);
                key[i] = (FULL || h < H) ? __hiloint2double((int)hi, (int)klo) : -INFINITY;
                odd |= !(fabs(x) < 1.0e300);
            }
            if (__any(odd)) {   // rare: redo the keys of non-finite scores (see row_select)
#pragma unroll
                for (int i = 0; i < NJ; ++i) {
                    const int h = j32 + 32 * i;
                    const double x = a[i] * swA[32 * i];
                    uint64_t b = (uint64_t)__double_as_longlong(x);
                    if (x != x) b = 0xFFEFFFFFFFFFFC00ull;
                    else if ((b & 0x7FF0000000000000ull) == 0x7FF0000000000000ull)
                        b = (b & 0x8000000000000000ull) | 0x7FEFFFFFFFFFF800ull;
                    const uint64_t code = (uint64_t)h;
                    b = (b & ~0x3FFull) | ((b >> 63) ? 0x3FFull - code : code);
                    if (FULL || h < H) key[i] = __longlong_as_double((long long)b);
                }
            }
            sort_desc<NJ>(key);
#pragma unroll
            for (int t = 0; t < NJ; ++t) listA[32 * t] = key[t];
            wave_lds_sync16();
            const double *head_p = listA;
#pragma unroll
            for (int q = 0; q < HP; ++q) {
                const double head = *head_p;
                // (16-wavefront workgroup: the passes run with the matrix pipe idle -- latency is their price, DPP moves;
                // beside a K-loop every VALU instruction costs MFMA time -- the LDS crossbar)
                const double m = W16 ? half_max_dpp(head) : half_max_f64(head);
                win[q] = m;                          // the same value from all 32 lanes
                head_p += (head == m) ? 32 : 0;
            }
            wave_lds_sync16();
            if (j32 < HP) {
                const uint64_t mb = (uint64_t)__double_as_longlong(win[HP - 1 - j32]);    // ascending: best last
                const int code = (int)(mb & 0x3FFull);
                myc = (mb >> 63) ? 0x3FF - code : code;
                *reinterpret_cast<int32_t *>(cand_t + (uint32_t)(lrow * HP + j32) * 4u) = myc;
            }
        } else if (j32 < HP) {
            myc = *reinterpret_cast<const int32_t *>(cand_t + (uint32_t)(lrow * HP + j32) * 4u);
        }
        F8_ESTAMP(r * 8 + 2);
        // the candidates' scores, then the row areas are free for the next pass's scores
        double ac = 0.0;
        if (j32 < HP) {
            ac = reinterpret_cast<const double *>(area + A_ROW)[myc];
            cl[j32] = myc;
        } else if (HP < PH && j32 < PH) {
            cl[j32] = 0;             // (an unused position: a valid latent for the Gram fetch below, never part of a state)
        }
        // Gram block of the candidates, requested NOW (an L2 round trip that the barrier, the next pass's score rows and the
        // singleton log-joints below cover): lane j32 fetches G[c_i, c_k] and G[c_(i+4), c_k], i = j32 >> 3, k = j32 & 7
        double G0 = 0.0, G1 = 0.0;
        if (est) {
            wave_lds_sync16();
            const uint32_t ck8 = (uint32_t)cl[j32 & 7] * 8u;
            const uint32_t ci0 = (uint32_t)cl[j32 >> 3], ci1 = (uint32_t)cl[4 + (j32 >> 3)];
            G0 = *reinterpret_cast<const double *>(gram_b + (ci0 * H8 + ck8));
            G1 = *reinterpret_cast<const double *>(gram_b + (ci1 * H8 + ck8));
        }
        if (!TAIL && !(PM_F8_SKIP & 16)) lds_barrier();
        F8_ESTAMP(r * 8 + 3);
        if (!TAIL && r + 1 < 4) {
#pragma unroll
            for (int i = 0; i < NJ; ++i) rowW[16 * i] = acc[i][r + 1 < 4 ? r + 1 : 3];
        }
        if (!est) continue;

        // ---------------- E_step (bsc_et.py:119-192) -----------------------------------------------------------------
        double wmuc = 0.0;
        if (wmu && j32 < HP) wmuc = wmu[myc];
        // singleton log-joints meanwhile: f_h = prior + ecoef (|W_h|^2 - 2 a_h + |y|^2)
#if PM_F8_SKIP & 128
        char *out = out_t + ((uint32_t)(lrow & 1) * ldl8 + (uint32_t)j32 * 8u);  // (timing only: every row into rows 0/1)
#else
        char *out = out_t + ((uint32_t)lrow * ldl8 + (uint32_t)j32 * 8u);        // &logpj[n, j32]
#endif
        const double f0 = ecoef * yn;
        double mx = f0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int h = j32 + 32 * i;
            double f = -1.0e300;     // no latent: never the maximum, never counted
            if (FULL || h < H) f = fma(m2e, a[i], ewA[32 * i]) + f0;
            a[i] = f;
            mx = vmax64(mx, f);
        }
        if (!(PM_F8_SKIP & 4)) {
#pragma unroll
            for (int i = 0; i < NJ; ++i)
                if (FULL || j32 + 32 * i < H) F8_STORE(out + 8 * (1 + 32 * i), a[i]);
            if (j32 == 0) F8_STORE(out, f0);
        }
        Pm[9 + j32] = G0;
        Pm[9 + 32 + j32] = G1;
        wave_lds_sync16();
        if (j32 < HP) Pm[1 + j32] = Pm[9 + (PH + 1) * j32] - 2.0 * (ac - wmuc);     // d_k = G_kk - 2 a_k
        if (j32 == 0) Pm[0] = 0.0;
        wave_lds_sync16();
        F8_ESTAMP(r * 8 + 4);
        // multi-cause states by size: e(s) = e(parent) + d_k + 2 (G terms)
        char *outS = out + 8 * (1 + H);
        double fs[SS::iters(2) + SS::iters(3) + (GAMMA >= 4 ? SS::iters(4) : 0)];
        int nf = 0;
#pragma unroll
        for (int g = 2; g <= GAMMA; ++g) {
            const double pg = ppil * (double)g;
#pragma unroll
            for (int k = 0; k < SS::iters(g); ++k) {
                const int s0 = SS::off(g) + 32 * k;                     // + j32 = this lane's state
                double f = -1.0e300;
                if (!(PM_F8_SKIP & 2) && (32 * (k + 1) <= SS::cnt(g) || j32 < SS::cnt(g) - 32 * k)) {
                    const uint32_t *t = stA + 4 * s0;
                    const uint32_t t01 = t[0], t23 = t[1], t4 = t[2];
                    const double e = (*reinterpret_cast<const double *>(Pb + (t01 & 0xFFFFu)) +
                                      *reinterpret_cast<const double *>(Pb + (t01 >> 16))) +
                                     2.0 * ((*reinterpret_cast<const double *>(Pb + (t23 & 0xFFFFu)) +
                                             *reinterpret_cast<const double *>(Pb + (t23 >> 16))) +
                                            *reinterpret_cast<const double *>(Pb + t4));
                    PeA[s0] = e;
                    f = fma(ecoef, yn + e, pg);
                    if (!(PM_F8_SKIP & 4)) F8_STORE(outS + 8 * s0, f);
                }
                fs[nf++] = f;
                mx = vmax64(mx, f);
            }
            wave_lds_sync16();
        }
        if (!lse) continue;
        if (PM_F8_SKIP & 8) {
            if (mx == 1.2345e300) lse[0] = mx;
            continue;
        }
        F8_ESTAMP(r * 8 + 5);
        // ---------------- log-sum-exp: only terms within exp(-37) of the largest are evaluated ----------------------
        mx = half_max_f64(mx);
        const double thr = mx + NEGLIGIBLE;
        double sum = 0.0, qe = 0.0;       // qe: sum of exp(f - mx) (f - prior) = ecoef sum of exp(.) e   (MSTATS)
        {
            const bool need = (j32 == 0) && f0 > thr;
            if (__any(need)) {
                sum = need ? exp_lds(f0 - mx, expE, expC) : 0.0;
                if (MSTATS) qe = sum * f0;
            }
        }
#pragma unroll
        for (int i = 0; i < NJ; ++i) {      // MSTATS: a[i] becomes exp(f - mx) of a term that counts, else 0
            const bool need = a[i] > thr;
            double ex = 0.0;
            if (__any(need)) {
                ex = need ? exp_lds(a[i] - mx, expE, expC) : 0.0;
                sum += ex;
                if (MSTATS) qe = need ? fma(ex, a[i] - ppil, qe) : qe;
            }
            if (MSTATS) a[i] = ex;
        }
        {
            int q = 0;
#pragma unroll
            for (int g = 2; g <= GAMMA; ++g) {
#pragma unroll
                for (int k = 0; k < SS::iters(g); ++k, ++q) {
                    const bool need = fs[q] > thr;
                    double ex = 0.0;
                    if (__any(need)) {
                        ex = need ? exp_lds(fs[q] - mx, expE, expC) : 0.0;
                        sum += ex;
                        if (MSTATS) qe = need ? fma(ex, fs[q] - ppil * (double)g, qe) : qe;
                    }
                    if (MSTATS) fs[q] = ex;
                }
            }
        }
        sum = half_sum_f64(sum);
        F8_ESTAMP(r * 8 + 6);
        if (j32 == 0) {
            mxs[r] = mx;
            sms[r] = sum;
        }
        if (MSTATS) {
            // posterior weights q = exp(f - mx) / sum (bsc_et.py:271-272) and what the M-step takes from them
            // (bsc_et.py:334-366, 395-415); rows beyond N (shadows of the last row) contribute nothing
            double inv = __builtin_amdgcn_rcp(sum);
            inv = fma(fma(-sum, inv, 1.0), inv, inv);
            inv = fma(fma(-sum, inv, 1.0), inv, inv);
            if (dfr) {
                const double qrow = half_sum_f64(qe * inv);
                if (j32 == 0 && live)
                    *reinterpret_cast<double *>(rec_t + ((uint32_t)lrow * (PM_BSC_DEFER_LD * 8u) + PAIR_ENTRIES * 8u)) = qrow;
            } else if (live) {
                m_sig += qe * inv;
            }
            // add[h]: the multi-cause states' share of E[s_h], gathered per candidate in LDS (P is free by now)
#pragma unroll
            for (int i = 0; i < NJ; ++i) Pm[j32 + 32 * i] = 0.0;
            // PAIRLDS: the second moments of the candidate pairs are gathered in LDS as well -- entry k (k + 1) / 2 + i
            // for positions i <= k -- and go to Wq as ONE predicated atomic instruction per datapoint.  Every
            // multi-cause state that carries weight used to issue its 3 / 6 / 10 global atomics itself (a cl[] lookup,
            // a min / max and a 64-bit address each): 0.084 ms of the 1.77 ms pass on the parameters of a running EM
            // loop (scratch/em_estep_time.py; per-XCD copies of Wq changed nothing, so it was the issue cost).
            if (PAIRLDS) {
                Bp[j32] = 0.0;
                if (j32 < PAIR_ENTRIES - 32) Bp[32 + j32] = 0.0;
            }
            wave_lds_sync16();
            {
                int q = 0;
#pragma unroll
                for (int g = 2; g <= GAMMA; ++g) {
#pragma unroll
                    for (int k = 0; k < SS::iters(g); ++k, ++q) {
                        const double ex = fs[q];
                        if (__any(ex != 0.0)) {
                            if (ex != 0.0) {
                                const double w = ex * inv;
                                unsigned mi = t_tab[SS::off(g) + 32 * k + j32] & 0xFFFFu;
                                while (mi) {     // E[s_i s_k] += weight for every pair i <= k of the state
                                    const int i = __builtin_ctz(mi);
                                    mi &= mi - 1;
                                    const int ci = cl[i];
                                    atomicAdd(&Pm[ci], ex);
                                    // (the diagonal of the second moments is not accumulated: E[s_h^2] = E[s_h], the
                                    // kernel leaves qdiag = mus and a zero diagonal in the Wq block)
                                    if (PAIRLDS) {
                                        unsigned mk = mi;
                                        while (mk) {
                                            const int kk = __builtin_ctz(mk);      // kk > i
                                            mk &= mk - 1;
                                            atomicAdd(&Bp[kk * (kk + 1) / 2 + i], w);
                                        }
                                    } else if (!(PM_F8_SKIP & (64 | 256)) && live) {
                                        unsigned mk = mi;
                                        while (mk) {
                                            const int kk = __builtin_ctz(mk);
                                            mk &= mk - 1;
                                            const int ck = cl[kk];
                                            const int lo = ci < ck ? ci : ck, hi = ci < ck ? ck : ci;
                                            pm_atomic_add(wq + (int64_t)lo * H + hi, PM_Q(w, 0));
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            wave_lds_sync16();
            if (PAIRLDS) {
                // lane j32 flushes entry j32 = (pair_i, pair_k) and, lanes 0..3, entry 32 + j32 = (4 + j32, 7)
                const double v = Bp[j32];
                if (dfr) {
                    char *rec = rec_t + ((uint32_t)lrow * (PM_BSC_DEFER_LD * 8u) + (uint32_t)j32 * 8u);
                    if (live) {
                        *reinterpret_cast<double *>(rec) = v;
                        if (j32 < PAIR_ENTRIES - 32) *reinterpret_cast<double *>(rec + 256) = Bp[32 + j32];
                    }
                } else {
                    const int ci = cl[pair_i], ck = cl[pair_k];
                    const int lo = ci < ck ? ci : ck, hi = ci < ck ? ck : ci;
                    if (!(PM_F8_SKIP & (64 | 256)) && live && v != 0.0) pm_atomic_add(wq + (int64_t)lo * H + hi, PM_Q(v, 0));
                    if (j32 < PAIR_ENTRIES - 32) {
                        const double v2 = Bp[32 + j32];
                        const int c2 = cl[4 + j32], c7 = cl[7];
                        const int lo2 = c2 < c7 ? c2 : c7, hi2 = c2 < c7 ? c7 : c2;
                        if (!(PM_F8_SKIP & (64 | 256)) && live && v2 != 0.0) pm_atomic_add(wq + (int64_t)lo2 * H + hi2, PM_Q(v2, 0));
                    }
                }
            }
            char *erow = exp_t + ((uint32_t)lrow * (uint32_t)lde * 8u + (uint32_t)j32 * 8u);
            // the row's non-zeros as a list (pm_bsc_wp_sparse_f64 multiplies them into the data): up to
            // PM_BSC_NZ_MAX (index, value) pairs in the order the lanes hold them, unused index slots 0xFFFF; a longer
            // row counts in the statistics' overflow scalar and the dense product runs instead.  With lists the DENSE row
            // is stored only when its list overflowed (slot 0 of the list then reads PM_BSC_NZ_OVERFLOW): a row with a
            // complete list IS its list, and N x H doubles of write traffic per step (410 MB at config 2) stay at home
            // [round 5; pm_bsc_expand_lists_gated_f64 rebuilds the dense rows of listed datapoints if the dense product
            // has to run after all].
            const bool lists = nz_idx != nullptr;
            uint16_t *nzi = nz_idx + (m0 + lrow) * PM_BSC_NZ_MAX;
            double *nzv = nz_val + (m0 + lrow) * PM_BSC_NZ_MAX;
            uint32_t nzn = 0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                const int h = j32 + 32 * i;
                double v = 0.0;
                if (FULL || h < H) {
                    v = (a[i] + Pm[h]) * inv;
                    if (!(PM_F8_SKIP & 32) && !lists) *reinterpret_cast<double *>(erow + 256 * i) = v;
                    if (!(PM_F8_SKIP & 64) && live && !dfr && __any(v != 0.0)) {
                        if (v != 0.0) atomicAdd(&t_mus[h], PM_Q(v, 0));       // (the one LDS accumulator all wavefronts share)
                    }
                }
                if (lists) {
                    const uint64_t bal = __ballot(v != 0.0);
                    const uint32_t mine = dsel ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                    if (mine) {
                        const uint32_t pos = nzn + __popc(mine & ((1u << j32) - 1u));
                        if (v != 0.0 && pos < PM_BSC_NZ_MAX && live) {
                            nzi[pos] = (uint16_t)h;
                            nzv[pos] = v;
                        }
                        nzn += __popc(mine);
                    }
                }
            }
            if (lists && nzn > PM_BSC_NZ_MAX && !(PM_F8_SKIP & 32)) {      // (uniform over the datapoint's 32 lanes)
#pragma unroll
                for (int i = 0; i < NJ; ++i) {
                    const int h = j32 + 32 * i;
                    if (FULL || h < H) *reinterpret_cast<double *>(erow + 256 * i) = (a[i] + Pm[h]) * inv;
                }
            }
            if (lists && live) {
                if (j32 < PM_BSC_NZ_MAX && (uint32_t)j32 >= nzn) nzi[j32] = 0xFFFFu;
                if (j32 == 0 && nzn > PM_BSC_NZ_MAX) {
                    nzi[0] = PM_BSC_NZ_OVERFLOW;       // (behind the slot's own entry: same wavefront, same address, in order)
                    pm_atomic_add(stats + pm_bsc_stats_offset_scalars_dev(H, Dstats) + 3, 1.0);
                }
            }
            wave_lds_sync16();       // P is the next pass's list area
        }
    }
    if (est && lse) {
        // the four passes' log-evidences at once: lane r of the half-wavefront takes pass r
        wave_lds_sync16();
        const int rr = TAIL ? 0 : (j32 & 3);
        const double lse_n = mxs[rr] + log_ge1(sms[rr]);
        const int lrow = TAIL ? slot : 16 * rg + fkM + 4 * rr;
        if (j32 < NPASS && lrow < rows_left) {
            lse[m0 + lrow] = lse_n;
            if (!dfr) {
                m_fs += lse_n;
                m_cnt += 1.0;
            }
        }
    }
    if (MSTATS) {
        // column sums of E[s] and the scalar statistics of this tile -> packed statistics buffer
        const double sig = pm_wave_sum(m_sig) / ecoef, fsum = pm_wave_sum(m_fs), cnt = pm_wave_sum(m_cnt);
        double *red = reinterpret_cast<double *>(smem + T_EXPC) ;     // (the exponential's constants are done with)
        lds_barrier();                       // every wavefront's LDS atomics into mus are done; nobody needs expC
        if (lane == 0) {
            red[wave] = sig;
            red[16 + wave] = fsum;
            red[32 + wave] = cnt;
        }
        lds_barrier();
        double *sc = stats + pm_bsc_stats_offset_scalars_dev(H, Dstats);
        if (tid < 3) {
            double v = 0.0;
            for (int w = 0; w < NWAVES; ++w) v += red[16 * tid + w];
            if (v != 0.0) pm_atomic_add(sc + tid, PM_Q(v, tid == 0 ? 1 : tid == 1 ? 2 : 0));    // sum q e | sum lse | count
        }
        double *g_mus = stats + pm_bsc_stats_offset_mus_dev(H, Dstats);
        double *g_qd = stats + pm_bsc_stats_offset_qdiag_dev(H, Dstats);
        if (tid < H) {
            const double v = t_mus[tid];
            if (v != 0.0) {
                pm_atomic_add(g_mus + tid, v);
                pm_atomic_add(g_qd + tid, v);      // diag(Wq) in full: see the state loop
            }
        }
    }
    F8_STAMP(3);
    F8_STAMP(4);
    (void)PeA;
    (void)t_tab;
    (void)t_mus;
    (void)exp_t;
    (void)wq;
    (void)Bp;
    (void)pair_i;
    (void)pair_k;
    (void)rec_t;
}

// ---- deferred statistics of a data-truncation step (round 6) ------------------------------------------------------------
// bsc_et.py:247-258 keeps the N_use datapoints with the largest evidence; which ones is known only after the E-step of
// every rank.  The pass above therefore leaves, per datapoint, the non-zero list of E[s] and a record of PM_BSC_DEFER_LD
// doubles [36 pair entries k (k + 1) / 2 + i of the candidates' second moments | sum_k q_k e_k (times ecoef) | pad]; this
// kernel adds the records of the datapoints with lse >= *cut into `stats` exactly as the pass would have (Wq pair block
// with global atomics, mus = qdiag through an LDS accumulator, the three scalars) and EMPTIES the lists of the others, so
// that the sparse product behind it skips them (a dropped datapoint whose list had overflowed gets a zero dense row).
// Two flat, coalesced sweeps -- one thread per record double, then one per list slot: 100 MB at N = 200k, no dependent
// loads beyond the datapoint's log-evidence (a wavefront per datapoint walking record after record took 0.10 ms).
__global__ __launch_bounds__(256) void bsc_defer_apply_kernel(
    const double *__restrict__ lse, const double *__restrict__ cut_dev, const int32_t *__restrict__ cand,
    const double *__restrict__ rec, uint16_t *__restrict__ nz_idx, const double *__restrict__ nz_val,
    double *__restrict__ expect, int64_t lde, double *__restrict__ stats, int64_t N, int H, int Dstats, double ecoef,
    int Hp) {
    __shared__ double s_mus[256];
    __shared__ double s_red[3][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_mus[tid] = 0.0;
    __syncthreads();
    double cut = cut_dev[0];
    // log(2^-1075): the reference's un-stabilised evidence sums are exactly 0 there and `all_denoms >= 0` keeps every
    // datapoint (bsc_et.py:253)
    if (cut < -745.1332191019412) cut = -INFINITY;
    double *wq = stats + pm_bsc_stats_offset_wq_dev(H, Dstats);
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + tid;
    double sig = 0.0, fs = 0.0, cnt = 0.0;
    // (a) the records: entry e = k (k + 1) / 2 + i < 36 is the pair (i, k) of candidate positions, i <= k (diagonal
    // entries are zero: E[s_h^2] = E[s_h] goes through qdiag); entry 36 the datapoint's sum q e; the thread of pad entry 37
    // takes the datapoint's log-evidence and count
    for (int64_t t = t0; t < N * PM_BSC_DEFER_LD; t += stride) {
        const int64_t n = t / PM_BSC_DEFER_LD;
        const int e = (int)(t - n * PM_BSC_DEFER_LD);
        const int ec = e < PAIR_ENTRIES ? e : 0;
        int pk = (ec >= 1) + (ec >= 3) + (ec >= 6) + (ec >= 10) + (ec >= 15) + (ec >= 21) + (ec >= 28);
        int pi = ec - pk * (pk + 1) / 2;
        pk = pk < Hp ? pk : 0;             // (positions H' .. 7 of the layout carry zeros: any valid candidate will do for the load)
        pi = pi < Hp ? pi : 0;
        // (all four loads leave before any of them is looked at: a dropped datapoint's record is read for nothing, but no
        // load waits for another)
        const double l = lse[n];
        const double v = rec[t];
        const int ci = cand[n * Hp + pi], ck = cand[n * Hp + pk];
        if (e > PAIR_ENTRIES + 1 || !(l >= cut)) continue;
        if (e < PAIR_ENTRIES) {
            if (v != 0.0) {
                const int lo = ci < ck ? ci : ck, hi = ci < ck ? ck : ci;
                pm_atomic_add(wq + (int64_t)lo * H + hi, PM_Q(v, 0));
            }
        } else if (e == PAIR_ENTRIES) {
            sig += v;
        } else {
            fs += l;
            cnt += 1.0;
        }
    }
    // (b) the lists: slot j of datapoint n (the sixteen slots of a datapoint sit in one wavefront: all of them read slot 0
    // before any of them rewrites it)
    for (int64_t t = t0; t < N * PM_BSC_NZ_MAX; t += stride) {
        const int64_t n = t / PM_BSC_NZ_MAX;
        const int j = (int)(t - n * PM_BSC_NZ_MAX);
        const double l = lse[n];
        const bool over = nz_idx[n * PM_BSC_NZ_MAX] == PM_BSC_NZ_OVERFLOW;
        const unsigned h = nz_idx[t];
        const double v = nz_val[t];
        if (!(l >= cut)) {
            nz_idx[t] = 0xFFFFu;
            if (over)
                for (int hh = j; hh < H; hh += PM_BSC_NZ_MAX) expect[n * lde + hh] = 0.0;
        } else if (!over) {
            if (h != 0xFFFFu) atomicAdd(&s_mus[h], PM_Q(v, 0));
        } else {
            for (int hh = j; hh < H; hh += PM_BSC_NZ_MAX) {
                const double w = expect[n * lde + hh];
                if (w != 0.0) atomicAdd(&s_mus[hh], PM_Q(w, 0));
            }
        }
    }
    sig = pm_wave_sum(sig);
    fs = pm_wave_sum(fs);
    cnt = pm_wave_sum(cnt);
    if (lane == 0) {
        s_red[0][wave] = sig;
        s_red[1][wave] = fs;
        s_red[2][wave] = cnt;
    }
    __syncthreads();
    double *sc = stats + pm_bsc_stats_offset_scalars_dev(H, Dstats);
    if (tid < 3) {
        double v = (s_red[tid][0] + s_red[tid][1]) + (s_red[tid][2] + s_red[tid][3]);
        if (tid == 0) v /= ecoef;
        if (v != 0.0) pm_atomic_add(sc + tid, PM_Q(v, tid == 0 ? 1 : tid == 1 ? 2 : 0));    // sum q e | sum lse | count
    }
    if (tid < H) {
        const double v = s_mus[tid];
        if (v != 0.0) {
            pm_atomic_add(stats + pm_bsc_stats_offset_mus_dev(H, Dstats) + tid, v);
            pm_atomic_add(stats + pm_bsc_stats_offset_qdiag_dev(H, Dstats) + tid, v);
        }
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace pm_fused8

using namespace pm_fused8;

#ifdef PM_F8_STAMPS
extern "C" int pm_f8_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pm_f8_stamps), sizeof(unsigned long long) * 8 * n);
}
extern "C" int pm_f8_read_estamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pm_f8_estamps), sizeof(unsigned long long) * 16 * 32);
}
#endif

// The kernel covers config 2's shape class: 128 < H <= 256 latents, H' = 5 .. 8 candidates, the complete state set of sizes
// 2 .. gamma for gamma in {3, 4} (what generate_state_matrix builds: S = 84 / 154), D a multiple of 8.  Everything else
// takes the 4-wavefront kernel of bsc_fused.hip or the two-kernel path.
// number of multi-cause states of the complete state set of sizes 2 .. gamma over hp candidate positions (0: not an instance)
static int fused8_states(int64_t hp, int64_t gamma) {
    if (hp < 5 || hp > 8 || gamma < 3 || gamma > 4) return 0;
    int s = 0, c = (int)hp;
    for (int g = 2; g <= (int)gamma; ++g) {
        c = c * ((int)hp - g + 1) / g;              // C(hp, g) from C(hp, g - 1)
        s += c;
    }
    return s;
}

extern "C" int pm_bsc_fused8_supported(int64_t H, int64_t D, int64_t Hprime, int64_t S) {
    if (H <= 128 || H > 256 || D < DK || D % DK != 0 || Hprime < 5 || Hprime > 8) return 0;
    return (S > 0 && (S == fused8_states(Hprime, 3) || S == fused8_states(Hprime, 4))) ? 1 : 0;
}

// pm_bsc_estep_fused8_f64 takes a WHOLE shard in one call -- whole rounds of 128-row tiles plus the TAIL kernel for a
// ragged remainder -- and also produces the M-step statistics when asked.
// Leading rows of an N-row shard the main launch takes (the TAIL launch takes the rest): whole rounds of resident
// workgroups (one per CU); everything when the ragged round is mostly full (>= 70 %) or too large for two rounds of 16-row
// workgroups.
extern "C" int64_t pm_bsc_fused8_main_rows(int64_t N, int64_t D) {
    if (N <= 0) return 0;
    if (D % TAIL_DK != 0) return N;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    const int64_t rnd = (int64_t)cus * 128;
    const int64_t main_rows = N / rnd * rnd, rest = N - main_rows;
    if (rest * 10 >= rnd * 7 || rest > 2 * (int64_t)cus * TAIL_ROWS) return N;
    return main_rows;
}

extern "C" int pm_bsc_fused8_whole_shard(int64_t H, int64_t Hprime, int64_t gamma, int64_t S) {
    if (H <= 128 || H > 256) return 0;
    return (S > 0 && S == fused8_states(Hprime, gamma)) ? 1 : 0;
}

extern "C" int pm_bsc_estep_fused8_f64(const double *Y, int64_t ldy, const double *Wt, int64_t ldw, const double *gram,
                                       const double *ynorm2, const double *wmu, const double *ymu,
                                       const uint16_t *state_masks, const uint16_t *state_parents,
                                       const int32_t *size_offsets_host, int64_t S, int64_t gamma,
                                       const pm_bsc_estep_params *params_host, int64_t N, int64_t D, int64_t H,
                                       int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl, double *lse,
                                       double *expect, int64_t lde, double *stats, int64_t D_stats, int part,
                                       void *stream) {
    return pm_bsc_estep_fused8_nz_f64(Y, ldy, Wt, ldw, gram, ynorm2, wmu, ymu, state_masks, state_parents,
                                      size_offsets_host, S, gamma, params_host, N, D, H, Hprime, mode, cand, logpj, ldl, lse,
                                      expect, lde, stats, D_stats, nullptr, nullptr, part, stream);
}

extern "C" int pm_bsc_estep_fused8_nz_f64(const double *Y, int64_t ldy, const double *Wt, int64_t ldw,
                                          const double *gram, const double *ynorm2, const double *wmu, const double *ymu,
                                          const uint16_t *state_masks, const uint16_t *state_parents,
                                          const int32_t *size_offsets_host, int64_t S, int64_t gamma,
                                          const pm_bsc_estep_params *params_host, int64_t N, int64_t D, int64_t H,
                                          int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl,
                                          double *lse, double *expect, int64_t lde, double *stats, int64_t D_stats,
                                          uint16_t *nz_idx, double *nz_val, int part, void *stream) {
    return pm_bsc_estep_fused8_defer_f64(Y, ldy, Wt, ldw, gram, ynorm2, wmu, ymu, state_masks, state_parents,
                                         size_offsets_host, S, gamma, params_host, N, D, H, Hprime, mode, cand, logpj, ldl,
                                         lse, expect, lde, stats, D_stats, nz_idx, nz_val, nullptr, part, stream);
}

extern "C" int pm_bsc_defer_apply_f64(const double *lse, const double *cut, const int32_t *cand, const double *records,
                                      uint16_t *nz_idx, const double *nz_val, double *expect, int64_t lde, double *stats,
                                      const pm_bsc_estep_params *params_host, int64_t N, int64_t H, int64_t D,
                                      int64_t Hprime, void *stream) {
    if (!lse || !cut || !cand || !records || !nz_idx || !nz_val || !expect || !stats || !params_host || N < 0 || lde < H)
        return PM_EINVAL;
    if (H <= 0 || H > 256 || Hprime < 5 || Hprime > 8 || D <= 0 || params_host->ecoef == 0.0) return PM_ERANGE;
    if (N == 0) return PM_OK;
    int64_t blocks = (N * PM_BSC_DEFER_LD + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bsc_defer_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), lse,
                       cut, cand, records, nz_idx, nz_val, expect, lde, stats, N, (int)H, (int)D, params_host->ecoef, (int)Hprime);
    return (int)hipGetLastError();
}

extern "C" int pm_bsc_estep_fused8_defer_f64(const double *Y, int64_t ldy, const double *Wt, int64_t ldw,
                                             const double *gram, const double *ynorm2, const double *wmu,
                                             const double *ymu, const uint16_t *state_masks,
                                             const uint16_t *state_parents, const int32_t *size_offsets_host, int64_t S,
                                             int64_t gamma, const pm_bsc_estep_params *params_host, int64_t N, int64_t D,
                                             int64_t H, int64_t Hprime, int mode, int32_t *cand, double *logpj,
                                             int64_t ldl, double *lse, double *expect, int64_t lde, double *stats,
                                             int64_t D_stats, uint16_t *nz_idx, double *nz_val, double *records, int part,
                                             void *stream) {
    if (records && !nz_idx) return PM_EINVAL;
    if ((nz_idx == nullptr) != (nz_val == nullptr) || (nz_idx && !stats)) return PM_EINVAL;
    if (!Y || !Wt || !gram || !ynorm2 || !cand || N < 0 || H <= 0 || D <= 0 || Hprime <= 0 || S < 0 || ldy < D ||
        ldw < D || !(mode & 3) || ((wmu == nullptr) != (ymu == nullptr)) || part < 0 || part > 2)
        return PM_EINVAL;
    if ((mode & 2) && (!params_host || !logpj || ldl < 1 + H + S || gamma < 1 || gamma > Hprime ||
                       (S > 0 && (!state_masks || !state_parents || !size_offsets_host))))
        return PM_EINVAL;
    if (!pm_bsc_fused8_supported(H, D, Hprime, S) || (mode & ~3)) return PM_ERANGE;
    if (stats && (!expect || lde < H || !lse || !(mode & 2) || D_stats <= 0)) return PM_EINVAL;
    if (!aligned16(Y) || !aligned16(Wt) || (ldy % 2) || (ldw % 2)) return PM_EINVAL;
    if (N == 0) return PM_OK;
    pm_bsc_estep_params P = params_host ? *params_host : pm_bsc_estep_params{0, 0, 0, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // H' = 5 .. 8 and the complete state set of sizes 2 .. gamma (what generate_state_matrix builds), in its order
    if (!pm_bsc_fused8_whole_shard(H, Hprime, gamma, S)) return PM_ERANGE;
    if ((mode & 2) && S > 0) {
        int off = 0, c = (int)Hprime;
        for (int g = 2; g <= gamma; ++g) {
            if (size_offsets_host[g - 2] != off) return PM_EINVAL;
            c = c * ((int)Hprime - g + 1) / g;
            off += c;
        }
    }
    const size_t shmem_s = (size_t)LEAN16_LDS_BYTES;      // (>= the ring: 4 stages x 24 KB)
#define PM_LAUNCH8SMTH(HPV, G, F, M, T, GRID, SH, NN, R0)                                                              \
    do {                                                                                                               \
        if (int e = (int)hipFuncSetAttribute(reinterpret_cast<const void *>(bsc_estep_fused8s_kernel<4, HPV, G, F, M, T>), \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SH)))                   \
            return e;                                                                                                  \
        hipLaunchKernelGGL((bsc_estep_fused8s_kernel<4, HPV, G, F, M, T>), dim3((unsigned)(GRID)),                     \
                           dim3((T) ? THREADS : 1024), (SH), s, Y, ldy,                                                \
                           Wt, ldw, (int)D, gram, ynorm2, wmu, ymu, state_masks, state_parents, P, (int64_t)(NN),    \
                           (int)H, mode, cand, logpj, ldl, lse, expect, lde, stats, (int)D_stats, (int64_t)(R0),      \
                           nz_idx, nz_val, records);                                                                   \
    } while (0)
#define PM_LAUNCH8SMT(G, F, M, T, GRID, SH, NN, R0)                                  \
    do {                                                                             \
        if (Hprime == 8) PM_LAUNCH8SMTH(8, G, F, M, T, GRID, SH, NN, R0);            \
        else if (Hprime == 7) PM_LAUNCH8SMTH(7, G, F, M, T, GRID, SH, NN, R0);       \
        else if (Hprime == 6) PM_LAUNCH8SMTH(6, G, F, M, T, GRID, SH, NN, R0);       \
        else PM_LAUNCH8SMTH(5, G, F, M, T, GRID, SH, NN, R0);                        \
    } while (0)
#define PM_LAUNCH8ST(G, F, T, GRID, SH, NN, R0)            \
    do {                                                    \
        if (stats) {                                        \
            PM_LAUNCH8SMT(G, F, true, T, GRID, SH, NN, R0); \
        } else {                                            \
            PM_LAUNCH8SMT(G, F, false, T, GRID, SH, NN, R0);\
        }                                                   \
    } while (0)
#define PM_LAUNCH8SGF(T, GRID, SH, NN, R0)                             \
    do {                                                                \
        if (gamma == 4) {                                               \
            if (H == 256) PM_LAUNCH8ST(4, true, T, GRID, SH, NN, R0);   \
            else PM_LAUNCH8ST(4, false, T, GRID, SH, NN, R0);           \
        } else {                                                        \
            if (H == 256) PM_LAUNCH8ST(3, true, T, GRID, SH, NN, R0);   \
            else PM_LAUNCH8ST(3, false, T, GRID, SH, NN, R0);           \
        }                                                               \
    } while (0)
    {
        // Whole rounds of resident workgroups (one 128-row tile per CU) go to the main launch; a ragged last round that
        // fills most of a round stays with it; a small remainder (up to two rounds of 16-row workgroups) goes to the TAIL
        // kernel, whose workgroups split K four ways; anything in between is cheaper as a partial round of tiles.
        const int64_t main_rows = pm_bsc_fused8_main_rows(N, D);
        const int64_t rest = N - main_rows;
        if (main_rows > 0 && part != 2) PM_LAUNCH8SGF(false, (main_rows + 127) / 128, shmem_s, main_rows, 0);
        const size_t shmem_t = (size_t)TAIL_RING_BYTES > (size_t)LEAN_LDS_BYTES ? (size_t)TAIL_RING_BYTES : (size_t)LEAN_LDS_BYTES;
        if (rest > 0 && part != 1) PM_LAUNCH8SGF(true, (rest + TAIL_ROWS - 1) / TAIL_ROWS, shmem_t, N, main_rows);
    }
#undef PM_LAUNCH8SGF
#undef PM_LAUNCH8ST
#undef PM_LAUNCH8SMT
#undef PM_LAUNCH8SMTH
    return (int)hipGetLastError();
}

PM_DET_SETTER(bsc_fused8)
