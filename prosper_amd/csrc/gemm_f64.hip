// f64 MFMA GEMM kernels for gfx950 (v_mfma_f64_16x16x4_f64).
//
//   gemm_nt : C[M,N]  = A[M,K] . B[N,K]^T   scores a = Y.W^T and Gram G = W.W^T
//   gemm_tn : C[M,N] += A[K,M]^T . B[K,N]   Wp = E[s]^T . Y, split over the datapoint index
//
// Hardware facts these kernels are built on (measured on MI355X, scratch/mfma_peak*.hip):
//   * v_mfma_f64_16x16x4_f64 issues every 64 cycles per SIMD with its accumulator in ARCH VGPRs
//     (77.2 TF/s chip-wide = 98 % of the 78.6 TF/s datasheet peak, clock holds ~2.39 GHz); with
//     the accumulator in AGPRs the same stream runs 2.1x slower (138 cycles).  Accumulators are
//     therefore kept in arch VGPRs (__launch_bounds__(256, 2) => <= 256 VGPRs, no AGPR spill).
//   * between two MFMAs of one wavefront ~56 issue cycles are free: global loads, LDS traffic and
//     address arithmetic placed there cost nothing; only what sits outside the MFMA stream
//     (barrier skew, un-overlapped LDS latency, a partially filled last round of workgroups) does.
//
// Tiling: 128x128 output tile per 256-thread workgroup = 4 wavefronts in a 2x2 grid, each
// wavefront owns (16 MT)x64 = MT x 4 MFMA tiles (MT = 4 for the main grid; MT = 2 / 1 give 64- and
// 32-row tiles for the ragged last round, see launch_nt), K advanced 16 at a time through two LDS
// buffers, two workgroups per CU.  Software pipeline per K-step (4 MFMA k-groups of 4):
//   k-groups 0..2: ds_read the next k-group's fragments, then MT*4 MFMAs on the current ones
//   k-group 3    : write the staged registers of K-step t+1 to the other LDS buffer, barrier,
//                  ds_read K-step t+1's first fragments, issue the global loads of K-step t+2,
//                  THEN the MFMAs of k-group 3 -- so LDS latency after the barrier and the whole
//                  global-load issue run in the shadow of 16 MFMAs (1024 pipe cycles).
//
// Fragment maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l & 15][k = l >> 4]      B: lane l holds B[k = l >> 4][j = l & 15]
//   C/D: 4 f64 per lane, col = l & 15, row = (l >> 4) + 4 * reg
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BN = 128, BK = 16;
constexpr int NT_LD = BK + 2;     // LDS row stride (doubles) of K-contiguous tiles: 144-B rows, so the
                                  // 16 rows x 2 k of one ds_read_b64 half-wave hit distinct bank pairs
constexpr int TN_BM = 128;
constexpr int TN_LD = TN_BM + 16; // LDS row stride of K-strided tiles: consecutive k land 32 banks apart

__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// C = A . B^T, A (M,K) and B (N,K) row-major.  One (32 MT) x 128 tile.
// ---------------------------------------------------------------------------------------------
template <int MT, bool ALIGNED, bool EDGE>
__device__ __forceinline__ void nt_tile(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                        int64_t ldb, double *__restrict__ C, int64_t ldc, int M, int N, int K,
                                        int m0, int n0, double *sm) {
    constexpr int BM = 32 * MT;
    constexpr int STAGE = (BM + BN) * NT_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    const int wm = wave >> 1, wn = wave & 1;

    // staging map: thread -> (row = tid/8 + 32 c, k pair = 2 (tid % 8))
    const int srow = tid >> 3, skc = (tid & 7) * 2;
    d2 ra[MT], rb[4];
    const double *pa = A + (int64_t)(m0 + srow) * lda + skc;
    const double *pb = B + (int64_t)(n0 + srow) * ldb + skc;

    auto gload = [&](int k0) {
        const int k = k0 + skc;
#pragma unroll
        for (int c = 0; c < MT; ++c) {
            const double *p = pa + (int64_t)(32 * c) * lda + k0;
            if (!EDGE) {
                ra[c] = *reinterpret_cast<const d2 *>(p);
            } else {
                d2 v = {0.0, 0.0};
                if (m0 + srow + 32 * c < M) {
                    if (ALIGNED) {
                        if (k < K) v = *reinterpret_cast<const d2 *>(p);
                    } else {
                        if (k < K) v.x = p[0];
                        if (k + 1 < K) v.y = p[1];
                    }
                }
                ra[c] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double *p = pb + (int64_t)(32 * c) * ldb + k0;
            if (!EDGE) {
                rb[c] = *reinterpret_cast<const d2 *>(p);
            } else {
                d2 v = {0.0, 0.0};
                if (n0 + srow + 32 * c < N) {
                    if (ALIGNED) {
                        if (k < K) v = *reinterpret_cast<const d2 *>(p);
                    } else {
                        if (k < K) v.x = p[0];
                        if (k + 1 < K) v.y = p[1];
                    }
                }
                rb[c] = v;
            }
        }
    };
    auto swrite = [&](int buf) {
        double *sa = sm + buf * STAGE + srow * NT_LD + skc;
        double *sb = sa + BM * NT_LD;
#pragma unroll
        for (int c = 0; c < MT; ++c) *reinterpret_cast<d2 *>(sa + 32 * c * NT_LD) = ra[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<d2 *>(sb + 32 * c * NT_LD) = rb[c];
    };

    const int frow = lane & 15, fk = lane >> 4;
    const int a_off = (wm * 16 * MT + frow) * NT_LD + fk;
    const int b_off = BM * NT_LD + (wn * 64 + frow) * NT_LD + fk;
    double fa[2][MT], fb[2][4];
    auto fread = [&](int buf, int kk, int slot) {
        const double *sa = sm + buf * STAGE + a_off + kk * 4;
        const double *sb = sm + buf * STAGE + b_off + kk * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[slot][i] = sa[i * 16 * NT_LD];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[slot][j] = sb[j * 16 * NT_LD];
    };

    d4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = (K + BK - 1) / BK;
    gload(0);
    swrite(0);
    __syncthreads();
    if (nk > 1) gload(BK);
    fread(0, 0, 0);

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                fread(buf, kk + 1, (kk + 1) & 1);
            } else if (t + 1 < nk) {
                swrite(buf ^ 1);
                __syncthreads();  // everyone: fragments of K-step t are in registers, K-step t+1 is in LDS
                fread(buf ^ 1, 0, 0);
                if (t + 2 < nk) gload((t + 2) * BK);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fa[kk & 1][i], fb[kk & 1][j], acc[i][j]);
        }
    }

    // epilogue: lane holds C[row = fk + 4 r][col = frow] of each 16x16 tile
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 16 * MT + i * 16 + fk + 4 * r;
            if (EDGE && row >= M) continue;
            double *crow = C + (int64_t)row * ldc + n0 + wn * 64 + frow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!EDGE || n0 + wn * 64 + j * 16 + frow < N) crow[j * 16] = acc[i][j][r];
            }
        }
    }
}

template <int MT, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(const double *__restrict__ A, int64_t lda,
                                                              const double *__restrict__ B, int64_t ldb,
                                                              double *__restrict__ C, int64_t ldc, int M, int N,
                                                              int K, int tiles_n) {
    constexpr int BM = 32 * MT;
    __shared__ __attribute__((aligned(16))) double sm[2 * (BM + BN) * NT_LD];
    const int bn = blockIdx.x % tiles_n, bm = blockIdx.x / tiles_n;
    const int m0 = bm * BM, n0 = bn * BN;
    const bool edge = !ALIGNED || (m0 + BM > M) || (n0 + BN > N) || (K % BK != 0);
    if (edge)
        nt_tile<MT, ALIGNED, true>(A, lda, B, ldb, C, ldc, M, N, K, m0, n0, sm);
    else
        nt_tile<MT, ALIGNED, false>(A, lda, B, ldb, C, ldc, M, N, K, m0, n0, sm);
}

// ---------------------------------------------------------------------------------------------
// C = A . B^T through an LDS-DMA ring (global_load_lds_dwordx4): the fast path of gemm_nt.
//
// The register-staged kernel above has ONE K-step of lead time between issuing a tile's global
// loads and needing them in LDS; measured on MI355X that is not enough (waiting on loads costs
// 56 -> 72 TF/s).  Here tiles travel HBM/L2 -> LDS directly, no VGPR staging, into a ring of 4
// stages of 8 K-columns, three K-steps ahead of the MFMAs.  128x128 tile, 4 wavefronts (2x2), two
// workgroups per CU (64 KB LDS each).
//
// Stage image: [A rows 0..127 | B rows 0..127] x 8 doubles (64 B per row), 16 chunks of 1 KB; one
// wave-instruction moves one chunk (16 rows x 64 B): lane l writes LDS slot l (16 B) = row l/4,
// position l%4, and fetches k-pair (l%4) ^ ((l/16)%4) of that row -- an XOR swizzle applied on the
// SOURCE address (the LDS side of an LDS-DMA is linear), undone by the fragment reads, which makes
// the 16 rows of a ds_read_b64 half-wave hit 16 distinct bank pairs.
// Per K-step and wavefront: 4 DMA instructions, 8 ds_read_b128, 32 MFMAs, one raw s_barrier placed
// before the last 16 MFMAs (their operands are already in registers), counted s_waitcnt vmcnt.
// Rows/columns past M/N are clamped on load (garbage accumulators that are never stored).
// ---------------------------------------------------------------------------------------------
constexpr int DK = 8, DSTAGES = 4;   // K columns per stage, ring stages (a stage: (BM + 128) x 8 doubles)

#define PM_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// BM: rows of A per tile -- 128, or 64 for the ragged last round of a tall product whose tiles would fill between a quarter
// and 70 % of the resident slots (K too short to split): twice as many workgroups of half the work fill that round (DSC /
// TSC / MCA scores at N = 100k: 782 tiles on 512 slots ran two rounds for 1.53 rounds of work).  A 64-row tile does 16 MFMAs
// per K-step and wavefront on 6 fragment reads instead of 32 on 8.
template <int Q>
__device__ __forceinline__ void nt_wait_vmcnt() {
    if (Q == 12) PM_WAIT_VMCNT(12);
    else if (Q == 9) PM_WAIT_VMCNT(9);
    else if (Q == 8) PM_WAIT_VMCNT(8);
    else if (Q == 6) PM_WAIT_VMCNT(6);
    else if (Q == 4) PM_WAIT_VMCNT(4);
    else if (Q == 3) PM_WAIT_VMCNT(3);
    else PM_WAIT_VMCNT(0);
}

template <bool SPLITK, int BM = 128>  // SPLITK: grid.y K-slices added to a zeroed C with atomics (own kernel name in profiles)
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_dma_kernel(const double *__restrict__ A, int64_t lda,
                                                                  const double *__restrict__ B, int64_t ldb,
                                                                  double *__restrict__ C, int64_t ldc, int M, int N,
                                                                  int K, int tiles_n, int kps, int main_tiles,
                                                                  int rest_kps) {
    // gridDim.y > 1: split-K -- this workgroup covers K-steps [blockIdx.y*kps, +kps) and ADDS its
    // partial tile to C with f64 atomics (C zeroed by the launcher).
    // Fused remainder (SPLITK == false, main_tiles < all tiles): workgroups [0, main_tiles) are whole rounds of
    // un-split tiles; the ones behind them are K-slices of the ragged last round's tiles (rows zeroed by the
    // launcher, atomics), dispatched as the last main round drains instead of in a launch of their own.
    static_assert(BM == 128 || BM == 64, "tile rows");
    constexpr int QA = BM / 64;                    // A chunks (16 rows) this wavefront moves per K-step
    constexpr int AI = BM / 32;                    // 16-row blocks of A per wavefront
    constexpr int ACH = BM / 16;                   // A chunks per stage
    constexpr int DSTAGE_T = (BM + 128) * DK;      // doubles per stage
    constexpr int L = QA + 2;                      // DMA instructions per K-step and wavefront
    __shared__ __attribute__((aligned(1024))) double sm[DSTAGES * DSTAGE_T];

    const int tid = threadIdx.x;
    // the wavefront index is uniform: say so, and the LDS-DMA destinations (M0) become scalar arithmetic
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an XCD and its L2).  With two
    // column tiles per row panel, remap so that both tiles of a panel run back-to-back on ONE XCD: the
    // A panel is then fetched from the fabric once and the second tile hits it in that XCD's L2
    // (speed / traffic only; any placement is correct).
    int tile = blockIdx.x;
    int kt0 = blockIdx.y * kps;
    bool slice = SPLITK;
    if (!SPLITK && tile >= main_tiles) {                       // K-slice of a remainder tile
        const int rest_tiles = ((M + BM - 1) / BM) * tiles_n - main_tiles;
        const int u = tile - main_tiles;
        tile = main_tiles + u % rest_tiles;
        kt0 = (u / rest_tiles) * rest_kps;
        kps = rest_kps;
        slice = true;
    } else if (tiles_n == 2 && ((tile >> 4) + 1) * 16 <= main_tiles) {
        tile = (tile & ~15) + ((tile & 7) << 1) + ((tile >> 3) & 1);
    }
    const int bn = tile % tiles_n, bm = tile / tiles_n;
    const int m0 = bm * BM, n0 = bn * 128;

    // DMA sources: this wavefront moves chunks {wave, wave+4} of A (BM = 64: chunk `wave`) and of B
    const int dr = lane >> 2, dj = (lane & 3) ^ ((lane >> 4) & 3);
    const double *src[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int ra = m0 + 16 * (wave + 4 * q) + dr, rb = n0 + 16 * (wave + 4 * q) + dr;
        ra = ra < M ? ra : M - 1;
        rb = rb < N ? rb : N - 1;
        src[q] = A + (int64_t)ra * lda + 2 * dj;
        src[2 + q] = B + (int64_t)rb * ldb + 2 * dj;
    }
    auto dma = [&](int kt, int stage) {
        double *dst = sm + stage * DSTAGE_T + wave * 128;  // chunk = 128 doubles
        const int k0 = (kt0 + kt) * DK;
        __builtin_amdgcn_global_load_lds(src[0] + k0, dst, 16, 0, 0);
        if (QA == 2) __builtin_amdgcn_global_load_lds(src[1] + k0, dst + 4 * 128, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src[2] + k0, dst + ACH * 128, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src[3] + k0, dst + (ACH + 4) * 128, 16, 0, 0);
    };

    // fragment reads.  The two MFMAs of a K-step may take the 8 K-columns in any order as long as A and B agree:
    // k-group fk of the first MFMA takes column 2 fk, of the second column 2 fk + 1 -- exactly the 16-byte slot
    // (k-pair fk) the DMA wrote, so ONE ds_read_b128 per 16-row block feeds both MFMAs (8 LDS reads per K-step
    // and wavefront instead of 16).  Pair p of row R sits at R*8 + ((p ^ ((R>>2)&3)) << 1).
    const int frow = lane & 15, fk = lane >> 4;
    const int sw = (frow >> 2) & 3;
    const int a_off = (wm * (BM / 2) + frow) * DK + ((fk ^ sw) << 1);
    const int b_off = BM * DK + (wn * 64 + frow) * DK + ((fk ^ sw) << 1);
    typedef double d2 __attribute__((ext_vector_type(2)));
    struct Frag {
        d2 a[AI], b[4];
    };
    auto fread = [&](int stage, Frag &f) {
        const double *sa = sm + stage * DSTAGE_T + a_off;
        const double *sb = sm + stage * DSTAGE_T + b_off;
#pragma unroll
        for (int i = 0; i < AI; ++i) f.a[i] = *reinterpret_cast<const d2 *>(sa + i * 16 * DK);
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = *reinterpret_cast<const d2 *>(sb + j * 16 * DK);
    };

    d4 acc[AI][4];
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    int nk = K / DK - kt0;  // host guarantees K % DK == 0
    nk = nk < kps ? nk : kps;
    if (nk <= 0) return;
#pragma unroll
    for (int t = 0; t < DSTAGES; ++t)
        if (t < nk) dma(t, t);
    // this wavefront's part of K-step 0 has landed
    if (nk >= 4) nt_wait_vmcnt<3 * L>();
    else if (nk == 3) nt_wait_vmcnt<2 * L>();
    else if (nk == 2) nt_wait_vmcnt<L>();
    else nt_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    fread(0, f0);

    // one K-step: 16 MFMAs on the first column of every pair, hand-over (barrier, next step's fragments, refill of
    // the stage just consumed), 16 MFMAs on the second column
    auto kstep = [&](int t, const Frag &cur, Frag &nxt) {
        const int stage = t & (DSTAGES - 1);
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(cur.a[i].x, cur.b[j].x, acc[i][j]);
        if (t + 1 < nk) {
            // my reads of this stage are done (lgkmcnt) and my share of K-step t+1 has landed (vmcnt);
            // after the barrier that holds for every wavefront: stage t may be refilled, t+1 may be read
            const int ahead = nk - t - 2;  // K-steps issued beyond t+1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ahead >= 2) nt_wait_vmcnt<2 * L>();
            else if (ahead == 1) nt_wait_vmcnt<L>();
            else nt_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            fread((t + 1) & (DSTAGES - 1), nxt);
            if (t + DSTAGES < nk) dma(t + DSTAGES, stage);
        }
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(cur.a[i].y, cur.b[j].y, acc[i][j]);
    };
    for (int t = 0; t < nk; t += 2) {
        kstep(t, f0, f1);
        if (t + 1 < nk) kstep(t + 1, f1, f0);
    }

    const bool split = slice;
    const bool interior = (m0 + BM <= M) && (n0 + 128 <= N);
    double *cbase = C + (int64_t)(m0 + wm * (BM / 2) + fk) * ldc + n0 + wn * 64 + frow;
    if (!split && interior) {  // the common case: plain stores, no guards
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) cbase[(int64_t)(i * 16 + 4 * r) * ldc + j * 16] = acc[i][j][r];
        return;
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * (BM / 2) + i * 16 + fk + 4 * r;
            if (row >= M) continue;
            double *crow = cbase + (int64_t)(i * 16 + 4 * r) * ldc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n0 + wn * 64 + j * 16 + frow < N) {
                    if (split) pm_atomic_add(crow + j * 16, PM_Q(acc[i][j][r], 0));
                    else crow[j * 16] = acc[i][j][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C += A^T . B, A (K,M) and B (K,N) row-major; the K range is split over blockIdx.y and the
// partial tiles are added with f64 atomics (global_atomic_add_f64).  Same pipeline as above.
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED, bool EDGE>
__device__ __forceinline__ void tn_tile(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                        int64_t ldb, double *__restrict__ C, int64_t ldc, int M, int N, int64_t kbeg,
                                        int64_t kend, int m0, int n0, double *sm) {
    constexpr int STAGE = 2 * BK * TN_LD;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    const int wm = wave >> 1, wn = wave & 1;

    // staging map: thread -> (k = tid/64 + 4 c, column pair = 2 (tid % 64)), c = 0..3
    const int sk = tid >> 6, scol = (tid & 63) * 2;
    d2 ra[4], rb[4];
    const double *pa = A + m0 + scol;
    const double *pb = B + n0 + scol;

    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t k = k0 + sk + 4 * c;
            if (!EDGE) {
                ra[c] = *reinterpret_cast<const d2 *>(pa + k * lda);
                rb[c] = *reinterpret_cast<const d2 *>(pb + k * ldb);
            } else {
                d2 va = {0.0, 0.0}, vb = {0.0, 0.0};
                if (k < kend) {
                    const double *qa = pa + k * lda, *qb = pb + k * ldb;
                    if (ALIGNED) {
                        if (m0 + scol < M) va = *reinterpret_cast<const d2 *>(qa);
                        if (n0 + scol < N) vb = *reinterpret_cast<const d2 *>(qb);
                    } else {
                        if (m0 + scol < M) va.x = qa[0];
                        if (m0 + scol + 1 < M) va.y = qa[1];
                        if (n0 + scol < N) vb.x = qb[0];
                        if (n0 + scol + 1 < N) vb.y = qb[1];
                    }
                }
                ra[c] = va;
                rb[c] = vb;
            }
        }
    };
    auto swrite = [&](int buf) {
        double *sa = sm + buf * STAGE + sk * TN_LD + scol;
        double *sb = sa + BK * TN_LD;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<d2 *>(sa + 4 * c * TN_LD) = ra[c];
            *reinterpret_cast<d2 *>(sb + 4 * c * TN_LD) = rb[c];
        }
    };

    const int fcol = lane & 15, fk = lane >> 4;
    const int a_off = fk * TN_LD + wm * 64 + fcol;
    const int b_off = BK * TN_LD + fk * TN_LD + wn * 64 + fcol;
    double fa[2][4], fb[2][4];
    auto fread = [&](int buf, int kk, int slot) {
        const double *sa = sm + buf * STAGE + a_off + kk * 4 * TN_LD;
        const double *sb = sm + buf * STAGE + b_off + kk * 4 * TN_LD;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[slot][i] = sa[i * 16];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[slot][j] = sb[j * 16];
    };

    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = (int)((kend - kbeg + BK - 1) / BK);
    gload(kbeg);
    swrite(0);
    __syncthreads();
    if (nk > 1) gload(kbeg + BK);
    fread(0, 0, 0);

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                fread(buf, kk + 1, (kk + 1) & 1);
            } else if (t + 1 < nk) {
                swrite(buf ^ 1);
                __syncthreads();
                fread(buf ^ 1, 0, 0);
                if (t + 2 < nk) gload(kbeg + (int64_t)(t + 2) * BK);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fa[kk & 1][i], fb[kk & 1][j], acc[i][j]);
        }
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + i * 16 + fk + 4 * r;
            if (EDGE && row >= M) continue;
            double *crow = C + (int64_t)row * ldc + n0 + wn * 64 + fcol;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!EDGE || n0 + wn * 64 + j * 16 + fcol < N) pm_atomic_add(crow + j * 16, PM_Q(acc[i][j][r], 0));
            }
        }
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_tn_f64_kernel(const double *__restrict__ A, int64_t lda,
                                                              const double *__restrict__ B, int64_t ldb,
                                                              double *__restrict__ C, int64_t ldc, int M, int N,
                                                              int64_t K, int tiles_n, int64_t k_per_split,
                                                              const double *__restrict__ gate) {
    __shared__ __attribute__((aligned(16))) double sm[2 * 2 * BK * TN_LD];
    if (gate && *gate == 0.0) return;      // (pm_gemm_tn_acc_gated_f64: the device decides whether this product runs)
    // All output tiles of one K-split read the same rows of A and B.  Workgroups are dealt round-robin
    // over the 8 XCDs (linear ids L and L+8 share an XCD), so hand XCD r the splits r, r+8, ... with all
    // their tiles back-to-back: each operand slab then crosses the fabric once per split instead of once
    // per XCD that happens to hold one of its tiles (speed / traffic only).
    int tile = blockIdx.x, split = blockIdx.y;
    if (gridDim.y % 8 == 0) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned r = L & 7u, q = L >> 3;
        tile = (int)(q % gridDim.x);
        split = (int)(r + 8u * (q / gridDim.x));
    }
    const int bn = tile % tiles_n, bm = tile / tiles_n;
    const int m0 = bm * TN_BM, n0 = bn * BN;
    const int64_t kbeg = (int64_t)split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const bool edge = !ALIGNED || (m0 + TN_BM > M) || (n0 + BN > N) || ((kend - kbeg) % BK != 0);
    if (edge)
        tn_tile<ALIGNED, true>(A, lda, B, ldb, C, ldc, M, N, kbeg, kend, m0, n0, sm);
    else
        tn_tile<ALIGNED, false>(A, lda, B, ldb, C, ldc, M, N, kbeg, kend, m0, n0, sm);
}

// ---------------------------------------------------------------------------------------------
// C += A^T . B with the LDS-DMA ring of the NT kernel: interior 128 x 128 tiles, 16-byte aligned
// operands, every K-split a whole number of 8-row steps.  A K-step is 8 rows of A and 8 rows of B; one
// `global_load_lds_dwordx4` moves one 1 KB row segment (64 lanes x 16 B), its LDS destination row is padded
// to TN_LD doubles so that the four k-rows a fragment read touches fall into different banks.
// Same loop as gemm_nt_f64_dma_kernel: 4 DMA instructions, 16 ds_read_b64 and 32 MFMAs per wavefront and
// K-step, one raw s_barrier, counted s_waitcnt vmcnt.  Partial tiles are added with f64 atomics.
// ---------------------------------------------------------------------------------------------
constexpr int TDSTAGE = 2 * DK * TN_LD;  // doubles per stage (18 KB)

// GATHER: the K rows are rows[0 .. *count) of A and B (a device-side list with a device-side length: GSC's dense datapoints,
// gsc_kernels.hip) -- the row index of a DMA instruction is wavefront-uniform anyway, it now comes from a scalar load a
// K-step ahead; positions past the list read `zero_row` (a row of zeros the caller keeps in both operands), so the ragged
// end needs no second kernel; the K-split is derived from *count here (the grid was sized without knowing it).
template <bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_tn_f64_dma_kernel(const double *__restrict__ A, int64_t lda,
                                                                  const double *__restrict__ B, int64_t ldb,
                                                                  double *__restrict__ C, int64_t ldc, int tiles_n,
                                                                  int64_t K, int64_t k_per_split,
                                                                  const double *__restrict__ gate,
                                                                  const int32_t *__restrict__ rows,
                                                                  const int32_t *__restrict__ count, int zero_row) {
    __shared__ __attribute__((aligned(1024))) double sm[DSTAGES * TDSTAGE];
    if (gate && *gate == 0.0) return;
    int cnt = 0;
    if (GATHER) {
        cnt = *count;
        if (cnt > (int)K) cnt = (int)K;                  // (K: capacity of the list)
        const int64_t steps = ((int64_t)cnt + DK - 1) / DK;
        K = steps * DK;
        k_per_split = (steps + gridDim.y - 1) / gridDim.y * DK;
        if (K == 0) return;
    }
    int tile = blockIdx.x, split = blockIdx.y;
    if (gridDim.y % 8 == 0) {   // all tiles of a K-split on one XCD (see gemm_tn_f64_kernel)
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned r = L & 7u, q = L >> 3;
        tile = (int)(q % gridDim.x);
        split = (int)(r + 8u * (q / gridDim.x));
    }
    const int bn = tile % tiles_n, bm = tile / tiles_n;
    const int m0 = bm * TN_BM, n0 = bn * BN;
    const int64_t kbeg = (int64_t)split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg) / DK);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    const int wm = wave >> 1, wn = wave & 1;
    // this wavefront moves rows {wave, wave + 4} of the A slab and of the B slab of every K-step
    const double *srca = A + (GATHER ? 0 : (kbeg + wave) * lda) + m0 + 2 * lane;
    const double *srcb = B + (GATHER ? 0 : (kbeg + wave) * ldb) + n0 + 2 * lane;
    auto row_of = [&](int64_t k) -> int64_t {           // (uniform: a scalar load)
        return (k < cnt) ? (int64_t)rows[k] : (int64_t)zero_row;
    };
    auto dma = [&](int kt, int stage) {
        double *dst = sm + stage * TDSTAGE + wave * TN_LD;
        const int64_t k0 = (int64_t)kt * DK;
        const int64_t r0 = GATHER ? row_of(kbeg + k0 + wave) : k0, r1 = GATHER ? row_of(kbeg + k0 + wave + 4) : k0 + 4;
        __builtin_amdgcn_global_load_lds(srca + r0 * lda, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(srca + r1 * lda, dst + 4 * TN_LD, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(srcb + r0 * ldb, dst + DK * TN_LD, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(srcb + r1 * ldb, dst + (DK + 4) * TN_LD, 16, 0, 0);
    };
    const int fcol = lane & 15, fk = lane >> 4;
    const int a_off = fk * TN_LD + wm * 64 + fcol;
    const int b_off = DK * TN_LD + fk * TN_LD + wn * 64 + fcol;
    double fa[2][4], fb[2][4];
    auto fread = [&](int stage, int kk, int slot) {
        const double *sa = sm + stage * TDSTAGE + a_off + kk * 4 * TN_LD;
        const double *sb = sm + stage * TDSTAGE + b_off + kk * 4 * TN_LD;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[slot][i] = sa[i * 16];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[slot][j] = sb[j * 16];
    };

    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

#pragma unroll
    for (int t = 0; t < DSTAGES; ++t)
        if (t < nk) dma(t, t);
    if (nk >= 4) PM_WAIT_VMCNT(12);
    else if (nk == 3) PM_WAIT_VMCNT(8);
    else if (nk == 2) PM_WAIT_VMCNT(4);
    else PM_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    fread(0, 0, 0);

    for (int t = 0; t < nk; ++t) {
        const int stage = t & (DSTAGES - 1);
        fread(stage, 1, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fa[0][i], fb[0][j], acc[i][j]);
        if (t + 1 < nk) {
            const int ahead = nk - t - 2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ahead >= 2) PM_WAIT_VMCNT(8);
            else if (ahead == 1) PM_WAIT_VMCNT(4);
            else PM_WAIT_VMCNT(0);
            __builtin_amdgcn_s_barrier();
            fread((t + 1) & (DSTAGES - 1), 0, 0);
            if (t + DSTAGES < nk) dma(t + DSTAGES, stage);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fa[1][i], fb[1][j], acc[i][j]);
    }

    double *cbase = C + (int64_t)(m0 + wm * 64 + fk) * ldc + n0 + wn * 64 + fcol;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) pm_atomic_add(cbase + (int64_t)(i * 16 + 4 * r) * ldc + j * 16, PM_Q(acc[i][j][r], 0));
}

// out[n] = sum_d Y[n,d]^2; one wavefront per row, lanes stride the row.
__global__ __launch_bounds__(256) void row_sqnorm_f64_kernel(const double *__restrict__ Y, int64_t ldy, int64_t N,
                                                              int D, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t n = wave0; n < N; n += nwaves) {
        const double *y = Y + n * ldy;
        double s = 0.0;
        for (int d = lane; d < D; d += 64) s = fma(y[d], y[d], s);
        s = pm_wave_sum(s);
        if (lane == 0) out[n] = s;
    }
}

// out[n] = sum_d w[d] Y[n,d]^2: y^T Sigma^-1 y for a diagonal noise covariance (gsc_et.py:418-419, 476).
__global__ __launch_bounds__(256) void row_wsqnorm_f64_kernel(const double *__restrict__ Y, int64_t ldy, int64_t N,
                                                               int D, const double *__restrict__ w,
                                                               double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t n = wave0; n < N; n += nwaves) {
        const double *y = Y + n * ldy;
        double s = 0.0;
        for (int d = lane; d < D; d += 64) s = fma(y[d] * w[d], y[d], s);
        s = pm_wave_sum(s);
        if (lane == 0) out[n] = s;
    }
}

// sums[d] += sum_n (Y[n,d] - center[d])^2 (center given) or sum_n Y[n,d] (center null): the two passes of
// CAModel.standard_init (camodels/__init__.py:209-217).  A workgroup walks a slab of rows with 256
// consecutive columns per pass (coalesced 2 KB row segments) and adds its partial column sums atomically.
__global__ __launch_bounds__(256) void col_moments_f64_kernel(const double *__restrict__ Y, int64_t ldy, int64_t N,
                                                               int D, const double *__restrict__ center,
                                                               double *__restrict__ sums, int64_t rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
    for (int d = blockIdx.x * 256 + threadIdx.x; d < D; d += gridDim.x * 256) {
        const double c = center ? center[d] : 0.0;
        double a0 = 0.0, a1 = 0.0;
        int64_t n = r0;
        for (; n + 1 < r1; n += 2) {
            const double u = Y[n * ldy + d] - c, v = Y[(n + 1) * ldy + d] - c;
            a0 += center ? u * u : u;
            a1 += center ? v * v : v;
        }
        if (n < r1) {
            const double u = Y[n * ldy + d] - c;
            a0 += center ? u * u : u;
        }
        pm_atomic_add(sums + d, PM_Q(a0 + a1, 1));
    }
}

// sums[d] += sum over the datapoints with lse[n] >= cut of Y[n,d]: my_data_sum of BSC_ET.M_step when 'mu' is learned
// (bsc_et.py:422-430), over the rows the truncation keeps (:247-258).  Same walk as col_moments_f64_kernel.
__global__ __launch_bounds__(256) void col_sum_kept_f64_kernel(const double *__restrict__ Y, int64_t ldy, int64_t N, int D,
                                                                const double *__restrict__ lse, double cut,
                                                                double *__restrict__ sums, int64_t rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
    for (int d = blockIdx.x * 256 + threadIdx.x; d < D; d += gridDim.x * 256) {
        double a0 = 0.0, a1 = 0.0;
        int64_t n = r0;
        for (; n + 1 < r1; n += 2) {
            const double u = Y[n * ldy + d], v = Y[(n + 1) * ldy + d];
            a0 += (lse[n] >= cut) ? u : 0.0;
            a1 += (lse[n + 1] >= cut) ? v : 0.0;
        }
        if (n < r1 && lse[n] >= cut) a0 += Y[n * ldy + d];
        pm_atomic_add(sums + d, PM_Q(a0 + a1, 1));
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int resident_slots() {  // workgroups of these kernels resident at once: 2 per CU
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        slots = 2 * cus;
    }
    return slots;
}

// (The two-launch form -- main rounds, then the split-K remainder -- and a forced split factor were environment
// switches in round 2; measured, settled, removed: the library reads no environment.)
constexpr bool fuse_remainder() { return true; }

template <int MT>
void launch_nt_mt(bool al, const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int M,
                  int N, int K, hipStream_t s) {
    const int tiles_m = (M + 32 * MT - 1) / (32 * MT), tiles_n = (N + BN - 1) / BN;
    dim3 grid((unsigned)((int64_t)tiles_m * tiles_n)), block(256);
    if (al)
        hipLaunchKernelGGL((gemm_nt_f64_kernel<MT, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
    else
        hipLaunchKernelGGL((gemm_nt_f64_kernel<MT, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K,
                           tiles_n);
}

// LDS-DMA kernel over M rows; nsplit > 1 zeroes C and splits K over grid.y
void launch_nt_dma(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int M, int N,
                   int K, int nsplit, hipStream_t s) {
    const int tiles_m = (M + 127) / 128, tiles_n = (N + BN - 1) / BN;
    const int nk = K / DK;
    int kps = (nk + nsplit - 1) / nsplit;
    nsplit = (nk + kps - 1) / kps;
    if (nsplit > 1) (void)hipMemset2DAsync(C, (size_t)ldc * sizeof(double), 0, (size_t)N * sizeof(double), (size_t)M, s);
    dim3 grid((unsigned)((int64_t)tiles_m * tiles_n), (unsigned)nsplit), block(256);
    if (nsplit > 1)
        hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<true>, grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, tiles_n, kps,
                           (int)grid.x, 0);
    else
        hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<false>, grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, tiles_n,
                           kps, (int)grid.x, 0);
}

// 64-row tiles, un-split (the ragged last round of a tall product with a short K: see the kernel)
void launch_nt_dma64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int M, int N,
                     int K, hipStream_t s) {
    const int tiles_m = (M + 63) / 64, tiles_n = (N + BN - 1) / BN;
    dim3 grid((unsigned)((int64_t)tiles_m * tiles_n), 1), block(256);
    hipLaunchKernelGGL((gemm_nt_f64_dma_kernel<false, 64>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, tiles_n,
                       K / DK, (int)grid.x, 0);
}

// whole rounds of un-split tiles + K-slices of the ragged last round's tiles in ONE launch (see the kernel)
void launch_nt_dma_fused(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int M, int N,
                         int K, int main_panels, int nsplit, hipStream_t s) {
    const int tiles_m = (M + 127) / 128, tiles_n = (N + BN - 1) / BN;
    const int nk = K / DK;
    int rest_kps = (nk + nsplit - 1) / nsplit;
    nsplit = (nk + rest_kps - 1) / rest_kps;
    const int main_tiles = main_panels * tiles_n, rest_tiles = (tiles_m - main_panels) * tiles_n;
    (void)hipMemset2DAsync(C + (int64_t)main_panels * 128 * ldc, (size_t)ldc * sizeof(double), 0, (size_t)N * sizeof(double),
                           (size_t)(M - main_panels * 128), s);
    dim3 grid((unsigned)(main_tiles + rest_tiles * nsplit)), block(256);
    hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<false>, grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, tiles_n, nk,
                       main_tiles, rest_kps);
}

}  // namespace

extern "C" int pm_gemm_nt_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                              int64_t M, int64_t N, int64_t K, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldb < K || ldc < N) return PM_EINVAL;
    if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return PM_ERANGE;
    const bool al = aligned16(A) && aligned16(B) && (lda % 2 == 0) && (ldb % 2 == 0) && (K % 2 == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);

    // Workgroups resident at once ("slots", 2 per CU) process the grid in rounds.  Whole rounds of
    // 128x128 tiles run un-split; the ragged remainder (and any problem smaller than one round, e.g.
    // the H x H Gram matrix) is split over K so that it still covers the chip for a fraction of a
    // round instead of leaving most CUs idle for a whole one.
    const int slots = resident_slots();
    const int64_t tiles_n = (N + BN - 1) / BN;
    if (al && K % DK == 0) {
        const int64_t panels = (M + 127) / 128;
        const int64_t per_round = slots / tiles_n > 0 ? slots / tiles_n : 1;
        const int64_t main_panels = (M / 128) / per_round * per_round;
        const int64_t rest_rows = M - main_panels * 128;
        int64_t nsplit = 1;
        if (rest_rows > 0) {
            const int64_t rest_tiles = (panels - main_panels) * tiles_n;
            nsplit = slots / rest_tiles;                   // fill the slots once
            const int64_t max_split = (K / DK) / 8;        // at least 8 K-steps per workgroup
            if (nsplit > max_split) nsplit = max_split;
            if (nsplit < 1 || rest_tiles * 10 >= (int64_t)slots * 7) nsplit = 1;
            // many K-slices of many tiles: the slices' f64 atomics (16 K per workgroup) outweigh the shorter K-loops
            // (3392 x 256 x 1024: 9 slices 67 us, 4 slices 56 us, 2 slices 77 us)
            if (nsplit > 4 && rest_tiles >= 32) nsplit = nsplit / 2 > 4 ? nsplit / 2 : 4;
        }
#ifdef PM_DETERMINISTIC
        // The K-slices of a split remainder meet in f64 atomics, whose order is not fixed -- and this product (scores, selection
        // distances, ragged remainders) has no bound of its own from which a quantum could be derived: the deterministic build
        // never splits K here (every element of C is then written once, by one workgroup, in a fixed summation order).  The
        // remainder of a shard runs a fraction of a round at full K instead: a few per cent of one launch.  [round-5 advisor
        // finding: the slices used to be quantised with the `gemm` unit's category 0, i.e. with whatever M-step bound the
        // previous step -- or another model of the process -- had installed, or with none at all in a fresh process.]
        nsplit = 1;
        // ... and as ONE launch: the remainder's tiles take the slots the last whole round frees (as a launch of its own it ran
        // alone behind the rounds: 37 us for 27 tiles at GSC's config 4, on the EM loop's critical path)
        if (main_panels > 0 && rest_rows > 0) {
            launch_nt_dma(A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, 1, s);
            return (int)hipGetLastError();
        }
#endif
        if (main_panels > 0 && rest_rows > 0 && nsplit > 1 && fuse_remainder()) {
            launch_nt_dma_fused(A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, (int)main_panels, (int)nsplit, s);
            return (int)hipGetLastError();
        }
        if (main_panels > 0) launch_nt_dma(A, lda, B, ldb, C, ldc, (int)(main_panels * 128), (int)N, (int)K, 1, s);
        if (rest_rows > 0) {
            const int64_t rest_tiles = (panels - main_panels) * tiles_n;
            if (nsplit == 1 && main_panels > 0 && rest_tiles * 4 >= slots && rest_tiles * 10 < (int64_t)slots * 7)
                launch_nt_dma64(A + main_panels * 128 * lda, lda, B, ldb, C + main_panels * 128 * ldc, ldc, (int)rest_rows,
                                (int)N, (int)K, s);
            else
                launch_nt_dma(A + main_panels * 128 * lda, lda, B, ldb, C + main_panels * 128 * ldc, ldc, (int)rest_rows,
                              (int)N, (int)K, (int)nsplit, s);
        }
        return (int)hipGetLastError();
    }
    // register-staged kernels: any alignment, any K; small tiles for small problems
    const int64_t t128 = (M + 127) / 128 * tiles_n;
    if (t128 * 2 <= slots && M > 32)
        (M + 31) / 32 * tiles_n <= slots ? launch_nt_mt<1>(al, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, s)
                                         : launch_nt_mt<2>(al, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, s);
    else
        launch_nt_mt<4>(al, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, s);
    return (int)hipGetLastError();
}

extern "C" int pm_gemm_tn_acc_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                                  int64_t M, int64_t N, int64_t K, void *stream) {
    return pm_gemm_tn_acc_gated_f64(A, lda, B, ldb, C, ldc, M, N, K, nullptr, stream);
}

extern "C" int pm_gemm_tn_acc_gated_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C,
                                        int64_t ldc, int64_t M, int64_t N, int64_t K, const double *gate,
                                        void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K < 0 || lda < M || ldb < N || ldc < N) return PM_EINVAL;
    if (M > INT32_MAX || N > INT32_MAX) return PM_ERANGE;
    if (K == 0) return PM_OK;
    const int tiles_m = (int)((M + TN_BM - 1) / TN_BM), tiles_n = (int)((N + BN - 1) / BN);
    const int64_t tiles = (int64_t)tiles_m * tiles_n;
    // K-splits: whole rounds of resident workgroups (2 per CU), at least 8 K-steps each
    const int slots = resident_slots();
    int64_t nsplit = (2 * slots + tiles - 1) / tiles;
    if (nsplit * tiles > 2 * slots && nsplit > 1) nsplit = 2 * slots / tiles > 0 ? 2 * slots / tiles : 1;
    const int64_t max_split = (K + 8 * BK - 1) / (8 * BK);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 65535) nsplit = 65535;
    int64_t kps = (K + nsplit - 1) / nsplit;
    kps = (kps + BK - 1) / BK * BK;
    nsplit = (K + kps - 1) / kps;
    const bool al = aligned16(A) && aligned16(B) && (lda % 2 == 0) && (ldb % 2 == 0) && (M % 2 == 0) && (N % 2 == 0);
    dim3 grid((unsigned)tiles, (unsigned)nsplit), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (al && M % TN_BM == 0 && N % BN == 0 && K >= 8 * DK) {
        // LDS-DMA kernel over the first K - K % 8 rows (every split a whole number of 8-row steps, kps is a
        // multiple of 16); the last K % 8 rows go through the register-staged kernel
        const int64_t K8 = K - K % DK;
        dim3 g8((unsigned)tiles, (unsigned)((K8 + kps - 1) / kps));
        hipLaunchKernelGGL(gemm_tn_f64_dma_kernel<false>, g8, block, 0, s, A, lda, B, ldb, C, ldc, tiles_n, K8, kps, gate,
                           nullptr, nullptr, 0);
        if (K8 < K)
            hipLaunchKernelGGL(gemm_tn_f64_kernel<true>, dim3((unsigned)tiles, 1), block, 0, s, A + K8 * lda, lda,
                               B + K8 * ldb, ldb, C, ldc, (int)M, (int)N, K - K8, tiles_n, (int64_t)BK, gate);
    } else if (al)
        hipLaunchKernelGGL(gemm_tn_f64_kernel<true>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, K,
                           tiles_n, kps, gate);
    else
        hipLaunchKernelGGL(gemm_tn_f64_kernel<false>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, K,
                           tiles_n, kps, gate);
    return (int)hipGetLastError();
}

// C += A[rows]^T . B[rows] over a device-side row list (see gemm_tn_f64_dma_kernel<true>): `rows` holds *count row
// indices (count <= max_rows, both on the device); `zero_row` is a row of zeros in A and in B.
extern "C" int pm_gemm_tn_acc_rows_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                                       int64_t M, int64_t N, const int32_t *rows, const int32_t *count, int64_t max_rows,
                                       int64_t zero_row, void *stream) {
    if (!A || !B || !C || !rows || !count || M <= 0 || N <= 0 || max_rows < 0 || lda < M || ldb < N || ldc < N || zero_row < 0)
        return PM_EINVAL;
    if (M > INT32_MAX || N > INT32_MAX || max_rows > INT32_MAX - 64 || zero_row > INT32_MAX) return PM_ERANGE;
    const bool al = aligned16(A) && aligned16(B) && (lda % 2 == 0) && (ldb % 2 == 0);
    if (!al || M % TN_BM != 0 || N % BN != 0) return PM_ERANGE;
    if (max_rows == 0) return PM_OK;
    const int tiles_m = (int)(M / TN_BM), tiles_n = (int)(N / BN);
    const int64_t tiles = (int64_t)tiles_m * tiles_n;
    const int slots = resident_slots();
#ifndef PM_TN_ROWS_DIV
#define PM_TN_ROWS_DIV 1
#endif
    int64_t nsplit = slots / tiles / PM_TN_ROWS_DIV;    // ONE round of resident workgroups, whatever *count turns out to be
    const int64_t max_split = (max_rows + 8 * DK - 1) / (8 * DK);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 65535) nsplit = 65535;
    hipLaunchKernelGGL(gemm_tn_f64_dma_kernel<true>, dim3((unsigned)tiles, (unsigned)nsplit), dim3(256), 0,
                       static_cast<hipStream_t>(stream), A, lda, B, ldb, C, ldc, tiles_n, max_rows, (int64_t)0, nullptr, rows,
                       count, (int)zero_row);
    return (int)hipGetLastError();
}

extern "C" int pm_row_sqnorm_f64(const double *Y, int64_t ldy, int64_t N, int64_t D, double *out, void *stream) {
    if (N == 0) return PM_OK;
    if (!Y || !out || N < 0 || D <= 0 || ldy < D) return PM_EINVAL;
    if (D > INT32_MAX) return PM_ERANGE;
    int64_t blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(row_sqnorm_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), Y,
                       ldy, N, (int)D, out);
    return (int)hipGetLastError();
}

extern "C" int pm_col_moments_f64(const double *Y, int64_t ldy, int64_t N, int64_t D, const double *center,
                                  double *sums, void *stream) {
    if (N == 0) return PM_OK;
    if (!Y || !sums || N < 0 || D <= 0 || ldy < D) return PM_EINVAL;
    if (D > INT32_MAX) return PM_ERANGE;
    const unsigned gx = (unsigned)((D + 255) / 256 > 64 ? 64 : (D + 255) / 256);
    int64_t gy = (2 * (int64_t)resident_slots() + gx - 1) / gx;       // ~4 workgroups per CU in flight
    if (gy > N) gy = N;
    const int64_t rows_per_block = (N + gy - 1) / gy;
    gy = (N + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(col_moments_f64_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, static_cast<hipStream_t>(stream), Y,
                       ldy, N, (int)D, center, sums, rows_per_block);
    return (int)hipGetLastError();
}

extern "C" int pm_col_sum_kept_f64(const double *Y, int64_t ldy, int64_t N, int64_t D, const double *lse, double cut,
                                   double *sums, void *stream) {
    if (N == 0) return PM_OK;
    if (!Y || !lse || !sums || N < 0 || D <= 0 || ldy < D) return PM_EINVAL;
    if (D > INT32_MAX) return PM_ERANGE;
    const unsigned gx = (unsigned)((D + 255) / 256 > 64 ? 64 : (D + 255) / 256);
    int64_t gy = (2 * (int64_t)resident_slots() + gx - 1) / gx;
    if (gy > N) gy = N;
    const int64_t rows_per_block = (N + gy - 1) / gy;
    gy = (N + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(col_sum_kept_f64_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, static_cast<hipStream_t>(stream), Y, ldy,
                       N, (int)D, lse, cut, sums, rows_per_block);
    return (int)hipGetLastError();
}

extern "C" int pm_row_wsqnorm_f64(const double *Y, int64_t ldy, int64_t N, int64_t D, const double *w, double *out,
                                  void *stream) {
    if (N == 0) return PM_OK;
    if (!Y || !w || !out || N < 0 || D <= 0 || ldy < D) return PM_EINVAL;
    if (D > INT32_MAX) return PM_ERANGE;
    int64_t blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(row_wsqnorm_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       Y, ldy, N, (int)D, w, out);
    return (int)hipGetLastError();
}

PM_DET_SETTER(gemm)
