// f64 MFMA GEMM kernels for gfx950 (v_mfma_f64_16x16x4_f64).
//
//   gemm_nt : C[M,N]  = A[M,K] . B[N,K]^T   scores a = Y.W^T and Gram G = W.W^T
//   gemm_tn : C[M,N] += A[K,M]^T . B[K,N]   Wp = E[s]^T . Y, split over the datapoint index
//
// Tiling (both): 128x128 output tile per 256-thread workgroup = 4 wavefronts in a 2x2 grid,
// each wavefront owns 64x64 = 4x4 MFMA tiles of 16x16 (accumulators: 16 x 4 f64 = 128 VGPRs),
// K advanced in steps of 16 through two LDS buffers: the global loads of tile t+1 are issued
// before the MFMAs of tile t and written to the other buffer after them, one barrier per step.
// An f64 MFMA occupies the matrix pipe for 64 cycles, so one wavefront issues 64 MFMAs
// (4096 cycles) per K-step against 8 16-byte global loads, 8 LDS writes and 32 ds_read_b64:
// the loop is matrix-pipe bound by construction.
//
// Fragment maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l & 15][k = l >> 4]      B: lane l holds B[k = l >> 4][j = l & 15]
//   C/D: 4 f64 per lane, col = l & 15, row = (l >> 4) + 4 * reg
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int NT_LD = BK + 2;    // LDS row stride (doubles) for K-contiguous tiles: 144 B rows ->
                                 // the 16 rows x 2 k of one ds_read_b64 half-wave hit 32 distinct bank pairs
constexpr int TN_LD = BM + 16;   // LDS row stride for K-strided tiles: consecutive k land 32 banks apart

__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// C = A . B^T, A (M,K) and B (N,K) row-major.
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(const double *__restrict__ A, int64_t lda,
                                                              const double *__restrict__ B, int64_t ldb,
                                                              double *__restrict__ C, int64_t ldc, int M, int N,
                                                              int K, int tiles_n) {
    __shared__ __attribute__((aligned(16))) double sm[2][(BM + BN) * NT_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bn = blockIdx.x % tiles_n, bm = blockIdx.x / tiles_n;
    const int m0 = bm * BM, n0 = bn * BN;

    // staging map: thread -> (row = tid/8 + 32 c, k pair = 2 (tid % 8)), c = 0..3
    const int srow = tid >> 3, skc = (tid & 7) * 2;
    d2 ra[4], rb[4];

    auto gload = [&](int k0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = srow + 32 * c;
            const int k = k0 + skc;
            d2 va = {0.0, 0.0}, vb = {0.0, 0.0};
            if (ALIGNED) {
                if (m0 + r < M && k < K) va = *reinterpret_cast<const d2 *>(A + (int64_t)(m0 + r) * lda + k);
                if (n0 + r < N && k < K) vb = *reinterpret_cast<const d2 *>(B + (int64_t)(n0 + r) * ldb + k);
            } else {
                if (m0 + r < M) {
                    const double *p = A + (int64_t)(m0 + r) * lda + k;
                    if (k < K) va.x = p[0];
                    if (k + 1 < K) va.y = p[1];
                }
                if (n0 + r < N) {
                    const double *p = B + (int64_t)(n0 + r) * ldb + k;
                    if (k < K) vb.x = p[0];
                    if (k + 1 < K) vb.y = p[1];
                }
            }
            ra[c] = va;
            rb[c] = vb;
        }
    };
    auto swrite = [&](int buf) {
        double *sa = sm[buf], *sb = sm[buf] + BM * NT_LD;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = srow + 32 * c;
            *reinterpret_cast<d2 *>(sa + r * NT_LD + skc) = ra[c];
            *reinterpret_cast<d2 *>(sb + r * NT_LD + skc) = rb[c];
        }
    };

    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = (K + BK - 1) / BK;
    gload(0);
    swrite(0);
    __syncthreads();

    const int frow = lane & 15, fk = lane >> 4;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) gload((t + 1) * BK);
        const double *sa = sm[t & 1] + (wm * 64 + frow) * NT_LD + fk;
        const double *sb = sm[t & 1] + BM * NT_LD + (wn * 64 + frow) * NT_LD + fk;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sa[i * 16 * NT_LD + kk * 4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sb[j * 16 * NT_LD + kk * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        if (t + 1 < nk) swrite((t + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane holds C[row = fk + 4 r][col = frow] of each 16x16 tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + i * 16 + fk + 4 * r;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + frow;
                if (col < N) C[(int64_t)row * ldc + col] = acc[i][j][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C += A^T . B, A (K,M) and B (K,N) row-major; the K range is split over blockIdx.y and the
// partial tiles are added with f64 atomics (global_atomic_add_f64).
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_tn_f64_kernel(const double *__restrict__ A, int64_t lda,
                                                              const double *__restrict__ B, int64_t ldb,
                                                              double *__restrict__ C, int64_t ldc, int M, int N,
                                                              int64_t K, int tiles_n, int64_t k_per_split) {
    __shared__ __attribute__((aligned(16))) double sm[2][2 * BK * TN_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bn = blockIdx.x % tiles_n, bm = blockIdx.x / tiles_n;
    const int m0 = bm * BM, n0 = bn * BN;
    const int64_t kbeg = (int64_t)blockIdx.y * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;

    // staging map: thread -> (k = tid/64 + 4 c, column pair = 2 (tid % 64)), c = 0..3
    const int sk = tid >> 6, scol = (tid & 63) * 2;
    d2 ra[4], rb[4];

    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t k = k0 + sk + 4 * c;
            d2 va = {0.0, 0.0}, vb = {0.0, 0.0};
            if (k < kend) {
                const double *pa = A + k * lda + m0 + scol;
                const double *pb = B + k * ldb + n0 + scol;
                if (ALIGNED) {
                    if (m0 + scol < M) va = *reinterpret_cast<const d2 *>(pa);
                    if (n0 + scol < N) vb = *reinterpret_cast<const d2 *>(pb);
                } else {
                    if (m0 + scol < M) va.x = pa[0];
                    if (m0 + scol + 1 < M) va.y = pa[1];
                    if (n0 + scol < N) vb.x = pb[0];
                    if (n0 + scol + 1 < N) vb.y = pb[1];
                }
            }
            ra[c] = va;
            rb[c] = vb;
        }
    };
    auto swrite = [&](int buf) {
        double *sa = sm[buf], *sb = sm[buf] + BK * TN_LD;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = sk + 4 * c;
            *reinterpret_cast<d2 *>(sa + k * TN_LD + scol) = ra[c];
            *reinterpret_cast<d2 *>(sb + k * TN_LD + scol) = rb[c];
        }
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

    const int fcol = lane & 15, fk = lane >> 4;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) gload(kbeg + (int64_t)(t + 1) * BK);
        const double *sa = sm[t & 1] + fk * TN_LD + wm * 64 + fcol;
        const double *sb = sm[t & 1] + BK * TN_LD + fk * TN_LD + wn * 64 + fcol;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sa[kk * 4 * TN_LD + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sb[kk * 4 * TN_LD + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        if (t + 1 < nk) swrite((t + 1) & 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + i * 16 + fk + 4 * r;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + fcol;
                if (col < N) pm_atomic_add(C + (int64_t)row * ldc + col, acc[i][j][r]);
            }
        }
    }
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

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pm_gemm_nt_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                              int64_t M, int64_t N, int64_t K, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldb < K || ldc < N) return PM_EINVAL;
    if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return PM_ERANGE;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    const bool al = aligned16(A) && aligned16(B) && (lda % 2 == 0) && (ldb % 2 == 0) && (K % 2 == 0);
    dim3 grid((unsigned)(tiles_m * (int64_t)tiles_n)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (al)
        hipLaunchKernelGGL(gemm_nt_f64_kernel<true>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K,
                           tiles_n);
    else
        hipLaunchKernelGGL(gemm_nt_f64_kernel<false>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K,
                           tiles_n);
    return (int)hipGetLastError();
}

extern "C" int pm_gemm_tn_acc_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                                  int64_t M, int64_t N, int64_t K, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K < 0 || lda < M || ldb < N || ldc < N) return PM_EINVAL;
    if (M > INT32_MAX || N > INT32_MAX) return PM_ERANGE;
    if (K == 0) return PM_OK;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    const int64_t tiles = (int64_t)tiles_m * tiles_n;
    // enough K-splits to put ~4 workgroups on every CU, at least 8 K-steps each
    int64_t nsplit = (1024 + tiles - 1) / tiles;
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
    if (al)
        hipLaunchKernelGGL(gemm_tn_f64_kernel<true>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, K,
                           tiles_n, kps);
    else
        hipLaunchKernelGGL(gemm_tn_f64_kernel<false>, grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, K,
                           tiles_n, kps);
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
