// C = A . B^T for SMALL outputs (M, N <= 1024: the H x H Gram matrix W W^T of every model's E-step, bsc_et.py:177-183 in
// its Gram form, GSC's W^T Sigma^-1 W, the H x H x D solve products) -- one launch, deterministic.
//
// pm_gemm_nt_f64 treats such a product as the ragged remainder of a big one: 128 x 128 tiles split over K with f64
// atomics.  That costs 23-27 us for the 134 Mflop of config 2's Gram matrix (2 x 2 tiles x 16 K-slices: the atomics and
// the zero fill in front of them are most of it) and, with more than two slices per tile, its bits depend on the order the
// atomics land in -- two ranks holding the same W got Gram matrices that differed in the last place (round 4: GSC's
// sigma_sq, computed from trace(U . W^T W) on the device, differed by 1 ulp between two ranks).
//
// Here: one workgroup of 8 wavefronts per 16 x 16 output tile (H = 256: 256 tiles = the chip, one per CU); wavefront w
// takes every eighth 16-column K-chunk, requests ALL its operands (32 bytes per lane and chunk: whole 128-byte lines per
// row) before its first v_mfma_f64_16x16x4_f64 -- one L2 round trip --, and the eight partial tiles are summed through LDS
// in a fixed order.  A == B (Gram): only tiles on or above the diagonal compute, the mirror image is stored with them, so
// the result is symmetric bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prosper_hip.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int SW = 8;          // wavefronts per workgroup
constexpr int CH = 16;         // K columns per chunk: lane (r, kq) holds columns 4 kq .. 4 kq + 3 of row r
// (UNR, template parameter: chunks a wavefront keeps in flight -- 8 for K >= 1024, fewer for shorter products)

// columns k .. k + 3 of `row` (zeros beyond kend); ALIGNED: rows start 16-byte aligned and K % 4 == 0
template <bool ALIGNED>
__device__ __forceinline__ void load4(const double *__restrict__ row, int64_t k, int64_t kend, double (&v)[4]) {
    if (ALIGNED) {
        if (k + 4 <= kend) {
            const d2 lo = *reinterpret_cast<const d2 *>(row + k), hi = *reinterpret_cast<const d2 *>(row + k + 2);
            v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
        } else {
            v[0] = v[1] = v[2] = v[3] = 0.0;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k + e < kend) ? row[k + e] : 0.0;
    }
}

template <bool ALIGNED, int UNR>
__global__ __launch_bounds__(64 * SW) void gemm_nt_small_kernel(const double *__restrict__ A, int64_t lda,
                                                                 const double *__restrict__ B, int64_t ldb,
                                                                 double *__restrict__ C, int64_t ldc, int M, int N,
                                                                 int64_t K, int gram) {
    __shared__ double s_t[SW][4][64];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (gram && bj < bi) return;                         // (the tile above the diagonal stores this one as its mirror)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int i0 = bi * 16, j0 = bj * 16;
    const int ia = (i0 + r < M) ? i0 + r : M - 1, jb = (j0 + r < N) ? j0 + r : N - 1;
    const double *pa = A + (int64_t)ia * lda + 4 * kq, *pb = B + (int64_t)jb * ldb + 4 * kq;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int64_t nch = (K + CH - 1) / CH;
    // chunk c belongs to wavefront c % SW; UNR of them per trip, every load of a trip ahead of its first MFMA
    for (int64_t c0 = wave; c0 < nch; c0 += (int64_t)SW * UNR) {
        double a[UNR][4], b[UNR][4];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t c = c0 + (int64_t)u * SW;
            const int64_t k = (c < nch ? c : nch - 1) * CH;              // (past the end: a valid address, zeroed below)
            const int64_t kend = (c < nch) ? K - 4 * kq : 0;              // columns this lane may read, relative to pa
            load4<ALIGNED>(pa, k, kend, a[u]);
            load4<ALIGNED>(pb, k, kend, b[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][e], b[u][e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s_t[wave][q][lane] = acc[q];
    __syncthreads();
    if (wave != 0) return;
    const int j = j0 + r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + kq + 4 * q;                   // C layout of the MFMA: row (lane >> 4) + 4 q, column lane & 15
        double v = s_t[0][q][lane];
#pragma unroll
        for (int w = 1; w < SW; ++w) v += s_t[w][q][lane];                 // fixed order
        if (i < M && j < N) {
            C[(int64_t)i * ldc + j] = v;
            if (gram && bi != bj) C[(int64_t)j * ldc + i] = v;
        }
    }
}


// C = A . B (A: M x K, B: K x N, both row-major) for small M: the W solve X = Wq^-1 . Wp of the M-step (bsc_et.py:380 via the
// device inverse) and its relatives.  Same scheme: the A fragment is 32 contiguous bytes per lane and chunk, the B fragment
// four 8-byte loads that sixteen lanes make contiguous (128 bytes per K row of the tile).
template <int UNR>
__global__ __launch_bounds__(64 * SW) void gemm_nn_small_kernel(const double *__restrict__ A, int64_t lda,
                                                                 const double *__restrict__ B, int64_t ldb,
                                                                 double *__restrict__ C, int64_t ldc, int M, int N,
                                                                 int64_t K, int aligned) {
    __shared__ double s_t[SW][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int ia = (i0 + r < M) ? i0 + r : M - 1, jb = (j0 + r < N) ? j0 + r : N - 1;
    const double *pa = A + (int64_t)ia * lda + 4 * kq, *pb = B + (int64_t)(4 * kq) * ldb + jb;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int64_t nch = (K + CH - 1) / CH;
    for (int64_t c0 = wave; c0 < nch; c0 += (int64_t)SW * UNR) {
        double a[UNR][4], b[UNR][4];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t c = c0 + (int64_t)u * SW;
            const int64_t k = (c < nch ? c : nch - 1) * CH;
            const int64_t kend = (c < nch) ? K - 4 * kq : 0;
            if (aligned) load4<true>(pa, k, kend, a[u]);
            else load4<false>(pa, k, kend, a[u]);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[u][e] = (k + e < kend) ? pb[(k + e) * ldb] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][e], b[u][e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s_t[wave][q][lane] = acc[q];
    __syncthreads();
    if (wave != 0) return;
    const int j = j0 + r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + kq + 4 * q;
        double v = s_t[0][q][lane];
#pragma unroll
        for (int w = 1; w < SW; ++w) v += s_t[w][q][lane];                 // fixed order
        if (i < M && j < N) C[(int64_t)i * ldc + j] = v;
    }
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pm_gemm_nt_small_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                                    int64_t M, int64_t N, int64_t K, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldb < K || ldc < N) return PM_EINVAL;
    if (M > 1024 || N > 1024) return PM_ERANGE;
    const int gram = (A == B && lda == ldb && M == N) ? 1 : 0;
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)((M + 15) / 16)), block(64 * SW);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool al = al16(A) && al16(B) && lda % 2 == 0 && ldb % 2 == 0 && K % 4 == 0;
    const int64_t per_wave = ((K + CH - 1) / CH + SW - 1) / SW;          // chunks per wavefront
#define PM_NT_SMALL(AL, U) \
    hipLaunchKernelGGL((gemm_nt_small_kernel<AL, U>), grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, K, gram)
    if (al) {
        if (per_wave <= 1) PM_NT_SMALL(true, 1);
        else if (per_wave <= 2) PM_NT_SMALL(true, 2);
        else if (per_wave <= 4) PM_NT_SMALL(true, 4);
        else PM_NT_SMALL(true, 8);
    } else {
        if (per_wave <= 2) PM_NT_SMALL(false, 2);
        else PM_NT_SMALL(false, 8);
    }
#undef PM_NT_SMALL
    return (int)hipGetLastError();
}

extern "C" int pm_gemm_nn_small_f64(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                                    int64_t M, int64_t N, int64_t K, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldb < N || ldc < N) return PM_EINVAL;
    if (M > 1024) return PM_ERANGE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int al = (al16(A) && lda % 2 == 0 && K % 4 == 0) ? 1 : 0;
    const int64_t per_wave = ((K + CH - 1) / CH + SW - 1) / SW;
    // (columns in slabs of 65536: the solve products of an M-step with more observed dimensions than one grid row holds)
    for (int64_t n0 = 0; n0 < N; n0 += 65536) {
        const int64_t Nc = N - n0 < 65536 ? N - n0 : 65536;
        const double *Bc = B + n0;
        double *Cc = C + n0;
        const dim3 grid((unsigned)((Nc + 15) / 16), (unsigned)((M + 15) / 16)), block(64 * SW);
#define PM_NN_SMALL(U) \
    hipLaunchKernelGGL((gemm_nn_small_kernel<U>), grid, block, 0, s, A, lda, Bc, ldb, Cc, ldc, (int)M, (int)Nc, K, al)
        if (per_wave <= 1) PM_NN_SMALL(1);
        else if (per_wave <= 2) PM_NN_SMALL(2);
        else if (per_wave <= 4) PM_NN_SMALL(4);
        else PM_NN_SMALL(8);
#undef PM_NN_SMALL
    }
    return (int)hipGetLastError();
}
