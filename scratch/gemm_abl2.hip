#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int BN = 128, BK = 16;
constexpr int NT_LD = BK + 2;
__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <int MT, bool ALIGNED, bool EDGE, int ABL>
__device__ __forceinline__ void nt_tile(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                        int64_t ldb, double *__restrict__ C, int64_t ldc, int M, int N, int K,
                                        int m0, int n0, double *sm) {
    constexpr int BM = 32 * MT;
    constexpr int STAGE = (BM + BN) * NT_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
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
                if (!(ABL & 1)) ra[c] = *reinterpret_cast<const d2 *>(p);
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
                if (!(ABL & 2)) rb[c] = *reinterpret_cast<const d2 *>(p);
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
                if (!(ABL & 4)) swrite(buf ^ 1);
                if (!(ABL & 8)) __syncthreads();  // everyone: fragments of K-step t are in registers, K-step t+1 is in LDS
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


template <int ABL>
__global__ __launch_bounds__(256, 2) void kern(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int M, int N, int K, int tiles_n) {
    __shared__ __attribute__((aligned(16))) double sm[2 * (128 + BN) * NT_LD];
    for (int i = threadIdx.x; i < 2 * (128 + BN) * NT_LD; i += 256) sm[i] = 1.0;
    __syncthreads();
    const int bn = blockIdx.x % tiles_n, bm = blockIdx.x / tiles_n;
    nt_tile<4, true, false, ABL>(A, lda, B, ldb, C, ldc, M, N, K, bm * 128, bn * BN, sm);
}
template <int ABL> void run(const char* what, const double* Y, const double* W, double* A, int N, int D, int H, int grid) {
    int tiles_n = (H + BN - 1) / BN;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kern<ABL>), dim3(grid), dim3(256), 0, 0, Y, (int64_t)D, W, (int64_t)D, A, (int64_t)H, N, H, D, tiles_n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((kern<ABL>), dim3(grid), dim3(256), 0, 0, Y, (int64_t)D, W, (int64_t)D, A, (int64_t)H, N, H, D, tiles_n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s ABL=%2d: %.3f ms  %.1f TF/s\n", what, ABL, ms, 2.0 * grid * 128.0 * BN * D / ms / 1e9);
}
int main() {
    int N = 200000, D = 1024, H = 256;
    double *Y, *W, *A; hipMalloc(&Y, (size_t)N * D * 8); hipMalloc(&W, H * D * 8); hipMalloc(&A, (size_t)(N + 128) * H * 8);
    std::vector<double> h((size_t)N * D); srand(1); for (auto& v : h) v = (rand() / (double)RAND_MAX) * 2 - 1;
    hipMemcpy(Y, h.data(), (size_t)N * D * 8, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), H * D * 8, hipMemcpyHostToDevice);
    run<0>("full", Y, W, A, N, D, H, 3072);
    run<1>("no A(Y) loads", Y, W, A, N, D, H, 3072);
    run<2>("no B(W) loads", Y, W, A, N, D, H, 3072);
    run<3>("no loads", Y, W, A, N, D, H, 3072);
    run<7>("no loads, no LDS writes", Y, W, A, N, D, H, 3072);
    run<15>("no loads, no writes, no barrier", Y, W, A, N, D, H, 3072);
    run<8>("no barrier only", Y, W, A, N, D, H, 3072);
    run<4>("loads issued, never written", Y, W, A, N, D, H, 3072);
    return 0;
}
