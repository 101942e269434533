#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <stdint.h>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
constexpr int DK = 8, DSTAGES = 4, DSTAGE = (128 + 128) * DK;  // doubles per stage (16 KB)

#define PM_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ __launch_bounds__(256, 2) void gemm_nt_f64_dma_kernel(const double *__restrict__ A, int64_t lda,
                                                                  const double *__restrict__ B, int64_t ldb,
                                                                  double *__restrict__ C, int64_t ldc, int M, int N,
                                                                  int K, int tiles_n, unsigned long long* stamps) {
    unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    __shared__ __attribute__((aligned(1024))) double sm[DSTAGES * DSTAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bn = blockIdx.x % tiles_n, bm = blockIdx.x / tiles_n;
    const int m0 = bm * 128, n0 = bn * 128;

    // DMA sources: this wavefront moves chunks {wave, wave+4} of A and of B
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
        double *dst = sm + stage * DSTAGE + wave * 128;  // chunk = 128 doubles
        const int k0 = kt * DK;
        __builtin_amdgcn_global_load_lds(src[0] + k0, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src[1] + k0, dst + 4 * 128, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src[2] + k0, dst + 8 * 128, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src[3] + k0, dst + 12 * 128, 16, 0, 0);
    };

    // fragment reads: element q = kk*4 + fk of row R sits at R*8 + (((q>>1) ^ ((R>>2)&3)) << 1) + (q&1)
    const int frow = lane & 15, fk = lane >> 4;
    const int sw = (frow >> 2) & 3;
    int a_off[2], b_off[2];  // per kk
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int q = kk * 4 + fk;
        const int col = (((q >> 1) ^ sw) << 1) + (q & 1);
        a_off[kk] = (wm * 64 + frow) * DK + col;
        b_off[kk] = 128 * DK + (wn * 64 + frow) * DK + col;
    }
    double fa[2][4], fb[2][4];
    auto fread = [&](int stage, int kk, int slot) {
        const double *sa = sm + stage * DSTAGE + a_off[kk];
        const double *sb = sm + stage * DSTAGE + b_off[kk];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[slot][i] = sa[i * 16 * DK];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[slot][j] = sb[j * 16 * DK];
    };

    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = K / DK;  // host guarantees K % DK == 0 and nk >= DSTAGES
#pragma unroll
    for (int t = 0; t < DSTAGES; ++t) dma(t, t);
    PM_WAIT_VMCNT(12);  // this wavefront's part of K-step 0 has landed
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
            // my reads of this stage are done (lgkmcnt) and my share of K-step t+1 has landed (vmcnt);
            // after the barrier that holds for every wavefront: stage t may be refilled, t+1 may be read
            const int ahead = nk - t - 2;  // K-steps issued beyond t+1
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

    if (threadIdx.x == 0) { stamps[2*blockIdx.x] = sr0; stamps[2*blockIdx.x+1] = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + i * 16 + fk + 4 * r;
            if (row >= M) continue;
            double *crow = C + (int64_t)row * ldc + n0 + wn * 64 + frow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n0 + wn * 64 + j * 16 + frow < N) crow[j * 16] = acc[i][j][r];
            }
        }
    }
}



int main() {
    int N = 200000, D = 1024, H = 256, grid = 3072;
    double *Y, *W, *A; hipMalloc(&Y, (size_t)N * D * 8); hipMalloc(&W, H * D * 8); hipMalloc(&A, (size_t)(N + 128) * H * 8);
    unsigned long long* st; hipMalloc(&st, 16 * grid);
    std::vector<double> h((size_t)N * D); srand(1); for (auto& v : h) v = (rand() / (double)RAND_MAX) * 2 - 1;
    hipMemcpy(Y, h.data(), (size_t)N * D * 8, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), H * D * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 60; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(gemm_nt_f64_dma_kernel, dim3(grid), dim3(256), 0, 0, Y, (int64_t)D, W, (int64_t)D, A, (int64_t)H, N, H, D, 2, st);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0 || rep == 1 || rep == 5 || rep == 20 || rep == 59) {
            std::vector<unsigned long long> hs(2 * grid); hipMemcpy(hs.data(), st, 16 * grid, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, t1 = 0; for (int b = 0; b < grid; ++b) { t0 = std::min(t0, hs[2*b]); t1 = std::max(t1, hs[2*b+1]); }
            // concurrency histogram over 20 time bins; mean block duration
            const int NB = 24; std::vector<double> conc(NB, 0.0); double dur = 0;
            double span = (double)(t1 - t0);
            for (int b = 0; b < grid; ++b) { dur += (double)(hs[2*b+1] - hs[2*b]);
                for (int k = 0; k < NB; ++k) { double lo = t0 + span * k / NB, hi = t0 + span * (k + 1) / NB;
                    double ov = std::min((double)hs[2*b+1], hi) - std::max((double)hs[2*b], lo); if (ov > 0) conc[k] += ov / (hi - lo); } }
            printf("rep %2d: %.3f ms (span %.3f ms) mean block %.1f us; concurrency per bin:", rep, ms, span / 100e3, dur / grid / 100.0);
            for (int k = 0; k < NB; ++k) printf(" %.0f", conc[k]);
            printf("\n");
        }
    }
    return 0;
}
