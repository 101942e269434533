// Bandwidth of the GEMM's A-tile load pattern alone (no MFMA): row-major 128 rows x 128 B per K-step
// vs a panel-tiled layout where each K-step tile is one contiguous 16 KB chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const double* __restrict__ Y, int64_t ld, double* out, int nk, int delay) {
    const int tid = threadIdx.x; const int srow = tid >> 3, skc = (tid & 7) * 2;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    d2 acc = {0, 0};
    for (int t = 0; t < nk; ++t) {
        d2 r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double* p;
            if (MODE == 0) p = Y + (m0 + srow + 32 * c) * ld + t * 16 + skc;
            else p = Y + m0 * ld + (int64_t)t * 128 * 16 + (srow + 32 * c) * 16 + skc;   // tile-contiguous
            r[c] = *reinterpret_cast<const d2*>(p);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc += r[c];
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(10);
    }
    if (acc.x == 12345.678) out[0] = acc.y;
}
template <int MODE> void run(const char* n, const double* Y, double* out, int blocks, int delay) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, Y, (int64_t)1024, out, 64, delay);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, Y, (int64_t)1024, out, 64, delay);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s blocks=%d delay=%d: %.3f ms  %.2f TB/s\n", n, blocks, delay, ms, blocks * 128.0 * 8192 / ms / 1e9);
}
int main() {
    double *Y, *out; size_t n = (size_t)200704 * 1024; hipMalloc(&Y, n * 8); hipMalloc(&out, 64); hipMemset(Y, 0, n * 8);
    for (int delay : {0, 4, 16}) {
        run<0>("row-major 128B x 128 rows", Y, out, 1536, delay);
        run<1>("tile-contiguous 16KB", Y, out, 1536, delay);
    }
    return 0;
}
