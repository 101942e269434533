// Microbenchmark: f64 MFMA rate, f64 VALU FMA rate, and both interleaved, on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NF>   // NF VALU fmas per MFMA
__global__ __launch_bounds__(256) void mix(double* out, int iters, double a0, double b0) {
    d4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    double f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = i * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NF; ++i) f[(u * NF + i) & 15] = fma(f[(u * NF + i) & 15], a, b);
        }
    }
    double s = 0;
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void valu_only(double* out, int iters, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    double f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = i * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = fma(f[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e-3;
}

int main() {
    double* out; hipMalloc(&out, sizeof(double) * 256 * 256 * 16);
    const int iters = 20000;
    for (int bpc : {1, 2, 4}) {
        int blocks = 256 * bpc;
        double waves = blocks * 4.0;
        auto rep = [&](const char* name, double t, double mf, double vf) {
            printf("%-22s blocks/CU=%d  %.3f ms  MFMA %.1f TF/s  VALU %.1f TF/s  total %.1f TF/s\n", name, bpc, t * 1e3,
                   mf / t / 1e12, vf / t / 1e12, (mf + vf) / t / 1e12);
        };
        double mflop = waves * iters * 4.0 * 2048.0;
        double t;
        t = timeit([&] { hipLaunchKernelGGL(mix<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        rep("mfma only", t, mflop, 0);
        t = timeit([&] { hipLaunchKernelGGL(valu_only, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        rep("valu fma only", t, 0, waves * iters * 64.0 * 128.0);
        t = timeit([&] { hipLaunchKernelGGL(mix<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        rep("mfma + 2 fma", t, mflop, waves * iters * 8.0 * 128.0);
        t = timeit([&] { hipLaunchKernelGGL(mix<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        rep("mfma + 4 fma", t, mflop, waves * iters * 16.0 * 128.0);
        t = timeit([&] { hipLaunchKernelGGL(mix<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        rep("mfma + 8 fma", t, mflop, waves * iters * 32.0 * 128.0);
        t = timeit([&] { hipLaunchKernelGGL(mix<12>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
        rep("mfma + 12 fma", t, mflop, waves * iters * 48.0 * 128.0);
    }
    return 0;
}
