// Same-wave interleave: K VALU ops of a class between consecutive f64 MFMAs of ONE wavefront per SIMD: cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CLS, int K>
__global__ __launch_bounds__(256, 2) void k(int iters, long long *cyc, double *sink) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    double x[8]; int u[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
    double c = 1.0000001; int ci = 7;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int r = (i + q) & 7;
                if (CLS == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[r]) : "v"(c));
                if (CLS == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[r]) : "v"(ci));
                if (CLS == 2) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[r]) : "v"(ci));
                if (CLS == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[r]) : "v"(ci) : "vcc");
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; int su = 0;
    for (int i = 0; i < 8; ++i) { s += acc[i][0] + acc[i][3] + x[i]; su += u[i]; }
    if (s == 12345.678 || su == 123456789) sink[0] = s + su;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int CLS, int K>
double run() {
    long long *cyc; double *sink; long long h[4];
    hipMalloc(&cyc, 4 * sizeof(long long)); hipMalloc(&sink, 16);
    const int IT = 4000;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k<CLS, K>), dim3(256), dim3(256), 0, 0, IT, cyc, sink); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    hipFree(cyc); hipFree(sink);
    return (double)h[0] / (8.0 * IT);
}
int main() {
    printf("cycles per MFMA with K VALU ops between MFMAs (one wave per SIMD)\n");
    printf("class        K=0    K=1    K=2    K=4    K=8    K=12\n");
    printf("v_fma_f64   %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<0,0>(), run<0,1>(), run<0,2>(), run<0,4>(), run<0,8>(), run<0,12>());
    printf("v_add_u32   %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<1,0>(), run<1,1>(), run<1,2>(), run<1,4>(), run<1,8>(), run<1,12>());
    printf("v_mov_dpp   %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<2,0>(), run<2,1>(), run<2,2>(), run<2,4>(), run<2,8>(), run<2,12>());
    printf("v_cndmask   %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<3,0>(), run<3,1>(), run<3,2>(), run<3,4>(), run<3,8>(), run<3,12>());
    return 0;
}
