// Does a VALU instruction stream on one wavefront slow down a back-to-back f64 MFMA stream on the OTHER wavefront of
// the same SIMD?  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 run VALU ops of
// one class.  Prints per-class: MFMA wave cycles alone / together, VALU wave cycles alone / together.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CLS>
__device__ __forceinline__ void valu16(double (&x)[8], int (&u)[8], double c, int ci) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (CLS == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
            if (CLS == 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (CLS == 2) asm volatile("v_cmp_eq_f64 vcc, %0, %1" ::"v"(x[i]), "v"(c) : "vcc");
            if (CLS == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ci) : "vcc");
            if (CLS == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ci));
            if (CLS == 5) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(ci));
            if (CLS == 6) asm volatile("s_nop 3");
            if (CLS == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (CLS == 8) asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ci));
            if (CLS == 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ci));
        }
}

template <int CLS>
__global__ __launch_bounds__(512) void k(int m_iters, int v_iters, long long *cyc, double *sink, int prio) {
    const int wave = threadIdx.x >> 6;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        d4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
        for (int it = 0; it < m_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678) sink[0] = s;
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        double x[8];
        int u[8];
        for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
        double c = 1.0000001; int ci = 7;
        for (int it = 0; it < v_iters; ++it) valu16<CLS>(x, u, c, ci);
        double s = 0; int su = 0;
        for (int i = 0; i < 8; ++i) { s += x[i]; su += u[i]; }
        if (s == 12345.678 || su == 123456789) sink[1] = s + su;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int CLS, int PRIO>
void run(const char *name) {
    long long *cyc; double *sink;
    hipMalloc(&cyc, 8 * sizeof(long long)); hipMalloc(&sink, 16);
    long long h[8];
    const int M = 4000, V = 4000;   // 16 MFMAs / 16 VALU ops per iteration
    double res[3][2];
    int cfg[3][2] = {{M, 0}, {0, V}, {M, V}};
    for (int c = 0; c < 3; ++c) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k<CLS>, dim3(256), dim3(512), 0, 0, cfg[c][0], cfg[c][1], cyc, sink, PRIO);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        res[c][0] = (double)h[0]; res[c][1] = (double)h[4];
    }
    // s_memtime counts at 100 MHz here (constant clock); report relative numbers and per-op ticks
    printf("%-14s mfma alone %8.0f  valu alone %8.0f | together: mfma %8.0f (x%.3f) valu %8.0f (x%.2f) | valu ticks/op alone %.4f together %.4f | mfma ticks/op %.4f\n",
           name, res[0][0], res[1][1], res[2][0], res[2][0] / res[0][0], res[2][1], res[2][1] / res[1][1],
           res[1][1] / (16.0 * V), res[2][1] / (16.0 * V), res[0][0] / (16.0 * M));
    hipFree(cyc); hipFree(sink);
}

int main() {
    run<0, 0>("v_fma_f64 p0"); run<0, 1>("v_fma_f64 p3"); run<4, 0>("v_add_u32 p0"); run<4, 1>("v_add_u32 p3");
    return 0;
}
