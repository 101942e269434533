// f64 VALU issue rate on one CU: a workgroup of W waves per SIMD runs R rounds of 32 independent v_fma_f64
// (or v_mul_f64 / v_add_f64 / v_fma_f32) per lane; cycles per wave-instruction per SIMD from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(double *out, int rounds, long long *cyc) {
    double a[16], b = threadIdx.x * 1e-9 + 1.0, c = 0.5;
    for (int i = 0; i < 16; ++i) a[i] = i + threadIdx.x;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 3) { float f = (float)a[i]; asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f) : "v"((float)b)); a[i] = f; }
                if (OP == 4) asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            }
    }
    long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> void run(const char *name) {
    double *out; long long *cyc; hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    for (int threads : {64, 256, 512, 1024}) {
        const int rounds = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, rounds, cyc);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, rounds, cyc);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double waves_per_simd = threads / 256.0 < 1 ? 1 : threads / 256.0;
        const double instr = rounds * 32.0 * waves_per_simd;    // wave-instructions per SIMD
        printf("%-10s threads %4d: %.2f us, %.2f ns per wave-instr per SIMD (s_memtime ticks/instr %.2f)\n", name, threads,
               ms * 1e3, ms * 1e6 / instr, c / instr);
    }
}
int main() { run<0>("fma_f64"); run<4>("fnma_f64"); run<1>("mul_f64"); run<2>("add_f64"); run<3>("fma_f32"); return 0; }
