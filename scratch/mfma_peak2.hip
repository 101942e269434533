// f64 MFMA issue-rate / clock microbenchmark with in-kernel stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mf(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = d4{0, 0, 0, 0};
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = a0 + threadIdx.x * 1e-9 + i; b[i] = b0 * (i + 1) + threadIdx.x * 1e-3; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u & 3], b[(u >> 2) & 3], acc[u], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int bpc, int iters, double* out, unsigned long long* st, double bscale) {
    int blocks = 256 * bpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mf<NACC>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0, bscale);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(mf<NACC>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0, bscale);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); double t = ms / 10 * 1e-3;
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[2 * b] / ((double)iters * NACC)); clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    double flops = blocks * 4.0 * iters * NACC * 2048.0;
    printf("NACC=%2d blocks/CU=%d data=%g: %.1f TF/s   cycles/MFMA/wave median %.1f   clock median %.0f MHz\n", NACC, bpc, bscale,
           flops / t / 1e12, cyc[cyc.size() / 2], clk[clk.size() / 2]);
}

int main() {
    double* out; hipMalloc(&out, sizeof(double) * 256 * 256 * 16);
    unsigned long long* st; hipMalloc(&st, sizeof(unsigned long long) * 2 * 256 * 16);
    for (double bs : {0.0, 1.37}) {
        for (int bpc : {1, 2}) {
            run<4>(bpc, 4000, out, st, bs);
            run<8>(bpc, 2000, out, st, bs);
            run<16>(bpc, 1000, out, st, bs);
        }
    }
    // long run to let DVFS settle
    for (int r = 0; r < 3; ++r) run<16>(2, 20000, out, st, 1.37);
    return 0;
}
