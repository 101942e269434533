// Throughput of LDS atomics on gfx950: ds_add_f64 vs ds_add_u64 vs ds_add_f32 vs ds_add_u32 vs plain read-modify-write,
// conflict-free addresses (lane l -> slot l of a row), 16 wavefronts per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, int iters, int rows) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < rows * 64; i += 1024) sm[i] = 0.0;
    __syncthreads();
    unsigned r = wave * 7 + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r = r * 1664525u + 1013904223u;
            const int row = (r >> 8) % rows;       // wave-uniform row, lane = column: conflict-free
            if (MODE == 0) atomicAdd(&sm[row * 64 + lane], 1.0);
            if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long *>(&sm[row * 64 + lane]), 1ull);
            if (MODE == 2) atomicAdd(reinterpret_cast<float *>(sm) + row * 64 + lane, 1.0f);
            if (MODE == 3) atomicAdd(reinterpret_cast<unsigned *>(sm) + row * 64 + lane, 1u);
            if (MODE == 4) sm[row * 64 + lane] += 1.0;     // (racy: timing only)
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = sm[5];
}
template <int MODE>
void run(const char *name) {
    double *out;
    hipMalloc(&out, 8 * 1024);
    const int iters = 2000, rows = 128;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, rows * 64 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 1024, rows * 64 * 8>>>(out, 10, rows);
    hipEventRecord(a);
    k<MODE><<<256, 1024, rows * 64 * 8>>>(out, iters, rows);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = 16.0 * iters * 8;
    printf("%-14s %8.3f ms  -> %6.1f clk per wave-instruction per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}
int main() {
    run<0>("ds_add_f64"); run<1>("ds_add_u64"); run<2>("ds_add_f32"); run<3>("ds_add_u32"); run<4>("rmw f64");
    return 0;
}
