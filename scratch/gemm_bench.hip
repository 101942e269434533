// Standalone driver for the GEMM kernels at config-2 shape (links libprosper_hip.so).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "prosper_hip.h"
int main(int argc, char** argv) {
    int64_t N = argc > 1 ? atoll(argv[1]) : 200000, D = 1024, H = 256;
    int reps = argc > 2 ? atoi(argv[2]) : 10;
    double *Y, *W, *A, *E, *Wp;
    hipMalloc(&Y, N * D * 8); hipMalloc(&W, H * D * 8); hipMalloc(&A, N * H * 8); hipMalloc(&E, N * H * 8); hipMalloc(&Wp, H * D * 8);
    std::vector<double> h(N * D);
    srand(1);
    for (auto& v : h) v = (rand() / (double)RAND_MAX) * 2 - 1;
    hipMemcpy(Y, h.data(), N * D * 8, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), H * D * 8, hipMemcpyHostToDevice);
    hipMemcpy(E, h.data(), N * H * 8, hipMemcpyHostToDevice);
    hipMemset(Wp, 0, H * D * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) pm_gemm_nt_f64(Y, D, W, D, A, H, N, H, D, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) pm_gemm_nt_f64(Y, D, W, D, A, H, N, H, D, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("gemm_nt  N=%ld: %.3f ms  %.1f TF/s\n", (long)N, ms, 2.0 * N * D * H / ms / 1e9);
    for (int w = 0; w < 2; ++w) pm_gemm_tn_acc_f64(E, H, Y, D, Wp, D, H, D, N, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) pm_gemm_tn_acc_f64(E, H, Y, D, Wp, D, H, D, N, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("gemm_tn  N=%ld: %.3f ms  %.1f TF/s\n", (long)N, ms, 2.0 * N * D * H / ms / 1e9);
    return 0;
}
