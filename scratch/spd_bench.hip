// Timing + residual check of the one-workgroup SPD inverse (prosper_amd/csrc/spd_inverse.hip is #included;
// -DPM_SPD_ABL=1..4 are its timing ablations).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../prosper_amd/csrc/spd_inverse.hip"
static double check(int n, int seed) {
    std::vector<double> B(n * n), A(n * n), X(n * n);
    srand(seed);
    for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double s = 0; for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k];
        A[i * n + j] = s + (i == j ? 0.5 : 0.0);
    }
    double *u, *full, *inv, *piv;
    hipMalloc(&u, n * n * 8); hipMalloc(&full, n * n * 8); hipMalloc(&inv, n * n * 8); hipMalloc(&piv, 16);
    hipMemcpy(u, A.data(), n * n * 8, hipMemcpyHostToDevice);
    pm_spd_inverse_f64(u, n, nullptr, n, full, inv, n, piv, nullptr);
    hipMemcpy(X.data(), inv, n * n * 8, hipMemcpyDeviceToHost);
    double pv[2]; hipMemcpy(pv, piv, 16, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double s = 0; for (int k = 0; k < n; ++k) s += A[i * n + k] * X[k * n + j];
        worst = fmax(worst, fabs(s - (i == j)));
    }
    printf("n=%3d residual max|A.inv - I| = %.3e  pivots %.3e %.3e\n", n, worst, pv[0], pv[1]);
    if (worst > 1e-8) {
        std::vector<double> X0(n * n);
        hipLaunchKernelGGL(spd_inverse_kernel, dim3(1), dim3(1024), 0, 0, u, (int64_t)n, (const double *)nullptr, n, full, inv, (int64_t)n, piv, (int64_t)0, (int64_t)0, (const double *)nullptr, 0, 0.0, (const double *)nullptr);
        hipMemcpy(X0.data(), inv, n * n * 8, hipMemcpyDeviceToHost);
        int cnt = 0;
        for (int i = 0; i < n && cnt < 40; ++i) for (int j = i; j < n && cnt < 40; ++j)
            if (fabs(X0[i * n + j] - X[i * n + j]) > 1e-9 * (1 + fabs(X0[i * n + j]))) { printf("  (%d,%d) old %.6e new %.6e\n", i, j, X0[i * n + j], X[i * n + j]); ++cnt; }
    }
    hipFree(u); hipFree(full); hipFree(inv); hipFree(piv);
    return worst;
}
int main() {
    for (int n : {1, 2, 31, 32, 33, 64, 100, 128, 200, 255, 256}) check(n, n);
    const int n = 256;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? n : 0.0) + 1.0 / (1 + abs(i - j));
    double *u, *full, *inv, *piv;
    hipMalloc(&u, n * n * 8); hipMalloc(&full, n * n * 8); hipMalloc(&inv, n * n * 8); hipMalloc(&piv, 16);
    hipMemcpy(u, A.data(), n * n * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int m : {256, 128}) for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) pm_spd_inverse_f64(u, n, nullptr, m, full, inv, n, piv, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("spd_inverse n=%d: %.1f us / call\n", m, ms / 20 * 1e3);
    }
    {   // warm start from the exact inverse of the same matrix: the Newton-Schulz path
        double *work, *prev; hipMalloc(&work, pm_spd_inverse_warm_work_len(n) * 8); hipMalloc(&prev, n * n * 8);
        pm_spd_inverse_f64(u, n, nullptr, n, full, prev, n, piv, nullptr);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i) pm_spd_inverse_warm_f64(u, n, nullptr, n, prev, n, work, full, inv, n, piv, nullptr);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double pv[2]; hipMemcpy(pv, piv, 16, hipMemcpyDeviceToHost);
            printf("spd_inverse_warm n=%d: %.1f us / call (pivots %.1f %.1f)\n", n, ms / 20 * 1e3, pv[0], pv[1]);
        }
    }
    return 0;
}
