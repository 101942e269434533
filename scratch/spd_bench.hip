// Timing ablations of the one-workgroup SPD inverse (prosper_amd/csrc/spd_inverse.hip is #included;
// -DABL=1 no division, 2 no ci/cj LDS reads, 3 no publish, 4 no barrier).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../prosper_amd/csrc/spd_inverse.hip"
int main() {
    const int n = 256;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? n : 0.0) + 1.0 / (1 + abs(i - j));
    double *u, *full, *inv, *piv;
    hipMalloc(&u, n * n * 8); hipMalloc(&full, n * n * 8); hipMalloc(&inv, n * n * 8); hipMalloc(&piv, 16);
    hipMemcpy(u, A.data(), n * n * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) pm_spd_inverse_f64(u, n, nullptr, n, full, inv, n, piv, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("spd_inverse n=%d: %.1f us / call\n", n, ms / 20 * 1e3);
    }
    return 0;
}
