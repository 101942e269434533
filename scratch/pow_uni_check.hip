// Accuracy of pm_pow_uni (prosper_amd/csrc/pm_common.h) against long double on the host, beside pm_pow_tab and libm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "../prosper_amd/csrc/pm_common.h"
__global__ void k(const double* x, double* y, double* yt, double c, int n) {
    __shared__ __attribute__((aligned(16))) double ut[256];
    __shared__ double ab[PM_UPOW_AB_LEN];
    __shared__ double tab[PM_POWTAB_LEN];
    pm_load_upow(ut, ab, pm_powtab_dev, c, threadIdx.x, blockDim.x);
    pm_load_powtab(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { y[i] = pm_pow_uni(x[i], ut, ab); yt[i] = pm_pow_tab(x[i], c, tab); }
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), y(n), yt(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> ue(-84, 90);
    for (int i = 0; i < n; ++i) x[i] = std::pow(10.0, ue(g));
    x[1] = 1.0; x[2] = 0.5; x[3] = 2.0; x[4] = 1e-300; x[5] = 1e300;
    double *dx, *dy, *dl;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dl, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (double rho : {21.0, 6.0, 2.0, 3.0, 4.3333333333333, 11.7, 1.0 / (1.0 - 1.0 / 1.971), 35.0, 1.2}) {
        const double c = 1.0 / rho - 1.0;
        k<<<n / 256, 256>>>(dx, dy, dl, c, n);
        hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(yt.data(), dl, n * 8, hipMemcpyDeviceToHost);
        double worst = 0, worstt = 0;
        for (int i = 1; i < n; ++i) {
            long double ref = powl((long double)x[i], (long double)c);
            worst = fmax(worst, (double)fabsl((y[i] - ref) / ref));
            worstt = fmax(worstt, (double)fabsl((yt[i] - ref) / ref));
        }
        printf("rho=%.5f c=%.6f  max rel err pm_pow_uni %.3g   pm_pow_tab %.3g\n", rho, c, worst, worstt);
    }
    return 0;
}
