// Accuracy of pm_pow_pos (prosper_amd/csrc/pm_common.h) against libm pow on the device and long double on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "../prosper_amd/csrc/pm_common.h"
__global__ void k(const double* x, double* y, double* yl, double c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { y[i] = pm_pow_pos(x[i], c); yl[i] = pow(x[i], c); }
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), y(n), yl(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> ue(-280, 280);
    for (int i = 0; i < n; ++i) x[i] = std::pow(10.0, ue(g));
    x[0] = 0.0; x[1] = 1.0; x[2] = 0.5; x[3] = 2.0;
    double *dx, *dy, *dl;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dl, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (double c : {1.0 / 21.0, 1.0 / 6.0, 1.0 / 1.05, 1.0 / 35.0}) {
        k<<<n / 256, 256>>>(dx, dy, dl, c, n);
        hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(yl.data(), dl, n * 8, hipMemcpyDeviceToHost);
        double worst = 0, worstl = 0;
        for (int i = 1; i < n; ++i) {
            long double ref = powl((long double)x[i], (long double)c);
            worst = fmax(worst, (double)fabsl((y[i] - ref) / ref));
            worstl = fmax(worstl, (double)fabsl((yl[i] - ref) / ref));
        }
        printf("c=%.5f  max rel err pm_pow_pos %.3g   libm pow %.3g   pow(0)=%g\n", c, worst, worstl, y[0]);
    }
    return 0;
}
