// accuracy of pm_pow_m5_6 against long double on the host: hipcc --offload-arch=gfx950 scratch/pow6_check.hip -Iprosper_amd/csrc -Iinclude -o /tmp/pow6 && /tmp/pow6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "pm_common.h"
#include "pm_powtab.h"
__global__ void k(const double *x, double *y, int n) {
    __shared__ double rt[PM_ROOT21_LEN + 1];
    pm_load_root6(rt, pm_powtab_dev, threadIdx.x, blockDim.x);
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = pm_pow_m5_6(x[i], rt);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), y(n);
    srand(1);
    for (int i = 0; i < n; ++i) x[i] = exp(-190.0 + 230.0 * (rand() / (double)RAND_MAX)) * (1.0 + rand() / (double)RAND_MAX);
    double *dx, *dy;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<256, 256>>>(dx, dy, n);
    hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) {
        const long double ref = powl((long double)x[i], -5.0L / 6.0L);
        const double e = fabs((double)(((long double)y[i] - ref) / ref));
        if (e > worst) worst = e;
    }
    printf("pm_pow_m5_6: max relative error %.3e over %d points in [e^-190, e^41]\n", worst, n);
    return worst < 4e-16 ? 0 : 1;
}
