// accuracy of pm_pow_m20_21 / pm_pow_m5_6 (either seed: -DPM_POW_HWSEED=0|1) against long double on the host:
// hipcc --offload-arch=gfx950 -DPM_POW_HWSEED=1 scratch/pow_hwseed_check.hip -Iprosper_amd/csrc -Iinclude -o /tmp/powhw && /tmp/powhw
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "pm_common.h"
#include "pm_powtab.h"
__global__ void k(const double *x, double *y21, double *y6, int n) {
    __shared__ double rt21[PM_ROOT21_LEN + 1], rt6[PM_ROOT21_LEN + 1];
    pm_load_root21(rt21, pm_powtab_dev, threadIdx.x, blockDim.x);
    pm_load_root6(rt6, pm_powtab_dev, threadIdx.x, blockDim.x);
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        y21[i] = pm_pow_m20_21(x[i], rt21);
        y6[i] = pm_pow_m5_6(x[i], rt6);
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), y21(n), y6(n);
    srand(1);
    for (int i = 0; i < n; ++i) x[i] = exp(-190.0 + 230.0 * (rand() / (double)RAND_MAX)) * (1.0 + rand() / (double)RAND_MAX);
    double *dx, *d21, *d6;
    hipMalloc(&dx, n * 8); hipMalloc(&d21, n * 8); hipMalloc(&d6, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<256, 256>>>(dx, d21, d6, n);
    hipMemcpy(y21.data(), d21, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(y6.data(), d6, n * 8, hipMemcpyDeviceToHost);
    double w21 = 0, w6 = 0;
    for (int i = 0; i < n; ++i) {
        const long double r21 = powl((long double)x[i], -20.0L / 21.0L), r6 = powl((long double)x[i], -5.0L / 6.0L);
        w21 = fmax(w21, fabs((double)(((long double)y21[i] - r21) / r21)));
        w6 = fmax(w6, fabs((double)(((long double)y6[i] - r6) / r6)));
    }
    printf("PM_POW_HWSEED=%d  pm_pow_m20_21: max relative error %.3e   pm_pow_m5_6: %.3e   (%d points in [e^-190, e^41])\n",
           PM_POW_HWSEED, w21, w6, n);
    return (w21 < 5e-16 && w6 < 5e-16) ? 0 : 1;
}
