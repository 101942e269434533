// Device helpers shared by the gfx950 kernels: 64-lane wavefront reductions and f64 atomics.
#ifndef PM_COMMON_H
#define PM_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PM_WAVE 64

// global_atomic_add_f64 (no return): agent scope, relaxed.  Built with -munsafe-fp-atomics so
// this lowers to the hardware instruction, not a compare-and-swap loop.
__device__ __forceinline__ void pm_atomic_add(double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double pm_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, PM_WAVE);
    return v;
}

__device__ __forceinline__ double pm_wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, PM_WAVE));
    return v;
}

// Wave-wide argmax of (value, index): larger value wins, ties go to the larger index.
__device__ __forceinline__ void pm_wave_argmax(double &v, int &idx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off, PM_WAVE);
        const int oi = __shfl_xor(idx, off, PM_WAVE);
        const bool take = (ov > v) || (ov == v && oi > idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    }
}

// Packed BSC statistics buffer: [ Wp (H*D) | Wq (H*H) | qdiag (H) | mus (H) | scalars ]
__host__ __device__ inline int64_t pm_bsc_stats_offset_wq_dev(int64_t H, int64_t D) { return H * D; }
__host__ __device__ inline int64_t pm_bsc_stats_offset_qdiag_dev(int64_t H, int64_t D) { return H * D + H * H; }
__host__ __device__ inline int64_t pm_bsc_stats_offset_mus_dev(int64_t H, int64_t D) { return H * D + H * H + H; }
__host__ __device__ inline int64_t pm_bsc_stats_offset_scalars_dev(int64_t H, int64_t D) { return H * D + H * H + 2 * H; }

#endif  // PM_COMMON_H
