// Distributed k-th largest of N doubles by radix select: the data-truncation cut of the M-step,
//   cut_denom = parallel.allsort(all_denoms)[-N_use]        (bsc_et.py:252, utils/parallel.py:87-110)
// The reference all-gathers all N values to every rank and merge-sorts them; only one order statistic is consumed.
// Here every rank histograms the next digit of an order-preserving 64-bit key over its own shard (values that still
// match the digits decided so far), the 4096 bins are all-reduced (32 KB instead of 8 N bytes), and a one-workgroup
// scan picks the bin holding the k-th largest and updates (prefix, k).  Six rounds (12 + 12 + 12 + 12 + 12 + 4 bits)
// pin all 64 bits: the result is EXACTLY the value a full sort would return.  Nothing but the final double ever
// reaches the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

constexpr int BINS = 4096;

// monotone map double -> uint64: larger double <=> larger key (-0.0 < +0.0; NaN above +inf / below -inf by sign)
__device__ __forceinline__ uint64_t key_of(double x) {
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

// hist[d] += number of x[i] whose key agrees with state[0] above bit shift+bits and has digit d at [shift, shift+bits)
__global__ __launch_bounds__(256) void kth_hist_kernel(const double *__restrict__ x, int64_t n,
                                                        const unsigned long long *__restrict__ state, int shift,
                                                        int bits, unsigned long long *__restrict__ hist) {
    __shared__ unsigned int s_h[BINS];
    for (int b = threadIdx.x; b < BINS; b += 256) s_h[b] = 0;
    __syncthreads();
    const int top = shift + bits;                     // bits [top, 64) are decided
    const uint64_t prefix = state[0];
    const uint64_t mask = (1ull << bits) - 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t k = key_of(x[i]);
        const bool match = (top >= 64) || ((k >> top) == (prefix >> top));
        if (match) atomicAdd(&s_h[(k >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += 256) {
        const unsigned int c = s_h[b];
        if (c) atomicAdd(&hist[b], (unsigned long long)c);
    }
}

// one workgroup: the digit d (scanning from the top) with count(digits > d) < k <= count(digits >= d);
// state[0] |= d << shift, state[1] = k - count(digits > d); the histogram is cleared for the next round
__global__ __launch_bounds__(256) void kth_scan_kernel(unsigned long long *__restrict__ hist,
                                                        unsigned long long *__restrict__ state, int shift, int bits) {
    __shared__ unsigned long long s_chunk[256];
    __shared__ int s_pick;
    __shared__ unsigned long long s_above;
    const int nb = 1 << bits, per = (nb + 255) / 256;
    const int t = threadIdx.x;
    unsigned long long mine[BINS / 256];
    unsigned long long sum = 0;
    for (int q = 0; q < per; ++q) {
        const int b = t * per + q;
        mine[q] = (b < nb) ? hist[b] : 0ull;
        sum += mine[q];
    }
    s_chunk[t] = sum;
    __syncthreads();
    if (t == 0) {
        const unsigned long long k = state[1];
        unsigned long long above = 0;
        int c = 255;
        for (; c > 0; --c) {
            if (above + s_chunk[c] >= k) break;
            above += s_chunk[c];
        }
        s_pick = c;
        s_above = above;
    }
    __syncthreads();
    if (t == s_pick) {
        const unsigned long long k = state[1];
        unsigned long long above = s_above;
        int q = per - 1;
        for (; q > 0; --q) {
            if (above + mine[q] >= k) break;
            above += mine[q];
        }
        const unsigned long long d = (unsigned long long)(t * per + q);
        state[0] |= d << shift;
        state[1] = k - above;
    }
    for (int q = 0; q < per; ++q) {
        const int b = t * per + q;
        if (b < nb) hist[b] = 0ull;
    }
}

// ---- round 6: the scan folded into the NEXT round's histogram kernel -------------------------------------------------------
// A select used to be 6 x (histogram, scan) + a conversion = 13 launches of a few microseconds of work each: 0.17 ms of a
// data-truncation step once that step had no other host round trip left.  Every workgroup of round r now first repeats the
// scan of round r - 1 for itself (the all-reduced histogram of 4096 bins and the state it started from: ~2 us, the same
// answer in every workgroup), workgroup 0 records the result for the round after, and the histograms of the six rounds are
// six separate buffers zeroed by ONE fill: 1 + 6 + 1 launches.  (One cooperative launch for the single-rank case -- the rounds
// separated by grid barriers on a global counter -- was built and measured: 0.14 ms against 0.09, seven barriers across eight
// XCDs cost more than seven launch gaps; not kept.)
__device__ __forceinline__ void kth_scan_block(const unsigned long long *__restrict__ hist,
                                               const unsigned long long *__restrict__ state_in, int shift, int bits,
                                               unsigned long long *s_chunk, int *s_pick, unsigned long long *s_out) {
    const int nb = 1 << bits, per = (nb + 255) / 256;
    const int t = threadIdx.x;
    unsigned long long mine[BINS / 256];
    unsigned long long sum = 0;
    for (int q = 0; q < per; ++q) {
        const int b = t * per + q;
        mine[q] = (b < nb) ? hist[b] : 0ull;
        sum += mine[q];
    }
    s_chunk[t] = sum;
    __syncthreads();
    const unsigned long long k = state_in[1];
    if (t < 64) {
        // suffix sums of the 256 chunk totals, four chunks per lane + a wavefront scan (not 256 dependent LDS reads)
        const unsigned long long c0 = s_chunk[4 * t], c1 = s_chunk[4 * t + 1], c2 = s_chunk[4 * t + 2], c3 = s_chunk[4 * t + 3];
        unsigned long long tot = c0 + c1 + c2 + c3, suf = tot;             // inclusive suffix over lanes >= t
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_down(suf, off);
            if (t + off < 64) suf += o;
        }
        const unsigned long long above_lane = suf - tot;                   // chunks of lanes > t
        // the HIGHEST chunk c with above(c) + chunk[c] >= k (above(c): everything in higher chunks); chunk 0 if none --
        // exactly the sequential scan of kth_scan_kernel
        unsigned long long above = above_lane, my_above = 0;
        int myq = -1;
        const unsigned long long cs[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int q = 3; q >= 0; --q) {
            if (myq < 0 && above + cs[q] >= k) {
                myq = q;
                my_above = above;
            }
            above += cs[q];
        }
        const unsigned long long bal = __ballot(myq >= 0);
        if (bal) {
            const int top = 63 - __builtin_clzll(bal);
            if (t == top) {
                *s_pick = 4 * t + myq;
                s_out[1] = my_above;
            }
        } else if (t == 0) {
            *s_pick = 0;
            s_out[1] = above - c0;
        }
    }
    __syncthreads();
    if (t == *s_pick) {
        unsigned long long above = s_out[1];
        int q = per - 1;
        for (; q > 0; --q) {
            if (above + mine[q] >= k) break;
            above += mine[q];
        }
        const unsigned long long d = (unsigned long long)(t * per + q);
        s_out[0] = state_in[0] | (d << shift);
        s_out[1] = k - above;
    }
    __syncthreads();
}

// states: 7 slots of (prefix, k); slot 0 = (0, k) from the caller, slot r = the state after the scan of round r - 1.
// hists: 6 x BINS, zero on entry of round 0.  Round r: scan hists[r - 1] (r > 0), then histogram this round's digit.
__global__ __launch_bounds__(256) void kth_round_kernel(const double *__restrict__ x, int64_t n,
                                                         unsigned long long *__restrict__ states,
                                                         unsigned long long *__restrict__ hists, int round,
                                                         int shift_prev, int bits_prev, int shift, int bits, long long k0) {
    __shared__ unsigned int s_h[BINS];
    __shared__ unsigned long long s_chunk[256];
    __shared__ int s_pick;
    __shared__ unsigned long long s_state[2];
    for (int b = threadIdx.x; b < BINS; b += 256) s_h[b] = 0;
    if (threadIdx.x == 0) s_pick = 0;
    if (round > 0) {
        kth_scan_block(hists + (size_t)(round - 1) * BINS, states + 2 * (round - 1), shift_prev, bits_prev, s_chunk, &s_pick,
                       s_state);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            states[2 * round] = s_state[0];
            states[2 * round + 1] = s_state[1];
        }
    } else {
        // (k0 >= 0: the rank comes as an argument -- nothing has to be written into the buffer in front of a select, whose
        // histograms the previous select's final kernel has left zeroed; workgroup 0 records it for the scan of round 1)
        if (threadIdx.x < 2) s_state[threadIdx.x] = k0 >= 0 ? (threadIdx.x ? (unsigned long long)k0 : 0ull) : states[threadIdx.x];
        if (k0 >= 0 && blockIdx.x == 0 && threadIdx.x < 2) states[threadIdx.x] = threadIdx.x ? (unsigned long long)k0 : 0ull;
        __syncthreads();
    }
    const int top = shift + bits;                     // bits [top, 64) are decided
    const uint64_t prefix = s_state[0];
    const uint64_t mask = (1ull << bits) - 1;
    unsigned long long *hist = hists + (size_t)round * BINS;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t k = key_of(x[i]);
        const bool match = (top >= 64) || ((k >> top) == (prefix >> top));
        if (match) atomicAdd(&s_h[(k >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += 256) {
        const unsigned int c = s_h[b];
        if (c) atomicAdd(&hist[b], (unsigned long long)c);
    }
}

// the last round's scan and the conversion back to a double
__global__ __launch_bounds__(256) void kth_final_kernel(unsigned long long *__restrict__ states,
                                                         unsigned long long *__restrict__ hists, int rounds,
                                                         int shift_prev, int bits_prev, double *__restrict__ out, int rezero) {
    __shared__ unsigned long long s_chunk[256];
    __shared__ int s_pick;
    __shared__ unsigned long long s_state[2];
    if (threadIdx.x == 0) s_pick = 0;
    kth_scan_block(hists + (size_t)(rounds - 1) * BINS, states + 2 * (rounds - 1), shift_prev, bits_prev, s_chunk, &s_pick,
                   s_state);
    if (threadIdx.x == 0) {
        states[2 * rounds] = s_state[0];
        states[2 * rounds + 1] = s_state[1];
        out[0] = value_of(s_state[0]);
    }
    // (rezero: the histograms are left zeroed for the next select on this buffer -- no fill launch in front of it)
    if (rezero) {
        __syncthreads();
        for (int i = threadIdx.x; i < rounds * BINS; i += 256) hists[i] = 0ull;
    }
}

__global__ void kth_value_kernel(const unsigned long long *__restrict__ state, double *__restrict__ out) {
    out[0] = value_of(state[0]);
}

}  // namespace

extern "C" int pm_kth_hist_f64(const double *x, int64_t n, const uint64_t *state, int shift, int bits, uint64_t *hist,
                               void *stream) {
    if (n < 0 || !state || !hist || shift < 0 || bits < 1 || bits > 12 || shift + bits > 64 || (n > 0 && !x))
        return PM_EINVAL;
    if (n == 0) return PM_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(kth_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, n,
                       reinterpret_cast<const unsigned long long *>(state), shift, bits,
                       reinterpret_cast<unsigned long long *>(hist));
    return (int)hipGetLastError();
}

extern "C" int pm_kth_scan(uint64_t *hist, uint64_t *state, int shift, int bits, void *stream) {
    if (!hist || !state || shift < 0 || bits < 1 || bits > 12 || shift + bits > 64) return PM_EINVAL;
    hipLaunchKernelGGL(kth_scan_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<unsigned long long *>(hist), reinterpret_cast<unsigned long long *>(state), shift,
                       bits);
    return (int)hipGetLastError();
}

extern "C" int pm_kth_value_f64(const uint64_t *state, double *out, void *stream) {
    if (!state || !out) return PM_EINVAL;
    hipLaunchKernelGGL(kth_value_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned long long *>(state), out);
    return (int)hipGetLastError();
}

extern "C" int pm_kth_round_f64(const double *x, int64_t n, uint64_t *states, uint64_t *hists, int round, int shift_prev,
                                int bits_prev, int shift, int bits, void *stream) {
    return pm_kth_round_k_f64(x, n, states, hists, round, shift_prev, bits_prev, shift, bits, -1, stream);
}

extern "C" int pm_kth_round_k_f64(const double *x, int64_t n, uint64_t *states, uint64_t *hists, int round, int shift_prev,
                                  int bits_prev, int shift, int bits, int64_t k0, void *stream) {
    if (k0 >= 0 && round != 0) return PM_EINVAL;
    if (n < 0 || !states || !hists || round < 0 || round > 5 || shift < 0 || bits < 1 || bits > 12 || shift + bits > 64 ||
        (n > 0 && !x) || (round > 0 && (shift_prev < 0 || bits_prev < 1 || bits_prev > 12 || shift_prev + bits_prev > 64)))
        return PM_EINVAL;
    // (an empty shard still walks the rounds: workgroup 0 carries the state forward for the ranks that hold data)
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kth_round_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, n,
                       reinterpret_cast<unsigned long long *>(states), reinterpret_cast<unsigned long long *>(hists), round,
                       shift_prev, bits_prev, shift, bits, (long long)k0);
    return (int)hipGetLastError();
}

extern "C" int pm_kth_final_f64(uint64_t *states, const uint64_t *hists, int rounds, int shift_prev, int bits_prev, double *out,
                                void *stream) {
    return pm_kth_final_z_f64(states, const_cast<uint64_t *>(hists), rounds, shift_prev, bits_prev, out, 0, stream);
}

extern "C" int pm_kth_final_z_f64(uint64_t *states, uint64_t *hists, int rounds, int shift_prev, int bits_prev, double *out,
                                  int rezero, void *stream) {
    if (!states || !hists || !out || rounds < 1 || rounds > 6 || shift_prev < 0 || bits_prev < 1 || bits_prev > 12 ||
        shift_prev + bits_prev > 64)
        return PM_EINVAL;
    hipLaunchKernelGGL(kth_final_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<unsigned long long *>(states), reinterpret_cast<unsigned long long *>(hists),
                       rounds, shift_prev, bits_prev, out, rezero);
    return (int)hipGetLastError();
}
