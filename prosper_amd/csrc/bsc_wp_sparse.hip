// my_Wp of BSC_ET.M_step (prosper/em/camodels/bsc_et.py:339-363) from the non-zero lists of E[s].
//
// The reference adds, per datapoint, np.outer(E[s], y) into the (H, D) accumulator.  Dense, that is the 2 N H D flop GEMM
// E[s]^T Y (pm_gemm_tn_acc_f64: 1.49 ms of a 3.6 ms EM iteration on config 2, at 90 % of the f64 MFMA peak).  But past the
// annealing phase the truncated posterior has weight on a handful of latents: 3.7 non-zeros of 256 per row on config 2
// (max 8; scratch/nnz_hist.py), so 98.5 % of those flops multiply zeros.  With the lists the E-step pass leaves
// (pm_bsc_estep_fused8_nz_f64) the product is N nnz D multiply-adds -- nothing -- and one read of the data: the kernel
// is bound by streaming Y once (N D 8 bytes; 1.64 GB on config 2 = 0.2 ms at the HBM peak, 0.27 ms at the 6 TB/s a
// plain reduction over Y reaches on this chip; measured 0.40 ms = 4.1 TB/s -- each workgroup reads 512-byte pieces of
// 8 KB rows; with empty lists the same loop streams at 4.8 TB/s, scratch/sparse_bench.py; reading a chunk-major copy of
// Y instead -- one contiguous stream per workgroup -- changes nothing, 0.393 vs 0.392 ms: the access pattern is not it).
//
// Layout: a workgroup owns a 64-column chunk of Wp for a group of datapoints and keeps its (H x 64) accumulator in LDS
// (128 KB at H = 256: one 16-wavefront workgroup per CU).  A wavefront takes one datapoint at a time: lane c holds
// y[n, c0 + c], lanes 0..15 hold the list; every non-zero is one ds_add_f64 of 64 lanes into the accumulator row of its
// latent (a row is 512 contiguous bytes: all banks once).  The accumulators are flushed with f64 atomics; all
// workgroups of a column chunk sit on one XCD, so a Wp line is only ever touched through one L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

constexpr int SP_DC = 64;          // columns per workgroup
constexpr int SP_WAVES = 16;
#ifndef PM_SP_UNROLL
#define PM_SP_UNROLL 8
#endif
constexpr int SP_UNROLL = PM_SP_UNROLL;   // datapoints per batch; two batches in flight per wavefront (16 x 512 B of Y)

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(SP_WAVES * 64) void bsc_wp_sparse_kernel(const uint16_t *__restrict__ nz_idx,
                                                                      const double *__restrict__ nz_val,
                                                                      const double *__restrict__ Y, int64_t ldy,
                                                                      double *__restrict__ Wp, int64_t ldw,
                                                                      const double *__restrict__ gate, int64_t N, int H,
                                                                      int D, int nchunks, int64_t rows_per_group) {
    extern __shared__ __attribute__((aligned(16))) double acc[];          // [H][SP_DC]
    if (*gate != 0.0) return;                  // some list overflowed: the dense product runs instead
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups are dealt round-robin over the 8 XCDs: chunk = 8 k + (blockIdx & 7) keeps a chunk on one XCD (fewer
    // than 8 chunks -- D <= 448 -- : plain enumeration, every workgroup has work)
    int chunk;
    int64_t grp;
    if (nchunks >= 8) {
        const int per8 = (nchunks + 7) >> 3;
        const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
        chunk = x + 8 * (q % per8);
        grp = q / per8;
    } else {
        chunk = blockIdx.x % nchunks;
        grp = blockIdx.x / nchunks;
    }
    if (chunk >= nchunks) return;
    for (int i = tid; i < H * SP_DC; i += SP_WAVES * 64) acc[i] = 0.0;
    __syncthreads();

    // (no predication in the loop: columns past D and datapoints past the group are clamped to valid addresses -- the
    // former are never flushed, the latter get an empty list)
    const uint32_t col = chunk * SP_DC + lane < D ? chunk * SP_DC + lane : D - 1;
    const uint32_t slot = lane & (PM_BSC_NZ_MAX - 1);
    const int lo = (int)(grp * rows_per_group);                // (N < 2^31: checked by the caller)
    const int hi = lo + rows_per_group < N ? (int)(lo + rows_per_group) : (int)N;
    double *arow = acc + lane;
    struct Batch {
        double y[SP_UNROLL], v[SP_UNROLL];
        int ix[SP_UNROLL];
        bool ok[SP_UNROLL];
    };
    auto load = [&](int n0, Batch &b) {
#pragma unroll
        for (int u = 0; u < SP_UNROLL; ++u) {
            int n = n0 + u * SP_WAVES;                         // wavefront-uniform: scalar base + 32-bit lane offset
            b.ok[u] = n < hi;
            n = b.ok[u] ? n : hi - 1;
#ifdef PM_SP_NO_NT
            b.y[u] = (Y + (int64_t)n * ldy)[col];
#else
            b.y[u] = __builtin_nontemporal_load((Y + (int64_t)n * ldy) + col);   // read once: leave L2 to the lists
#endif
            b.ix[u] = (nz_idx + (int64_t)n * PM_BSC_NZ_MAX)[slot];
            b.v[u] = (nz_val + (int64_t)n * PM_BSC_NZ_MAX)[slot];
        }
    };
    auto accumulate = [&](const Batch &b) {
#pragma unroll
        for (int u = 0; u < SP_UNROLL; ++u) {
            // (the valid slots are the leading ones; lanes 16.. repeat lanes 0..15)
            const int cnt = b.ok[u] ? __popc((uint32_t)__ballot(b.ix[u] != 0xFFFF) & 0xFFFFu) : 0;
            for (int t = 0; t < cnt; ++t) {
                const int h = __builtin_amdgcn_readlane(b.ix[u], t);
                const double w = readlane_f64(b.v[u], t);
                atomicAdd(arow + h * SP_DC, w * b.y[u]);
            }
        }
    };
    // two batches in flight: the loads of one are issued before the other is accumulated
    constexpr int STEP = SP_WAVES * SP_UNROLL;
    Batch b0, b1;
    load(lo + wave, b0);
    for (int n0 = lo + wave; n0 < hi; n0 += 2 * STEP) {
        load(n0 + STEP, b1);
        accumulate(b0);
        load(n0 + 2 * STEP, b0);
        accumulate(b1);
    }
    __syncthreads();
    for (int i = tid; i < H * SP_DC; i += SP_WAVES * 64) {
        const double a = acc[i];
        const int h = i / SP_DC, c = chunk * SP_DC + (i % SP_DC);
        if (a != 0.0 && c < D) pm_atomic_add(Wp + (int64_t)h * ldw + c, a);
    }
}

}  // namespace

extern "C" int pm_bsc_wp_sparse_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy,
                                    double *stats, int64_t N, int64_t H, int64_t D, void *stream) {
    if (!stats) return PM_EINVAL;
    return pm_wp_sparse_f64(nz_idx, nz_val, Y, ldy, stats, D, stats + pm_bsc_stats_offset_scalars_dev(H, D) + 3, N, H, D,
                            stream);
}

extern "C" int pm_wp_sparse_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *Wp,
                                int64_t ldw, const double *gate, int64_t N, int64_t H, int64_t D, void *stream) {
    if (!nz_idx || !nz_val || !Y || !Wp || !gate || N < 0 || H <= 0 || D <= 0 || ldy < D || ldw < D) return PM_EINVAL;
    if (H > 256 || D > INT32_MAX || N > INT32_MAX - 4096) return PM_ERANGE;
    if (N == 0) return PM_OK;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    const int nchunks = (int)((D + SP_DC - 1) / SP_DC), per8 = (nchunks + 7) / 8;
    const int slots = nchunks >= 8 ? 8 * per8 : nchunks;       // workgroups per group of datapoints (see the kernel)
    // one workgroup per CU: groups of datapoints so that slots x groups covers the CUs, whole unroll rounds each
    int64_t groups = cus / slots;      // (H <= 128, 64 KB accumulators, two workgroups per CU: slower -- twice the flush)
    if (groups < 1) groups = 1;
    const int64_t round = SP_WAVES * SP_UNROLL;
    int64_t rpg = (N + groups - 1) / groups;
    rpg = (rpg + round - 1) / round * round;
    groups = (N + rpg - 1) / rpg;
    const size_t shmem = sizeof(double) * (size_t)H * SP_DC;
    if (int e = (int)hipFuncSetAttribute(reinterpret_cast<const void *>(bsc_wp_sparse_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
        return e;
    hipLaunchKernelGGL(bsc_wp_sparse_kernel, dim3((unsigned)(slots * groups)), dim3(SP_WAVES * 64), shmem,
                       static_cast<hipStream_t>(stream), nz_idx, nz_val, Y, ldy, Wp, ldw, gate, N, (int)H, (int)D, nchunks,
                       rpg);
    return (int)hipGetLastError();
}
