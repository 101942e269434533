// my_Wp of BSC_ET.M_step (prosper/em/camodels/bsc_et.py:339-363) from the non-zero lists of E[s].
//
// The reference adds, per datapoint, np.outer(E[s], y) into the (H, D) accumulator.  Dense, that is the 2 N H D flop GEMM
// E[s]^T Y (pm_gemm_tn_acc_f64: 1.49 ms of a 3.6 ms EM iteration on config 2, at 90 % of the f64 MFMA peak).  But past the
// annealing phase the truncated posterior has weight on a handful of latents: 3.7 non-zeros of 256 per row on config 2
// (max 8; scratch/nnz_hist.py), so 98.5 % of those flops multiply zeros.  With the lists the E-step pass leaves
// (pm_bsc_estep_fused8_nz_f64) the product is N nnz D multiply-adds -- nothing -- and one read of the data: the kernel
// is bound by streaming Y once (N D 8 bytes; 1.64 GB on config 2 = 0.2 ms at the HBM peak, 0.27 ms at the 6 TB/s a
// plain reduction over Y reaches on this chip; measured 0.32-0.33 ms = 5.0 TB/s, with empty lists 0.27 ms = 6.1 TB/s;
// scratch/sparse_bench.py).  The first version ran 0.40 ms: it loaded every datapoint's list with two small loads of its
// own (see the loop).  Reading a chunk-major copy of Y -- one contiguous stream per workgroup instead of 512-byte pieces
// of 8 KB rows -- changed nothing (0.393 vs 0.392 ms): the access pattern was not it.
//
// Layout: a workgroup owns a 64-column chunk of Wp for a group of datapoints and keeps its (H x 64) accumulator in LDS
// (128 KB at H = 256: one 16-wavefront workgroup per CU).  A wavefront takes eight consecutive datapoints per batch:
// lane c holds y[n, c0 + c] of each, the lanes together hold the eight lists; every non-zero is one ds_add_f64 of 64
// lanes into the accumulator row of its latent (a row is 512 contiguous bytes: all banks once).  The accumulators are
// flushed with f64 atomics; all workgroups of a column chunk sit on one XCD, so a Wp line is only ever touched through
// one L2.
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

// The dense rows the list-writing E-step pass did not store (pm_bsc_estep_fused8_nz_f64 keeps the dense E[s] row of a
// datapoint only when its list overflowed), rebuilt from the lists for the dense product: sixteen lanes per datapoint clear
// the row, then lane t scatters entry t (same wavefront, program order).  `group` of `groups` sixteen-lane groups.
__device__ __forceinline__ void expand_listed_rows(const uint16_t *__restrict__ nz_idx, const double *__restrict__ nz_val,
                                                   double *__restrict__ expect, int64_t lde, int64_t N, int H,
                                                   int64_t group, int64_t groups, int t) {
    for (int64_t n = group; n < N; n += groups) {
        if (nz_idx[n * PM_BSC_NZ_MAX] == PM_BSC_NZ_OVERFLOW) continue;       // (its dense row is there)
        double *row = expect + n * lde;
        for (int h = t; h < H; h += 16) row[h] = 0.0;
        __builtin_amdgcn_wave_barrier();
        const uint16_t h = nz_idx[n * PM_BSC_NZ_MAX + t];
        if (h != 0xFFFFu && h < H) row[h] = nz_val[n * PM_BSC_NZ_MAX + t];
    }
}

// (amdgpu_waves_per_eu(6, 6): 80 registers instead of 82 -- two spilled dwords -- so that four of these wavefronts per SIMD
// leave room for ONE workgroup of the gathered f64 GEMM (176 registers a wavefront) on the same CU: GSC's M-step runs the two
// on two streams, an HBM stream beside an MFMA kernel, 1.29 -> 1.24 ms per EM iteration at config 4; at 82 registers the GEMM's
// workgroups wait for a free CU and nothing overlaps)
__global__ __launch_bounds__(SP_WAVES * 64) __attribute__((amdgpu_waves_per_eu(6, 6))) void bsc_wp_sparse_kernel(const uint16_t *__restrict__ nz_idx,
                                                                      const double *__restrict__ nz_val,
                                                                      const double *__restrict__ Y, int64_t ldy,
                                                                      double *__restrict__ Wp, int64_t ldw,
                                                                      const double *__restrict__ gate, int64_t N, int H,
                                                                      int D, int nchunks, int64_t rows_per_group,
                                                                      int transposed, double *__restrict__ expand_to,
                                                                      int64_t lde) {
    extern __shared__ __attribute__((aligned(16))) double acc[];          // [H][SP_DC]
    if (gate && *gate != 0.0) {                // some list overflowed: the dense product runs instead --
        // -- on `expand_to`, whose rows of listed datapoints this launch fills in first (no launch of its own: the
        // gated-off pair of a step costs one empty launch, not two)
        if (expand_to)
            expand_listed_rows(nz_idx, nz_val, expand_to, lde, N, H, (int64_t)blockIdx.x * (SP_WAVES * 4) + (threadIdx.x >> 4),
                               (int64_t)gridDim.x * (SP_WAVES * 4), threadIdx.x & 15);
        return;
    }
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
    const int lo = (int)(grp * rows_per_group);                // (N < 2^31: checked by the caller)
    const int hi = lo + rows_per_group < N ? (int)(lo + rows_per_group) : (int)N;
    double *arow = acc + lane;
    // A wavefront takes EIGHT CONSECUTIVE datapoints per batch: their lists are 128 contiguous uint16 / doubles, so the
    // whole batch's lists are FOUR coalesced loads (lane l: slot l & 15 of datapoint l >> 4, and of datapoint 4 + (l >> 4))
    // instead of sixteen -- with a load per datapoint and array the list traffic cost a third of the data stream
    // (0.34 ms with empty lists against 0.24 ms for the loads of Y alone).  (Packing them further -- index pairs in a
    // dword, value pairs in 16 bytes: two loads -- made the per-non-zero extraction scalar-heavy and the loop
    // issue-bound: 0.46 ms.)
    static_assert(SP_UNROLL == 8 && PM_BSC_NZ_MAX == 16, "64 lanes = the lists of four datapoints");
    struct Batch {
        double y[SP_UNROLL];
        double v[2];
        int ix[2];
        int n0;
    };
    // (batches that lie wholly inside the group -- all but possibly the group's last one, which is walked one list at a
    // time behind the loop: no predication and no second kind of load inside the pipelined loop)
    auto load = [&](int n0, Batch &b) {                        // n0: wavefront-uniform first datapoint of the batch
        const bool in = n0 + SP_UNROLL <= hi;
        const int nb = in ? n0 : lo;                           // (past the end: any valid batch, its lists are not used)
        b.n0 = in ? n0 : -1;
#pragma unroll
        for (int u = 0; u < SP_UNROLL; ++u) {
#ifdef PM_SP_NO_NT
            b.y[u] = (Y + (int64_t)(nb + u) * ldy)[col];
#else
            b.y[u] = __builtin_nontemporal_load((Y + (int64_t)(nb + u) * ldy) + col);   // read once: leave L2 to the lists
#endif
        }
        const uint16_t *pi = nz_idx + (int64_t)nb * PM_BSC_NZ_MAX;
        const double *pv = nz_val + (int64_t)nb * PM_BSC_NZ_MAX;
        b.ix[0] = pi[lane];
        b.ix[1] = pi[64 + lane];
        b.v[0] = pv[lane];
        b.v[1] = pv[64 + lane];
    };
    auto accumulate = [&](const Batch &b) {
        if (b.n0 < 0) return;                                   // uniform
        const uint64_t valid[2] = {__ballot(b.ix[0] != 0xFFFF), __ballot(b.ix[1] != 0xFFFF)};
#pragma unroll
        for (int u = 0; u < SP_UNROLL; ++u) {
            // the valid slots of a list are the leading ones
            const int cnt = __popc((uint32_t)(valid[u >> 2] >> (16 * (u & 3))) & 0xFFFFu);
            for (int t = 0; t < cnt; ++t) {
                const int src = 16 * (u & 3) + t;
                const int h = __builtin_amdgcn_readlane(b.ix[u >> 2], src);
                atomicAdd(arow + h * SP_DC, PM_Q(readlane_f64(b.v[u >> 2], src) * b.y[u], 0));
            }
        }
    };
    // three batches in flight: two are loading while one is accumulated
    constexpr int STEP = SP_WAVES * SP_UNROLL;
    if (hi - lo >= SP_UNROLL) {
        Batch b0, b1, b2;
        load(lo + SP_UNROLL * wave, b0);
        load(lo + SP_UNROLL * wave + STEP, b1);
        for (int n0 = lo + SP_UNROLL * wave; n0 + SP_UNROLL <= hi; n0 += 3 * STEP) {
            load(n0 + 2 * STEP, b2);
            accumulate(b0);
            load(n0 + 3 * STEP, b0);
            accumulate(b1);
            load(n0 + 4 * STEP, b1);
            accumulate(b2);
        }
    }
    // the group's ragged end (fewer than eight datapoints: the shard's last rows), one list at a time, by wavefront 0
    if (wave == 0) {
        for (int n = lo + (hi - lo) / SP_UNROLL * SP_UNROLL; n < hi; ++n) {
            const double yv = (Y + (int64_t)n * ldy)[col];
            const int ixs = lane < PM_BSC_NZ_MAX ? (int)nz_idx[(int64_t)n * PM_BSC_NZ_MAX + lane] : 0xFFFF;
            const double vs = lane < PM_BSC_NZ_MAX ? nz_val[(int64_t)n * PM_BSC_NZ_MAX + lane] : 0.0;
            const int cnt = __popc((uint32_t)__ballot(ixs != 0xFFFF) & 0xFFFFu);
            for (int t = 0; t < cnt; ++t)
                atomicAdd(arow + __builtin_amdgcn_readlane(ixs, t) * SP_DC, PM_Q(readlane_f64(vs, t) * yv, 0));
        }
    }
    __syncthreads();
    for (int i = tid; i < H * SP_DC; i += SP_WAVES * 64) {
        // (transposed output: consecutive threads take consecutive latents of one column, so the atomics of a wavefront
        // still land in whole lines; the strided LDS read behind it is 8 K elements per workgroup)
        const int h = transposed ? i % H : i / SP_DC, cc = transposed ? i / H : i % SP_DC, c = chunk * SP_DC + cc;
        const double a = acc[h * SP_DC + cc];
        if (a != 0.0 && c < D) pm_atomic_add(transposed ? Wp + (int64_t)c * ldw + h : Wp + (int64_t)h * ldw + c, a);
    }
}

}  // namespace

extern "C" int pm_bsc_wp_sparse_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy,
                                    double *stats, int64_t N, int64_t H, int64_t D, void *stream) {
    if (!stats) return PM_EINVAL;
    return pm_wp_sparse_f64(nz_idx, nz_val, Y, ldy, stats, D, stats + pm_bsc_stats_offset_scalars_dev(H, D) + 3, N, H, D,
                            stream);
}

static int wp_sparse_launch(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *Wp,
                            int64_t ldw, const double *gate, int64_t N, int64_t H, int64_t D, int transposed, void *stream,
                            double *expand_to = nullptr, int64_t lde = 0);

// pm_bsc_wp_sparse_f64 for lists whose pass kept only the overflowed datapoints' dense rows: if the gate is set, the launch
// completes `expect` from the lists instead of returning at once (pm_bsc_expand_lists_gated_f64 without its launch).
extern "C" int pm_bsc_wp_sparse_expand_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy,
                                           double *stats, double *expect, int64_t lde, int64_t N, int64_t H, int64_t D,
                                           void *stream) {
    if (!stats || !expect || lde < H) return PM_EINVAL;
    return wp_sparse_launch(nz_idx, nz_val, Y, ldy, stats, D, stats + pm_bsc_stats_offset_scalars_dev(H, D) + 3, N, H, D, 0,
                            stream, expect, lde);
}

extern "C" int pm_wp_sparse_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *Wp,
                                int64_t ldw, const double *gate, int64_t N, int64_t H, int64_t D, void *stream) {
    if (!gate || ldw < D) return PM_EINVAL;
    return wp_sparse_launch(nz_idx, nz_val, Y, ldy, Wp, ldw, gate, N, H, D, 0, stream);
}

// C (D x H, leading dimension ldc) += Y^T . V with V given by its rows' lists: the transposed output, no gate -- the
// sparse rows of GSC's moment contraction [Y | xpt_s | xpt_sz]^T . xpt_sz (gsc_et.py:592-625), whose dense rows
// (empty lists here) go through pm_gemm_tn_acc_rows_f64.
extern "C" int pm_wp_sparse_t_f64(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *C,
                                  int64_t ldc, int64_t N, int64_t H, int64_t D, void *stream) {
    if (ldc < H) return PM_EINVAL;
    return wp_sparse_launch(nz_idx, nz_val, Y, ldy, C, ldc, nullptr, N, H, D, 1, stream);
}

static int wp_sparse_launch(const uint16_t *nz_idx, const double *nz_val, const double *Y, int64_t ldy, double *Wp,
                            int64_t ldw, const double *gate, int64_t N, int64_t H, int64_t D, int transposed, void *stream,
                            double *expand_to, int64_t lde) {
    if (!nz_idx || !nz_val || !Y || !Wp || N < 0 || H <= 0 || D <= 0 || ldy < D) return PM_EINVAL;
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
                       rpg, transposed, expand_to, lde);
    return (int)hipGetLastError();
}

PM_DET_SETTER(bsc_wp_sparse)

// ---------------------------------------------------------------------------------------------
// The dense rows the list-writing E-step pass did not store (pm_bsc_estep_fused8_nz_f64 stores the dense E[s] row of a
// datapoint only when its list overflowed): rebuilt from the lists when -- and only when -- the dense product has to
// run (*gate != 0: some list of the shard overflowed).  Sixteen lanes per datapoint: the row is cleared with 16-byte
// stores, then lane t scatters entry t.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bsc_expand_lists_kernel(const uint16_t *__restrict__ nz_idx,
                                                               const double *__restrict__ nz_val,
                                                               double *__restrict__ expect, int64_t lde, int64_t N, int H,
                                                               const double *__restrict__ gate) {
    if (*gate == 0.0) return;
    expand_listed_rows(nz_idx, nz_val, expect, lde, N, H, (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4),
                       (int64_t)gridDim.x * (blockDim.x >> 4), threadIdx.x & 15);
}

extern "C" int pm_bsc_expand_lists_gated_f64(const uint16_t *nz_idx, const double *nz_val, double *expect, int64_t lde,
                                             int64_t N, int64_t H, const double *gate, void *stream) {
    if (!nz_idx || !nz_val || !expect || !gate || N < 0 || H <= 0 || lde < H) return PM_EINVAL;
    if (H > 65534) return PM_ERANGE;
    if (N == 0) return PM_OK;
    const int64_t blocks = (N + 15) / 16;
    hipLaunchKernelGGL(bsc_expand_lists_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), nz_idx, nz_val, expect, lde, N, (int)H, gate);
    return (int)hipGetLastError();
}
