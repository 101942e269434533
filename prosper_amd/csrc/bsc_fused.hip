// Binary Sparse Coding: scores GEMM + select_Hprimes + E_step in ONE kernel (bsc_et.py:98-192).
//
// The two-kernel path (gemm_f64.hip -> bsc_rows16.hip) writes the (N,H) scores to HBM and reads them back
// (820 MB per pass at config 2) and runs the row kernel strictly after the GEMM.  Here a workgroup owns 64
// datapoints x ALL H latents: wavefront w accumulates rows 16 w .. 16 w + 15 against every latent, and
// v_mfma_f64_16x16x4_f64 leaves that block in exactly the layout the row pass works in (bsc_rows16_body.h): the four
// 16-lane DPP rows of the wavefront each hold one datapoint, lane j of a row holds latents j, j + 16, ... .  So the
// epilogue -- top-H', Gram gather, state energies, log-joints, log-sum-exp -- runs out of the accumulators: no scores
// buffer, no second launch, no cross-wavefront exchange.
//
// K-loop: LDS-DMA ring as in gemm_nt_f64_dma_kernel (global_load_lds_dwordx4, XOR swizzle on the source address,
// counted s_waitcnt vmcnt, one raw s_barrier per K-step).  Stage = [64 datapoint rows | 16 NJ latent rows] x 8 doubles.
// Per K-step and wavefront: 1 + NJ/4 DMA instructions, 1 + NJ ds_read_b128, 2 NJ MFMAs; the latent fragments are read
// four column blocks at a time, one group ahead of the MFMAs that consume them.
// Two workgroups per CU: while one is in its epilogue (VALU, memory latency) the other keeps the matrix pipe busy.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"
#include "bsc_rows16_body.h"

namespace {

using namespace pm_rows16;

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int DK = 8;          // K columns per ring stage
constexpr int AROWS = 64;      // datapoints per workgroup: 4 wavefronts x 16 rows

__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef PM_FUSED_STAMPS   // diagnostic build (scratch/fused_stamps.sh): per-workgroup timeline, never in the shipped library
__device__ unsigned long long pm_fused_stamps[8192][8];
#define PM_STAMP(slot)                                                                      \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 8192)                                          \
            pm_fused_stamps[blockIdx.x][slot] = __builtin_amdgcn_s_memrealtime();           \
    } while (0)
#else
#define PM_STAMP(slot)
#endif

// FULL: H == 16 NJ, no latent-index guards in the row passes.  MSTATS: the row passes also produce the per-datapoint part
// of the M-step (E[s] rows into `expect`, statistics into the packed buffer `stats`): see row_estep_compute.
template <int NJ, int STAGES, bool FULL, bool MSTATS>
__global__ __launch_bounds__(256, 2) void bsc_estep_fused_kernel(
    const double *__restrict__ Y, int64_t ldy, const double *__restrict__ Wt, int64_t ldw, int D,
    const double *__restrict__ gram, const double *__restrict__ ynorm2, const double *__restrict__ wmu,
    const double *__restrict__ ymu, const uint16_t *__restrict__ masks, const uint16_t *__restrict__ parents,
    SizeOffsets so, int S, int gamma, pm_bsc_estep_params P, int64_t N, int H, int Hp, int mode,
    int32_t *__restrict__ cand, double *__restrict__ logpj, int64_t ldl, double *__restrict__ lse,
    double *__restrict__ expect, int64_t lde, double *__restrict__ stats, int Dstats) {
    constexpr int STAGE = (AROWS + 16 * NJ) * DK;   // doubles per stage
    constexpr int L = 1 + NJ / 4;                   // DMA instructions per K-step and wavefront
    constexpr int NG = NJ / 4;                      // groups of four column blocks
    static_assert(NG % 2 == 0, "fragment buffers alternate per group; a K-step must start on buffer 0");
    extern __shared__ __attribute__((aligned(1024))) double sm[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * AROWS;
    PM_STAMP(0);
#ifdef PM_FUSED_STAMPS
    if (tid == 0 && blockIdx.x < 8192)
        pm_fused_stamps[blockIdx.x][7] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) |
                                         __builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
#endif

    // per-thread table entries, fetched now (unconditional, clamped loads: no wait before the K-loop) and parked in
    // registers until the epilogue
    const int th = tid < H ? tid : H - 1;
    const double t_g = gram[(int64_t)th * H + th];
    double t_wmu = 0.0;
    if (wmu) t_wmu = wmu[th];
    uint32_t t_tab = 0;
    if (S > 0) {
        const int ts = tid < S ? tid : S - 1;
        t_tab = (uint32_t)masks[ts] | ((uint32_t)parents[ts] << 16);
    }

    // DMA sources: this wavefront moves its own 16 datapoint rows and latent blocks wave, wave + 4, ...
    // An address is (wave-uniform base, advanced per K-step on the scalar unit) + (per-lane 32-bit byte offset, fixed):
    // the saddr form of global_load_lds, issued through inline assembly (the builtin only takes a 64-bit vector address,
    // five 64-bit vector adds per K-step -- and every vector instruction in this loop delays the next MFMA).
    const int dr = lane >> 2, dj = (lane & 3) ^ ((lane >> 4) & 3);
    const char *sbase[5];
    uint32_t soff[5];
    static_assert(L <= 5, "sbase[] holds the datapoint rows + NJ/4 latent blocks");
    {
        int64_t r0 = m0 + 16 * wave;
        r0 = r0 < N ? r0 : N - 1;
        int64_t ra = r0 + dr;
        ra = ra < N ? ra : N - 1;
        sbase[0] = reinterpret_cast<const char *>(Y + r0 * ldy);
        soff[0] = (uint32_t)((ra - r0) * ldy * 8 + 16 * dj);
#pragma unroll
        for (int q = 0; q < NJ / 4; ++q) {
            int b0 = 16 * (wave + 4 * q);
            b0 = b0 < H ? b0 : H - 1;
            int rb = b0 + dr;
            rb = rb < H ? rb : H - 1;
            sbase[1 + q] = reinterpret_cast<const char *>(Wt + (int64_t)b0 * ldw);
            soff[1 + q] = (uint32_t)((int64_t)(rb - b0) * ldw * 8 + 16 * dj);
        }
    }
    // LDS byte address of this wavefront's first chunk in stage 0 (chunk = 16 rows x 8 doubles = 1 KB)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) double *)(sm)) + (unsigned)wave * 1024u;
    auto dma1 = [&](unsigned dst, uint32_t voff, const char *base) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(base)
                     : "memory");
    };
    auto dma = [&](int kt, int stage) {
        const unsigned dst = lds0 + (unsigned)stage * (unsigned)(STAGE * 8);
        const int64_t k0 = (int64_t)kt * (DK * 8);
        dma1(dst, soff[0], sbase[0] + k0);
#pragma unroll
        for (int q = 0; q < NJ / 4; ++q) dma1(dst + (4 + 4 * q) * 1024u, soff[1 + q], sbase[1 + q] + k0);
    };

    // fragment reads (see gemm_nt_f64_dma_kernel): k-group fk of the first MFMA of a K-step takes column 2 fk, of the
    // second column 2 fk + 1 -- the 16-byte slot the DMA wrote; pair p of row R sits at R*8 + ((p ^ ((R>>2)&3)) << 1)
    const int frow = lane & 15, fk = lane >> 4;
    const int sw = (frow >> 2) & 3;
    const int a_off = (wave * 16 + frow) * DK + ((fk ^ sw) << 1);
    const int b_off = AROWS * DK + frow * DK + ((fk ^ sw) << 1);
    auto read_a = [&](int stage) { return *reinterpret_cast<const d2 *>(sm + stage * STAGE + a_off); };
    auto read_b = [&](int stage, int g, d2 (&f)[4]) {
        const double *sb = sm + stage * STAGE + b_off + g * 4 * 16 * DK;
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const d2 *>(sb + q * 16 * DK);
    };

    d4 acc[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = D / DK;   // host guarantees D % DK == 0, D >= DK
#pragma unroll
    for (int t = 0; t < STAGES; ++t)
        if (t < nk) dma(t, t);
    // this wavefront's part of K-step 0 has landed
    {
        const int behind = (nk < STAGES ? nk : STAGES) - 1;   // K-steps issued beyond step 0
        if (behind >= 3) wait_vmcnt<3 * L>();
        else if (behind == 2) wait_vmcnt<2 * L>();
        else if (behind == 1) wait_vmcnt<L>();
        else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();

    d2 fa[2];
    d2 fb[2][4];
    fa[0] = read_a(0);
    read_b(0, 0, fb[0]);

    // One K-step.  `stage`, `nstage` (ring slots of K-steps t and t+1) and `par` (which of fa[] holds step t's datapoint
    // fragment) are compile-time constants in the unrolled main loop -- LDS addresses become immediates, no register
    // copies -- and run-time values in the generic loop.
    auto kstep = [&](int t, int stage, int nstage, int par) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
                read_b(stage, g + 1, fb[(g + 1) & 1]);
            } else if (t + 1 < nk) {
                // my reads of this stage are done (lgkmcnt) and my share of K-step t+1 has landed (vmcnt); after the
                // barrier that holds for every wavefront: stage t may be refilled, t+1 may be read
                int ahead = nk - t - 2;
                ahead = ahead < STAGES - 2 ? ahead : STAGES - 2;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ahead >= 2) wait_vmcnt<2 * L>();
                else if (ahead == 1) wait_vmcnt<L>();
                else wait_vmcnt<0>();
#ifndef PM_FUSED_NO_BARRIER          // timing-only ablation (wrong results): what the hand-over barrier costs
                __builtin_amdgcn_s_barrier();
#endif
                const d2 an = read_a(nstage);
                if (par) fa[0] = an;
                else fa[1] = an;
                read_b(nstage, 0, fb[0]);
                if (t + STAGES < nk) dma(t + STAGES, stage);
            }
            const d2 af = par ? fa[1] : fa[0];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 * g + q] = mfma16(af.x, fb[g & 1][q].x, acc[4 * g + q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 * g + q] = mfma16(af.y, fb[g & 1][q].y, acc[4 * g + q]);
        }
    };
    int t = 0;
    if (STAGES % 2 == 0) {
        for (; t + STAGES <= nk; t += STAGES) {
#pragma unroll
            for (int u = 0; u < STAGES; ++u) kstep(t + u, u, (u + 1) % STAGES, u & 1);
        }
    }
    for (int stage = t % STAGES; t < nk; ++t) {       // (STAGES even: t is a multiple of STAGES here, its parity is 0)
        const int nstage = (stage + 1 == STAGES) ? 0 : stage + 1;
        kstep(t, stage, nstage, t & 1);
        stage = nstage;
    }

    PM_STAMP(1);
    // ---------------- epilogue: the row passes, straight from the accumulators --------------------------------
    // every wavefront is done with the ring: its LDS becomes the row passes' tables and datapoint areas
    __builtin_amdgcn_s_barrier();
    unsigned char *smem = reinterpret_cast<unsigned char *>(sm);
    const Layout lay = make_layout(H, Hp, S, 1);
    build_tables(smem, lay, tid, t_g, t_wmu, t_tab, P.ecoef, P.prior_scale * P.pil_bar, gram, wmu, H, masks, parents, S,
                 Hp);
    __syncthreads();

    PM_STAMP(2);
    const RowParams A{gram, ynorm2, wmu, ymu, S, gamma, P, N, H, Hp, mode, cand, logpj, ldl, lse, expect, lde,
                      MSTATS ? stats + pm_bsc_stats_offset_wq_dev(H, Dstats) : nullptr};
    const RowLds Lds = row_lds(smem, lay, wave * 4 + fk);
    MAcc macc{0.0, 0.0, 0.0};

    // Row fk (16 lanes) of this wavefront holds datapoint m0 + 16 wave + fk + 4 r in element r of every accumulator.
    const int64_t nbase = m0 + 16 * wave + fk;
    if (Hp <= 8) {
        // Up to 8 candidates (a Gram block of <= 64 entries, 4 per lane): three phases over the four passes, so that the
        // global loads of ALL passes (|y|^2, Gram entries of the candidates) are in flight together instead of one
        // dependent round trip per pass.
        int mycs[4];
        double acs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // 1. selection; the score of the candidate each lane holds, through the LDS row
            double a[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) a[i] = acc[i][r];
#pragma unroll
            for (int i = 0; i < NJ; ++i)
                if (16 * i < lay.HT) Lds.row[frow + 16 * i] = a[i];
            mycs[r] = row_select<NJ, 0, FULL>(a, A, Lds, lane, nbase + 4 * r);
            acs[r] = Lds.row[frow < Hp ? mycs[r] : 0];
            wave_lds_sync16();
        }
        PM_STAMP(3);
        if (!(mode & 2)) return;
        RowFetch<4> F[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) F[r] = row_fetch<4>(mycs[r], A, Lds, lane, nbase + 4 * r);      // 2. loads
        PM_STAMP(5);
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // 3. energies, log-joints, log-sum-exp
            double a[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) a[i] = acc[i][r];
            row_estep_fetched<NJ, FULL, 4, MSTATS>(a, acs[r], mycs[r], F[r], A, so, Lds, lane, nbase + 4 * r, &macc);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t n = nbase + 4 * r;
            double a[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) a[i] = acc[i][r];
            const int c = row_select<NJ, 0, FULL>(a, A, Lds, lane, n);
            if (mode & 2) {
                // the scores as an indexable row: the candidates' scores are looked up by latent index
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    if (16 * i < lay.HT) Lds.row[frow + 16 * i] = a[i];
                wave_lds_sync16();
                row_estep<NJ, FULL>(a, Lds.row, c, A, so, Lds, lane, n);
            }
        }
    }
    if (MSTATS) {
        // column sums of E[s] and the scalar statistics of this tile -> packed statistics buffer
        double sig = pm_wave_sum(macc.sig), fs = pm_wave_sum(macc.fs), cnt = pm_wave_sum(macc.cnt);
        __syncthreads();                     // every wavefront's LDS atomics into mus are done
        double *red = reinterpret_cast<double *>(smem + lay.off_dp);     // (datapoint areas are free now)
        if (lane == 0) {
            red[wave * 3 + 0] = sig;
            red[wave * 3 + 1] = fs;
            red[wave * 3 + 2] = cnt;
        }
        __syncthreads();
        double *sc = stats + pm_bsc_stats_offset_scalars_dev(H, Dstats);
        if (tid < 3) {
            double v = 0.0;
            for (int w = 0; w < 4; ++w) v += red[w * 3 + tid];
            if (v != 0.0) pm_atomic_add(sc + tid, PM_Q(v, tid == 0 ? 1 : tid == 1 ? 2 : 0));
        }
        double *g_mus = stats + pm_bsc_stats_offset_mus_dev(H, Dstats);
        if (tid < H) {
            const double v = Lds.mus[tid];
            if (v != 0.0) pm_atomic_add(g_mus + tid, v);
        }
    }
    PM_STAMP(4);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Four ring stages (80 KB at H = 256): the K-loop is unrolled over the ring, every LDS address an immediate.  (A
// three-stage ring and a start stagger of the second resident workgroup set were measured in round 2 -- -1 % and no gain
// -- and are gone, and with them the library's only environment look-ups: no mutable state outside the arguments.)

template <int NJ, int ST>
size_t fused_lds_bytes(int64_t H, int64_t Hp, int64_t S) {
    const size_t ring = sizeof(double) * ST * (AROWS + 16 * NJ) * DK;
    const size_t epi = (size_t)make_layout((int)H, (int)Hp, (int)S, 1).bytes;
    return ring > epi ? ring : epi;
}

}  // namespace

// The fused kernel covers H <= 256 latents (16 accumulator tiles per wavefront), D a multiple of 8, 16-byte aligned
// rows, and whatever pm_bsc_rows16_supported admits for the row passes.
extern "C" int pm_bsc_fused_supported(int64_t H, int64_t D, int64_t Hprime, int64_t S) {
    if (H <= 0 || H > 256 || D < DK || D % DK != 0) return 0;
    if (!pm_bsc_rows16_supported(H, Hprime, S)) return 0;
    return make_layout((int)H, (int)Hprime, (int)S, 1).bytes <= 80 * 1024 ? 1 : 0;   // two workgroups per CU
}

#ifdef PM_FUSED_STAMPS
extern "C" int pm_fused_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pm_fused_stamps), sizeof(unsigned long long) * 8 * n);
}
#endif

// Workgroups of the fused kernel one CU holds at once for this shape (LDS- and register-limited; 2 is the design
// point: one in its K-loop on the matrix pipe while the other runs its row passes).  <= 0: not supported / error.
extern "C" int pm_bsc_fused_occupancy(int64_t H, int64_t D, int64_t Hprime, int64_t S) {
    if (!pm_bsc_fused_supported(H, D, Hprime, S)) return 0;
    int n = 0;
    hipError_t e;
#define PM_OCC(NJ, ST)                                                                                                 \
    do {                                                                                                               \
        const size_t shmem = fused_lds_bytes<NJ, ST>(H, Hprime, S);                                                       \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(bsc_estep_fused_kernel<NJ, ST, false, false>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                               \
        if (e == hipSuccess)                                                                                           \
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bsc_estep_fused_kernel<NJ, ST, false, false>, 256,    \
                                                             shmem);                                                   \
    } while (0)
    if (H <= 128) PM_OCC(8, 4);
    else PM_OCC(16, 4);
#undef PM_OCC
    return e == hipSuccess ? n : -(int)e;
}

extern "C" int pm_bsc_estep_fused_f64(const double *Y, int64_t ldy, const double *Wt, int64_t ldw, const double *gram,
                                      const double *ynorm2, const double *wmu, const double *ymu,
                                      const uint16_t *state_masks, const uint16_t *state_parents,
                                      const int32_t *size_offsets_host, int64_t S, int64_t gamma,
                                      const pm_bsc_estep_params *params_host, int64_t N, int64_t D, int64_t H,
                                      int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl, double *lse,
                                      double *expect, int64_t lde, double *stats, int64_t D_stats, void *stream) {
    if (!Y || !Wt || !gram || !ynorm2 || !cand || N < 0 || H <= 0 || D <= 0 || Hprime <= 0 || S < 0 || ldy < D ||
        ldw < D || !(mode & 3) || ((wmu == nullptr) != (ymu == nullptr)))
        return PM_EINVAL;
    if ((mode & 2) && (!params_host || !logpj || ldl < 1 + H + S || gamma < 1 || gamma > Hprime ||
                       (S > 0 && (!state_masks || !state_parents || !size_offsets_host))))
        return PM_EINVAL;
    if (!pm_bsc_fused_supported(H, D, Hprime, S) || (mode & ~3)) return PM_ERANGE;   // (BSC's own ranking only)
    if (stats && (!expect || lde < H || !lse || !(mode & 2) || D_stats <= 0)) return PM_EINVAL;
    if (stats && Hprime > 8) return PM_ERANGE;          // (the M-statistics ride on the three-phase row passes)
    if (!aligned16(Y) || !aligned16(Wt) || (ldy % 2) || (ldw % 2)) return PM_EINVAL;
    if (N == 0) return PM_OK;
    SizeOffsets so;
    for (int g = 0; g < PM_MAX_HPRIME; ++g) so.off[g] = (int)S;
    if ((mode & 2) && S > 0)
        for (int g = 0; g < gamma; ++g) so.off[g] = size_offsets_host[g];  // off[g-2] = first state of size g
    pm_bsc_estep_params P = params_host ? *params_host : pm_bsc_estep_params{0, 0, 0, 0};
    const int64_t tiles = (N + AROWS - 1) / AROWS;
    if (tiles > INT32_MAX) return PM_ERANGE;
    dim3 grid((unsigned)tiles), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PM_LAUNCH_FM(NJ, ST, F, M)                                                                                     \
    do {                                                                                                               \
        const size_t shmem = fused_lds_bytes<NJ, ST>(H, Hprime, S);                                                    \
        if (int e = (int)hipFuncSetAttribute(reinterpret_cast<const void *>(bsc_estep_fused_kernel<NJ, ST, F, M>),     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))                  \
            return e;                                                                                                  \
        hipLaunchKernelGGL((bsc_estep_fused_kernel<NJ, ST, F, M>), grid, block, shmem, s, Y, ldy, Wt, ldw, (int)D,     \
                           gram, ynorm2, wmu, ymu, state_masks, state_parents, so, (int)S, (int)gamma, P, N, (int)H,  \
                           (int)Hprime, mode, cand, logpj, ldl, lse, expect, lde, stats, (int)D_stats);               \
    } while (0)
#define PM_LAUNCH_F(NJ, ST, F)              \
    do {                                    \
        if (stats) {                        \
            PM_LAUNCH_FM(NJ, ST, F, true);  \
        } else {                            \
            PM_LAUNCH_FM(NJ, ST, F, false); \
        }                                   \
    } while (0)
#define PM_LAUNCH(NJ, ST)                  \
    do {                                   \
        if (H == 16 * NJ) {                \
            PM_LAUNCH_F(NJ, ST, true);     \
        } else {                           \
            PM_LAUNCH_F(NJ, ST, false);    \
        }                                  \
    } while (0)
    if (H <= 128) PM_LAUNCH(8, 4);
    else PM_LAUNCH(16, 4);
#undef PM_LAUNCH
#undef PM_LAUNCH_F
#undef PM_LAUNCH_FM
    return (int)hipGetLastError();
}

PM_DET_SETTER(bsc_fused)
