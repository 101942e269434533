// Binary Sparse Coding row kernels, fast path: 16 lanes per datapoint.
//
// A 64-lane wavefront is four DPP rows of 16 lanes; each row owns one datapoint, so a wavefront
// works on four datapoints at once and every reduction over a datapoint's latents / states (top-H',
// max, sum) is a 4-step DPP butterfly inside the row -- no LDS crossbar (ds_bpermute), no
// cross-row traffic.  Lane j of a row holds latents h = j + 16 i (i < VPL), i.e. each row reads and
// writes whole 128-byte segments of its datapoint's rows of `scores` / `logpj` / `expect`.
//
//   bsc_select_estep16_kernel   select_Hprimes (bsc_et.py:98-115) and/or E_step (bsc_et.py:119-192)
//                               in ONE pass over the scores: the scores stay in registers between
//                               the two, candidates never round-trip through HBM
//   bsc_mstep_rows16_kernel     per-datapoint part of M_step (bsc_et.py:271-272,334-366,395-415)
//
// The per-datapoint pass of the first kernel lives in bsc_rows16_body.h (shared with the fused
// scores-GEMM + E-step kernel of bsc_fused.hip).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"
#include "bsc_rows16_body.h"

namespace {

using namespace pm_rows16;

// ---------------------------------------------------------------------------------------------
// select_Hprimes + E_step: pm_rows16::row_select / row_estep per group of 16 datapoints (mode bits: see RowParams)
// ---------------------------------------------------------------------------------------------
// ESTEP = false: the selection-only instantiation (mode bit 1 clear: what MCA / MMCA / DSC / TSC and a stand-alone
// select_Hprimes call) -- a third of the registers, twice the wavefronts per SIMD.
template <int VPL, bool ESTEP>
__global__ __launch_bounds__(256, ESTEP ? (VPL <= 4 ? 4 : VPL <= 8 ? 3 : VPL <= 16 ? 2 : 1)
                                        : (VPL <= 8 ? 4 : VPL <= 16 ? 3 : 1)) void bsc_select_estep16_kernel(
    const double *__restrict__ scores, int64_t lds, const double *__restrict__ gram,
    const double *__restrict__ ynorm2, const double *__restrict__ wmu, const double *__restrict__ ymu,
    const uint16_t *__restrict__ masks, const uint16_t *__restrict__ parents, SizeOffsets so, int S, int gamma,
    pm_bsc_estep_params P, int64_t N, int H, int Hp, int mode, int32_t *__restrict__ cand,
    double *__restrict__ logpj, int64_t ldl, double *__restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Layout lay = make_layout(H, Hp, S, 0);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    const int j = lane & 15, row = lane >> 4;
    {
        const int th = tid < H ? tid : H - 1, ts = tid < S ? tid : S - 1;
        const uint32_t tab_s = S > 0 ? ((uint32_t)masks[ts] | ((uint32_t)parents[ts] << 16)) : 0u;
        build_tables(smem, lay, tid, gram[(int64_t)th * H + th], wmu ? wmu[th] : 0.0, tab_s, P.ecoef,
                     P.prior_scale * P.pil_bar, gram, wmu, H, masks, parents, S, Hp);
    }
    __syncthreads();

    const RowParams A{gram, ynorm2, wmu, ymu, S, gamma, P, N, H, Hp, mode, cand, logpj, ldl, lse};
    const RowLds L = row_lds(smem, lay, wave * 4 + row);

    const int64_t groups = (N + ROWS - 1) / ROWS;
    for (int64_t g0 = blockIdx.x; g0 < groups; g0 += gridDim.x) {
        // last rows first: the scores GEMM has just written the shard front to back, so its tail is what the
        // memory-side cache still holds
        const int64_t grp = groups - 1 - g0;
        const int64_t n = grp * ROWS + wave * 4 + row;
        const int64_t nn = n < N ? n : N - 1;   // dead rows shadow the last datapoint, write nothing
        const double *arow = scores + nn * lds;
        double a[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            a[i] = (h < H) ? arow[h] : 0.0;
        }
        const int myc = row_select<VPL>(a, A, L, lane, n);
        if (ESTEP) row_estep<VPL>(a, arow, myc, A, so, L, lane, n);
    }
}

// ---------------------------------------------------------------------------------------------
// select_Hprimes of Discrete / Ternary Sparse Coding (dsc_et.py:347-410, tsc_et.py:142-213) in ONE pass over the scores:
// the ranking values -- DSC: R[h] = -max_k (pre1 (v_k^2 G_hh - 2 v_k a_h) + log pi_k), the H' smallest, best first; TSC:
// R[h] = -(G_hh + 2 a_h), R[H + h] = -(G_hh - 2 a_h) over the 2 H one-cause states, the H' largest, best last, stored as
// latents (state % H) -- are formed in registers from the scores row and ranked by row_select.  Rounds 1-3 wrote them to an
// (N, H) / (N, 2H) buffer with one kernel and ranked that with a second (+ a modulo launch for TSC): 0.08 / 0.135 ms of a
// 0.70 / 0.65 ms EM iteration.  TSC needs H == 16 VPL (the second half of the keys continues the lane layout).
// ---------------------------------------------------------------------------------------------
template <int VPL, bool TSC>
__global__ __launch_bounds__(256, (TSC ? 2 * VPL : VPL) <= 8 ? 4 : (TSC ? 2 * VPL : VPL) <= 16 ? 3 : 1) void xsc_select16_kernel(
    const double *__restrict__ scores, int64_t lds, const double *__restrict__ gram, pm_dsc_params P, int64_t N, int H,
    int Hp, int32_t *__restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VK = TSC ? 2 * VPL : VPL;              // keys per lane
    const int HK = TSC ? 2 * H : H;
    const Layout lay = make_layout(HK, Hp, 0, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, row = lane >> 4;
    double *s_w2 = reinterpret_cast<double *>(smem);     // (the w2 table of the layout: |W_h|^2; the other tables stay unused)
    __shared__ double s_v[PM_DSC_MAX_K], s_lp[PM_DSC_MAX_K];
    if (tid < PM_DSC_MAX_K) {
        s_v[tid] = P.values[tid];
        s_lp[tid] = P.logpi[tid];
    }
    for (int h = tid; h < H; h += 256) s_w2[h] = gram[(int64_t)h * H + h];
    __syncthreads();
    RowParams A{};
    A.N = N;
    A.H = HK;
    A.Hp = Hp;
    A.mode = 1;
    A.cand = cand;
    A.cand_mod = TSC ? H : 0;
    const RowLds L = row_lds(smem, lay, wave * 4 + row);
    const int64_t groups = (N + ROWS - 1) / ROWS;
    for (int64_t g0 = blockIdx.x; g0 < groups; g0 += gridDim.x) {
        const int64_t grp = groups - 1 - g0;             // (last rows first: see bsc_select_estep16_kernel)
        const int64_t n = grp * ROWS + wave * 4 + row;
        const int64_t nn = n < N ? n : N - 1;
        const double *arow = scores + nn * lds;
        double r[VK], a[VPL], w2[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            const int hc = h < H ? h : H - 1;
            a[i] = arow[hc];
            w2[i] = s_w2[hc];
        }
        if (TSC) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const double a2 = 2.0 * a[i];
                r[i] = -(w2[i] + a2);
                r[VPL + i] = -(w2[i] - a2);
            }
        } else {
#pragma unroll
            for (int i = 0; i < VPL; ++i) r[i] = -INFINITY;
            for (int k = 0; k < P.K; ++k) {              // (uniform; the values sit in LDS: indexing the kernel-argument
                if (k == P.K0) continue;                 // struct with a loop variable is a dependent scalar load per use)
                const double v = s_v[k], lp = s_lp[k];
#pragma unroll
                for (int i = 0; i < VPL; ++i) r[i] = fmax(r[i], P.pre1 * (v * v * w2[i] - 2.0 * v * a[i]) + lp);
            }
#pragma unroll
            for (int i = 0; i < VPL; ++i) r[i] = -r[i];
        }
        // DSC: smallest first, values as they are (ranking flags 1 | 2); TSC: largest, as they are (2)
        if (TSC) (void)row_select<VK, 2, true>(r, A, L, lane, n);
        else (void)row_select<VK, 3, false>(r, A, L, lane, n);
    }
}

// ---------------------------------------------------------------------------------------------
// M_step, per-datapoint part
// ---------------------------------------------------------------------------------------------
template <int VPL, bool NZ>      // NZ: list mode (nz_idx / nz_val given): E[s] rows merged through LDS, non-zeros listed
__global__ __launch_bounds__(256, VPL <= 16 ? (NZ && VPL == 16 ? 3 : 4) : 1) void bsc_mstep_rows16_kernel(
    const double *__restrict__ logpj, int64_t ldl, const double *__restrict__ lse, double lse_cut,
    const int32_t *__restrict__ cand, const uint16_t *__restrict__ masks, int S, pm_bsc_estep_params P, int64_t N,
    int H, int D, int Hp, double *__restrict__ expect, int64_t lde, double *__restrict__ stats,
    uint16_t *__restrict__ nz_idx, double *__restrict__ nz_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ qdiag (H) | mus (H) | per datapoint: m2 (Hp*Hp) | red (3*4) | masks (S) ]
    double *s_qdiag = reinterpret_cast<double *>(smem);
    double *s_mus = s_qdiag + H;
    double *s_m2all = s_mus + H;
    double *s_red = s_m2all + ROWS * Hp * Hp;
    uint16_t *s_masks = reinterpret_cast<uint16_t *>(s_red + 12);
    double *s_rows = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(s_masks) + (((size_t)S * 2 + 7) & ~size_t(7)));   // list mode only

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    const int j = lane & 15, row = lane >> 4;
    for (int h = tid; h < H; h += 256) {
        s_qdiag[h] = 0.0;
        s_mus[h] = 0.0;
    }
    for (int s = tid; s < S; s += 256) s_masks[s] = masks[s];
    double *s_m2 = s_m2all + (wave * 4 + row) * Hp * Hp;
    for (int p = j; p < Hp * Hp; p += 16) s_m2[p] = 0.0;
    __syncthreads();

    double *Wq = stats + pm_bsc_stats_offset_wq_dev(H, D);
    const double ppil = P.prior_scale * P.pil_bar;
    const double inv_ecoef = 1.0 / P.ecoef;
    const double qcut = -50.0;  // exp(-50) ~ 2e-22: below every statistic's rounding

    double qd[VPL];  // column sums of q1 over this lane's datapoints (mus = these + the candidate terms)
#pragma unroll
    for (int i = 0; i < VPL; ++i) qd[i] = 0.0;
    double sig = 0.0, fs = 0.0, cnt = 0.0, overflow = 0.0;

    const int64_t groups = (N + ROWS - 1) / ROWS;
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int64_t n = grp * ROWS + wave * 4 + row;
        const bool live = n < N;
        const int64_t nn = live ? n : N - 1;
        const double l = lse[nn];
        const bool keep = live && (l >= lse_cut);  // uniform per DPP row
        double *erow = expect + nn * lde;
        const double *f = logpj + nn * ldl;

        int myc = (j < Hp) ? cand[nn * Hp + j] : 0;

        double es[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            double q = 0.0;
            if (keep && h < H) {
                const double fh = f[1 + h];
                const double dlt = fh - l;
                if (dlt > qcut) {
                    q = exp(dlt);
                    sig += q * ((fh - ppil) * inv_ecoef);
                }
            }
            es[i] = q;
            qd[i] += q;
        }
        if (keep && j == 0) {
            const double f0 = f[0];
            const double dlt = f0 - l;
            if (dlt > qcut) sig += exp(dlt) * (f0 * inv_ecoef);
            fs += l;
            cnt += 1.0;
        }
        // multi-cause states: only the few with a non-negligible posterior touch the moments
        if (keep) {
            for (int s = j; s < S; s += 16) {
                const double fv = f[1 + H + s];
                const double dlt = fv - l;
                if (dlt > qcut) {
                    const unsigned mask = s_masks[s];
                    const double q = exp(dlt);
                    sig += q * ((fv - ppil * (double)__builtin_popcount(mask)) * inv_ecoef);
                    unsigned mi = mask;
                    while (mi) {
                        const int i = __builtin_ctz(mi);
                        mi &= mi - 1;
                        atomicAdd(&s_m2[i * Hp + i], q);
                        unsigned mk = mi;
                        while (mk) {
                            const int k = __builtin_ctz(mk);
                            mk &= mk - 1;
                            atomicAdd(&s_m2[i * Hp + k], q);  // i < k: upper triangle
                        }
                    }
                }
            }
        }
        wave_lds_sync16();
        // scatter E[s_i s_k] of the candidate block into Wq (upper triangle), add E[s_c] to the row
        for (int p0 = 0; p0 < Hp * Hp; p0 += 16) {            // uniform trip count: every lane feeds the bpermutes
            const int p = p0 + j;
            const bool valid = p < Hp * Hp;
            const int i = valid ? p / Hp : 0, k = valid ? p - i * Hp : 0;
            const double m2 = valid ? s_m2[p] : 0.0;
            // candidates i and k of this datapoint, from the lanes that hold them
            const int ci = __builtin_amdgcn_ds_bpermute(((lane & 48) + i) << 2, myc);
            const int ck = __builtin_amdgcn_ds_bpermute(((lane & 48) + k) << 2, myc);
            if (k >= i && m2 != 0.0) {
                const int lo = ci < ck ? ci : ck, hi = ci < ck ? ck : ci;
                pm_atomic_add(Wq + (int64_t)lo * H + hi, PM_Q(m2, 0));
            }
        }
        // E[s_c] = q1_c + sum_{s containing c} q_s: the owner lane of latent c adds the diagonal term
        if (j < Hp) {
            const double m1 = s_m2[j * Hp + j];
            if (m1 != 0.0) atomicAdd(&s_mus[myc], PM_Q(m1, 0));
        }
        // the row of singleton weights goes out first; the candidates' multi-cause terms are added to it by the
        // lanes that own them once those stores have completed (8 f64 atomics per datapoint instead of a
        // 16 x VPL select chain into the register row)
        const double m1c = (j < Hp) ? s_m2[j * Hp + j] : 0.0;
        if (NZ) {
            // list mode: the candidates' multi-cause terms are merged into the row through LDS (one row of H doubles per
            // datapoint slot, behind the masks), so that the row is stored complete and its non-zeros can be listed
            double *srow = s_rows + (size_t)(wave * 4 + row) * H;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (j + 16 * i < H) srow[j + 16 * i] = es[i];
            wave_lds_sync16();
            if (keep && j < Hp && m1c != 0.0) srow[myc] += m1c;       // (candidates are distinct latents)
            wave_lds_sync16();
#pragma unroll
            for (int i = 0; i < VPL; ++i) es[i] = (j + 16 * i < H) ? srow[j + 16 * i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            if (live && h < H) erow[h] = es[i];
        }
        if (NZ) {
            // the row's non-zeros as a list for pm_bsc_wp_sparse_f64 (format of pm_bsc_estep_fused8_nz_f64)
            uint16_t *nzi = nz_idx + nn * PM_BSC_NZ_MAX;
            double *nzv = nz_val + nn * PM_BSC_NZ_MAX;
            const int rb = lane & 48;
            uint32_t nzn = 0;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const double v = es[i];
                const int h = j + 16 * i;
                const uint32_t mine = (uint32_t)((__ballot(v != 0.0) >> rb) & 0xFFFFull);
                const uint32_t pos = nzn + __builtin_popcount(mine & ((1u << j) - 1u));
                if (v != 0.0 && pos < PM_BSC_NZ_MAX && live) {
                    nzi[pos] = (uint16_t)h;
                    nzv[pos] = v;
                }
                nzn += __builtin_popcount(mine);
            }
            if (live) {
                if ((uint32_t)j >= nzn) nzi[j] = 0xFFFFu;               // (16 lanes = PM_BSC_NZ_MAX slots)
                if (j == 0 && nzn > PM_BSC_NZ_MAX) overflow += 1.0;
            }
        }
        wave_lds_sync16();
        for (int p = j; p < Hp * Hp; p += 16) s_m2[p] = 0.0;
        if (!NZ) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (keep && j < Hp && m1c != 0.0) pm_atomic_add(erow + myc, m1c);
        }
        wave_lds_sync16();
    }

    // column sums: rows of a wavefront, then wavefronts of the workgroup, then one atomic per latent
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int h = j + 16 * i;
        if (h < H) {
            atomicAdd(&s_qdiag[h], PM_Q(qd[i], 0));
            atomicAdd(&s_mus[h], PM_Q(qd[i], 0));
        }
    }
    sig = pm_wave_sum(sig);
    fs = pm_wave_sum(fs);
    cnt = pm_wave_sum(cnt);
    overflow = pm_wave_sum(overflow);
    if (lane == 0 && overflow != 0.0) pm_atomic_add(stats + pm_bsc_stats_offset_scalars_dev(H, D) + 3, overflow);
    if (lane == 0) {
        s_red[wave * 3 + 0] = sig;
        s_red[wave * 3 + 1] = fs;
        s_red[wave * 3 + 2] = cnt;
    }
    __syncthreads();
    double *sc = stats + pm_bsc_stats_offset_scalars_dev(H, D);
    if (tid < 3) {
        double v = 0.0;
        for (int w = 0; w < 4; ++w) v += s_red[w * 3 + tid];
        if (v != 0.0) pm_atomic_add(sc + tid, PM_Q(v, tid == 0 ? 1 : tid == 1 ? 2 : 0));
    }
    double *g_qdiag = stats + pm_bsc_stats_offset_qdiag_dev(H, D);
    double *g_mus = stats + pm_bsc_stats_offset_mus_dev(H, D);
    for (int h = tid; h < H; h += 256) {
        const double a = s_qdiag[h], b = s_mus[h];
        if (a != 0.0) pm_atomic_add(g_qdiag + h, a);
        if (b != 0.0) pm_atomic_add(g_mus + h, b);
    }
}

constexpr size_t PM_ROWS16_LDS_MAX = 156 * 1024;
inline int64_t grid_groups(int64_t N) {
    const int64_t groups = (N + ROWS - 1) / ROWS;
    const int64_t cap = 256 * 8;      // persistent grid: two rounds of resident workgroups walk the groups
    return groups < cap ? (groups < 1 ? 1 : groups) : cap;
}

static int allow_lds16(const void *kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

// The fast path covers H <= 512 (VPL <= 32 scores per lane) and states that fit the LDS areas.
extern "C" int pm_bsc_rows16_supported(int64_t H, int64_t Hprime, int64_t S) {
    if (H <= 0 || H > 512 || Hprime <= 0 || Hprime > PM_MAX_HPRIME || Hprime > H || S < 0 || S > 4096) return 0;
    // (up to the CU's whole LDS less a margin for the kernels' static arrays: beyond 80 KB one workgroup per CU -- four wavefronts
    // -- which still beats the generic two-launch path by a third at H = 256, H' = 10, gamma = 4: DESIGN.md, shape sweep)
    return make_layout((int)H, (int)Hprime, (int)S, 0).bytes <= PM_ROWS16_LDS_MAX ? 1 : 0;
}

// The list-writing M-step pass needs one more score row per datapoint slot (ROWS * H doubles) than the plain one.
static size_t mstep_rows16_lds(int64_t H, int64_t Hprime, int64_t S, bool lists) {
    size_t shmem = sizeof(double) * (2 * H + ROWS * Hprime * Hprime + 12) + sizeof(uint16_t) * S;
    if (lists) shmem = ((shmem + 7) & ~size_t(7)) + sizeof(double) * ROWS * (size_t)H;
    return shmem;
}

extern "C" int pm_bsc_rows16_nz_supported(int64_t H, int64_t Hprime, int64_t S) {
    if (!pm_bsc_rows16_supported(H, Hprime, S) || H > 256) return 0;       // (list indices are uint16 slots of <= 256 latents)
    return mstep_rows16_lds(H, Hprime, S, true) <= PM_ROWS16_LDS_MAX ? 1 : 0;
}

extern "C" int pm_bsc_select_estep_f64(const double *scores, int64_t lds, const double *gram, const double *ynorm2,
                                       const double *wmu, const double *ymu, const uint16_t *state_masks,
                                       const uint16_t *state_parents, const int32_t *size_offsets_host, int64_t S,
                                       int64_t gamma, const pm_bsc_estep_params *params_host, int64_t N, int64_t H,
                                       int64_t Hprime, int mode, int32_t *cand, double *logpj, int64_t ldl,
                                       double *lse, void *stream) {
    if (!scores || !gram || !ynorm2 || !cand || N < 0 || H <= 0 || Hprime <= 0 || S < 0 || lds < H ||
        !(mode & 3) || ((wmu == nullptr) != (ymu == nullptr)))
        return PM_EINVAL;
    if ((mode & 2) && (!params_host || !logpj || ldl < 1 + H + S || gamma < 1 || gamma > Hprime ||
                       (S > 0 && (!state_masks || !state_parents || !size_offsets_host))))
        return PM_EINVAL;
    if (!pm_bsc_rows16_supported(H, Hprime, S)) return PM_ERANGE;
    if (N == 0) return PM_OK;
    SizeOffsets so;
    for (int g = 0; g < PM_MAX_HPRIME; ++g) so.off[g] = (int)S;
    if ((mode & 2) && S > 0)
        for (int g = 0; g < gamma; ++g) so.off[g] = size_offsets_host[g];  // off[g-2] = first state of size g
    pm_bsc_estep_params P = params_host ? *params_host : pm_bsc_estep_params{0, 0, 0, 0};
    const size_t shmem = (size_t)make_layout((int)H, (int)Hprime, (int)S, 0).bytes;
    dim3 grid((unsigned)grid_groups(N)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PM_LAUNCH_E(V, E)                                                                                           \
    do {                                                                                                            \
        if (int e = allow_lds16(reinterpret_cast<const void *>(bsc_select_estep16_kernel<V, E>), shmem)) return e;  \
        hipLaunchKernelGGL((bsc_select_estep16_kernel<V, E>), grid, block, shmem, s, scores, lds, gram, ynorm2, wmu,\
                           ymu, state_masks, state_parents, so, (int)S, (int)gamma, P, N, (int)H, (int)Hprime,     \
                           mode, cand, logpj, ldl, lse);                                                            \
    } while (0)
#define PM_LAUNCH(V)                \
    do {                            \
        if (mode & 2) {             \
            PM_LAUNCH_E(V, true);   \
        } else {                    \
            PM_LAUNCH_E(V, false);  \
        }                           \
    } while (0)
    if (H <= 16) PM_LAUNCH(1);
    else if (H <= 32) PM_LAUNCH(2);
    else if (H <= 64) PM_LAUNCH(4);
    else if (H <= 128) PM_LAUNCH(8);
    else if (H <= 256) PM_LAUNCH(16);
    else PM_LAUNCH(32);
#undef PM_LAUNCH
#undef PM_LAUNCH_E
    return (int)hipGetLastError();
}

// select_Hprimes of DSC (params given) / TSC (params NULL) in one pass: see xsc_select16_kernel.  PM_ERANGE where it does
// not apply (then: pm_dsc_select_scores_f64 / pm_tsc_select_scores_f64 + pm_bsc_select_estep_f64).
extern "C" int pm_xsc_select_supported(int64_t H, int64_t Hprime, int tsc) {
    const int64_t HK = tsc ? 2 * H : H;
    if (!pm_bsc_rows16_supported(HK, Hprime, 0)) return 0;
    if (tsc && !(H == 16 || H == 32 || H == 64 || H == 128 || H == 256)) return 0;
    return 1;
}

extern "C" int pm_xsc_select_f64(const double *scores, int64_t lds, const double *gram, const pm_dsc_params *params_host,
                                 int64_t N, int64_t H, int64_t Hprime, int32_t *cand, void *stream) {
    if (!scores || !gram || !cand || N < 0 || H <= 0 || Hprime <= 0 || lds < H) return PM_EINVAL;
    const bool tsc = params_host == nullptr;
    if (!tsc && (params_host->K < 2 || params_host->K > PM_DSC_MAX_K || params_host->K0 < 0 || params_host->K0 >= params_host->K))
        return PM_EINVAL;
    if (!pm_xsc_select_supported(H, Hprime, tsc ? 1 : 0)) return PM_ERANGE;
    if (N == 0) return PM_OK;
    const int64_t HK = tsc ? 2 * H : H;
    const size_t shmem = (size_t)make_layout((int)HK, (int)Hprime, 0, 0).bytes;
    dim3 grid((unsigned)grid_groups(N)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    pm_dsc_params P{};
    if (!tsc) P = *params_host;
#define PM_LAUNCH_X(V, T)                                                                                    \
    do {                                                                                                     \
        if (int e = allow_lds16(reinterpret_cast<const void *>(xsc_select16_kernel<V, T>), shmem)) return e; \
        hipLaunchKernelGGL((xsc_select16_kernel<V, T>), grid, block, shmem, s, scores, lds, gram, P, N, (int)H, \
                           (int)Hprime, cand);                                                               \
    } while (0)
    if (tsc) {
        if (H == 16) PM_LAUNCH_X(1, true);
        else if (H == 32) PM_LAUNCH_X(2, true);
        else if (H == 64) PM_LAUNCH_X(4, true);
        else if (H == 128) PM_LAUNCH_X(8, true);
        else PM_LAUNCH_X(16, true);
    } else {
        if (H <= 16) PM_LAUNCH_X(1, false);
        else if (H <= 32) PM_LAUNCH_X(2, false);
        else if (H <= 64) PM_LAUNCH_X(4, false);
        else if (H <= 128) PM_LAUNCH_X(8, false);
        else if (H <= 256) PM_LAUNCH_X(16, false);
        else PM_LAUNCH_X(32, false);
    }
#undef PM_LAUNCH_X
    return (int)hipGetLastError();
}

extern "C" int pm_bsc_mstep_rows16_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                                       const int32_t *cand, const uint16_t *state_masks, int64_t S,
                                       const pm_bsc_estep_params *params_host, int64_t N, int64_t H, int64_t D,
                                       int64_t Hprime, double *expect, int64_t lde, double *stats, void *stream) {
    return pm_bsc_mstep_rows16_nz_f64(logpj, ldl, lse, lse_cut, cand, state_masks, S, params_host, N, H, D, Hprime, expect,
                                      lde, stats, nullptr, nullptr, stream);
}

extern "C" int pm_bsc_mstep_rows16_nz_f64(const double *logpj, int64_t ldl, const double *lse, double lse_cut,
                                          const int32_t *cand, const uint16_t *state_masks, int64_t S,
                                          const pm_bsc_estep_params *params_host, int64_t N, int64_t H, int64_t D,
                                          int64_t Hprime, double *expect, int64_t lde, double *stats, uint16_t *nz_idx,
                                          double *nz_val, void *stream) {
    if ((nz_idx == nullptr) != (nz_val == nullptr)) return PM_EINVAL;
    if (!logpj || !lse || !cand || !params_host || !expect || !stats || N < 0 || H <= 0 || D <= 0 || Hprime <= 0 ||
        S < 0 || ldl < 1 + H + S || lde < H || (S > 0 && !state_masks))
        return PM_EINVAL;
    if (!pm_bsc_rows16_supported(H, Hprime, S)) return PM_ERANGE;
    if (N == 0) return PM_OK;
    if (nz_idx && !pm_bsc_rows16_nz_supported(H, Hprime, S)) return PM_ERANGE;
    const size_t shmem = mstep_rows16_lds(H, Hprime, S, nz_idx != nullptr);
    if (shmem > PM_ROWS16_LDS_MAX) return PM_ERANGE;
    dim3 grid((unsigned)grid_groups(N)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PM_LAUNCH(V)                                                                                               \
    do {                                                                                                           \
        if (nz_idx) {                                                                                              \
            if (int e = allow_lds16(reinterpret_cast<const void *>(bsc_mstep_rows16_kernel<V, true>), shmem)) return e; \
            hipLaunchKernelGGL((bsc_mstep_rows16_kernel<V, true>), grid, block, shmem, s, logpj, ldl, lse, lse_cut, cand, \
                               state_masks, (int)S, *params_host, N, (int)H, (int)D, (int)Hprime, expect, lde, stats, \
                               nz_idx, nz_val);                                                                    \
        } else {                                                                                                   \
            if (int e = allow_lds16(reinterpret_cast<const void *>(bsc_mstep_rows16_kernel<V, false>), shmem)) return e; \
            hipLaunchKernelGGL((bsc_mstep_rows16_kernel<V, false>), grid, block, shmem, s, logpj, ldl, lse, lse_cut, cand, \
                               state_masks, (int)S, *params_host, N, (int)H, (int)D, (int)Hprime, expect, lde, stats, \
                               nz_idx, nz_val);                                                                    \
        }                                                                                                          \
    } while (0)
    if (H <= 16) PM_LAUNCH(1);
    else if (H <= 32) PM_LAUNCH(2);
    else if (H <= 64) PM_LAUNCH(4);
    else if (H <= 128) PM_LAUNCH(8);
    else if (H <= 256) PM_LAUNCH(16);
    else PM_LAUNCH(32);
#undef PM_LAUNCH
    return (int)hipGetLastError();
}

PM_DET_SETTER(bsc_rows16)
