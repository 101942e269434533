// Gaussian (spike-and-slab) Sparse Coding, scalar observation noise (prosper/em/camodels/gsc_et.py,
// class GSC, sigma_sq_type == 'scalar') on gfx950.
//
// One pass per datapoint over the truncated state set computes, from the scores a = W^T y (f64 MFMA
// GEMM), the Gram matrix G = W^T W and psi_sq:
//   component scores + top-H' (gsc_et.py:721-728, 752-809), candidates sorted by latent index
//   every state's un-normalised posterior weight exp(beta * lp)       (gsc_et.py:316-356, 476-522)
//     Lambda = G_aa / s2 + Psi_a^-1,  b = a_a - G_aa mu_a,  |r|^2 = |y|^2 - 2 mu_a.a_a + mu_a^T G_aa mu_a
//     lp = -(logdet Psi_a + logdet Lambda) - |r|^2 / s2 + b^T Lambda^-1 b / s2^2 + sum logit(pi_a)
//     kappa = Lambda^-1 b / s2 + mu_a,  E[z z^T] = kappa kappa^T + Lambda^-1
//   the normalised expectations xpt_s, xpt_sz (N,H) and the SUMS over datapoints of xpt_ss, xpt_szsz
//   (H,H) -- the reference materialises those per datapoint as (N,H,H) (26 GB each at config 4,
//   gsc_et.py:436-438) although its M-step only ever consumes their sums (gsc_et.py:603-610,662-671).
// Weights follow the reference exactly: un-stabilised exp, NaN / underflow clamped to `tiny`, the null
// state's weight not clamped.
//
// Mapping: 16 lanes per datapoint (four datapoints per wavefront), lane j holds latents h = j + 16 i;
// a lane walks multi-cause states s = j + 16 t and solves their g x g systems (g <= GMAX) in registers;
// per-datapoint accumulators live in LDS; reductions over a datapoint are DPP row butterflies.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"
#include "pm_common.h"

namespace {

constexpr int ROWS = 16;

template <int CTRL>
__device__ __forceinline__ unsigned gdpp32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t gdpp64(uint64_t v) {
    const unsigned lo = gdpp32<CTRL>((unsigned)v), hi = gdpp32<CTRL>((unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ double gdppf(double v) {
    return __longlong_as_double((long long)gdpp64<CTRL>((uint64_t)__double_as_longlong(v)));
}
__device__ __forceinline__ double g_row_max_f64(double v) {
    v = __builtin_fmax(v, gdppf<0xB1>(v));
    v = __builtin_fmax(v, gdppf<0x4E>(v));
    v = __builtin_fmax(v, gdppf<0x141>(v));
    v = __builtin_fmax(v, gdppf<0x140>(v));
    return v;
}
__device__ __forceinline__ double g_row_sum(double v) {
    v += gdppf<0xB1>(v);
    v += gdppf<0x4E>(v);
    v += gdppf<0x141>(v);
    v += gdppf<0x140>(v);
    return v;
}
// sum over the four 16-lane rows of a wave (lanes j, j+16, j+32, j+48); epilogue only
__device__ __forceinline__ double g_col_sum(double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
// Ordering point for the per-datapoint LDS arrays, which only the lanes of ONE wavefront touch: the LDS pipeline
// executes a wavefront's operations in issue order, so a wavefront-scope fence (a compiler barrier; a
// workgroup-scope one also drains the vector-memory counter and with it the pending global atomics) suffices.
__device__ __forceinline__ void g_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 1 / x for a normal x of either sign: hardware reciprocal + two Newton steps (full f64 accuracy up to the
// last bit or two; the IEEE division sequence costs four times as many instructions)
__device__ __forceinline__ double g_recip(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// log|det| accumulated as a product of pivot mantissas and a sum of pivot exponents: ONE log per state instead
// of one per pivot (mantissas lie in [0.5, 1): sixteen of them cannot underflow)
struct LogDet {
    double mant = 1.0;
    int expo = 0;
    __device__ __forceinline__ void times(double piv) {
        mant *= __builtin_amdgcn_frexp_mant(piv);
        expo += __builtin_amdgcn_frexp_exp(piv);
    }
    __device__ __forceinline__ double value() const { return log(fabs(mant)) + (double)expo * 0.6931471805599453; }
};

// In-place inverse of a symmetric G x G matrix whose leading g x g block is live (the rest is the identity),
// pivots multiplied into `ld`: Gauss-Jordan without pivoting, fully unrolled (symmetric positive systems).
template <int G>
__device__ __forceinline__ void sym_inverse(double (&M)[G][G], LogDet &ld) {
#pragma unroll
    for (int p = 0; p < G; ++p) {
        const double piv = M[p][p];
        ld.times(piv);
        const double ip = g_recip(piv);
#pragma unroll
        for (int c = 0; c < G; ++c) M[p][c] *= ip;
        M[p][p] = ip;
#pragma unroll
        for (int r = 0; r < G; ++r) {
            if (r == p) continue;
            const double f = M[r][p];
#pragma unroll
            for (int c = 0; c < G; ++c) {
                if (c == p) continue;
                M[r][c] -= f * M[p][c];
            }
            M[r][p] = -f * ip;
        }
    }
}


// The un-stabilised posterior weight exp(x) of the reference (gsc_et.py:354-356: NaN and everything below the smallest
// normal double become `tiny`) from the 2^(j/128) table (pm_exp_tab: 14 instructions, relative error 2.3e-16; libm's exp
// is ~45, and a datapoint evaluates 11-12 of them per lane: a tenth of the kernel's instructions).  exp(x) < tiny exactly
// when x < log(tiny): decided on the argument; arguments above 700 (never seen: a log-joint) take libm.
// (round 6: the table form saturates at exp(-708.0); an argument in [log(tiny), -708.0) -- a weight within a factor 1.5 of the
// clamp, i.e. a datapoint ALL of whose states have underflowed: 8 of config 4's 200 000 -- used to get exp(-708.0), up to 48 %
// too much; found by comparing every row of the shard with the oracle, test_config4_full_shard_against_oracle.  libm there.)
__device__ __forceinline__ double gsc_weight(double x, const double *etab) {
    const double tiny = 2.2250738585072014e-308;
    if (__builtin_expect(x > 700.0 || (x < -708.0 && x >= -708.3964185322641), 0)) {
        const double e = exp(x);
        return e < tiny ? tiny : e;
    }
    double p = pm_exp_tab(x, etab);
    if (!(x >= -708.3964185322641)) p = tiny;            // (NaN included)
    return p;
}

// per-latent tables (H doubles each), prepared on the host per EM step
struct GscTables {
    const double *c0;    // nc_h - mu_h^2 G_hh / s2,  nc_h = -(log psi_hh + log lam_h)
    const double *c1;    // 2 mu_h / s2
    const double *gm;    // G_hh mu_h
    const double *il;    // 1 / (lam_h s2^2)
    const double *kl;    // 1 / (lam_h s2)
    const double *ilam;  // 1 / lam_h
    const double *mu;
    const double *lpi;   // log(pi_h) - log(1 - pi_h)
};

// LPJ: also write every state's log-joint (no annealing, prior included -- what compute_lpj returns,
// gsc_et.py:811-944) to logpj (N, 1 + H + S): [null ; singletons h = 0..H-1 ; multi-cause states in table order].
// LACC: the column sums of xpt_s / xpt_sz and the singletons' diagonal of sum xpt_szsz are accumulated in LDS -- [3][4
// wavefronts][H] accumulators, ds_add_f64 per lane and datapoint row, a copy per wavefront so that the four never contend --
// and folded at the end: no second pass over the N x H moments (gsc_colsum_kernel read 410 MB again at config 4: 0.09 ms).
#ifndef PM_GSC_LAUNDER
#define PM_GSC_LAUNDER 1
#endif
#ifndef PM_GSC_ABL
#define PM_GSC_ABL 0      // timing-only ablation builds (scratch/gsc_abl.sh): bits switch phases off, results are wrong
#endif
// -DPM_GSC_STAMPS (scratch/gsc_stamps.py, never in the shipped library): wavefront 0 of every 37th workgroup writes s_memrealtime
// (100 MHz) at the phase boundaries of its first datapoints into pm_gsc_stamps[workgroup / 37][datapoint][phase]
#ifdef PM_GSC_STAMPS
__device__ unsigned long long pm_gsc_stamps[32][12][10];
#define GSC_STAMP(ph)                                                                                          \
    do {                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 32 && stamp_dp < 12)                  \
            pm_gsc_stamps[blockIdx.x / 37][stamp_dp][ph] = __builtin_amdgcn_s_memrealtime();                    \
    } while (0)
#else
#define GSC_STAMP(ph)
#endif
#ifndef PM_GSC_WPE
#define PM_GSC_WPE 3        // wavefronts per SIMD the common instantiations are compiled for (register budget 512 / WPE)
#endif
// LIST (with LACC): the M-step's contraction [Y | xpt_s | xpt_sz]^T . xpt_sz (gsc_et.py:592-625) as a sparse + a dense part.
// A row of xpt_sz whose entries above `thr` number at most PM_BSC_NZ_MAX leaves them as a list (format of the BSC / DSC
// statistics passes: nz_idx uint16 x 16 with 0xFFFF behind the last, nz_val f64 x 16) for pm_wp_sparse_t_f64; any other row
// gets an empty list and its index appended to `dense_rows` (gathered by the workgroup in LDS, ONE global atomic on
// *dense_count per workgroup) for pm_gemm_tn_acc_rows_f64.  thr = tables[8 H + 1]: 2^-57 / N of the smallest |column sum| of
// xpt_sz of the previous EM step (pm_gsc_mstep_finish_f64) -- what a list drops from a column of the product is below
// N thr max|left operand| <= 2^-57 of that column's own scale for ANY N (round 6: N is the all-ranks count the finish kernel gets), under the rounding of the sums
// themselves; thr = 0 (first step, a dead latent's column) keeps everything: every row is dense then, correct and slow.
constexpr int GSC_DENSE_CAP = 512;          // datapoints per workgroup in LIST mode at most (the launcher sizes the grid)
template <int VPL, int GMAX, bool LPJ, bool LACC, bool LIST = false>
__global__ __launch_bounds__(256, (VPL <= 8 && GMAX <= 3) ? PM_GSC_WPE : 1) void gsc_estep_kernel(const double *__restrict__ scores, int64_t lds,
                                                         const double *__restrict__ gram,
                                                         const double *__restrict__ psi,
                                                         const double *__restrict__ ynorm2, GscTables T,
                                                         const uint16_t *__restrict__ masks, int S, double beta,
                                                         double inv_s2_host, int64_t N, int H, int Hp, int do_select,
                                                         int32_t *__restrict__ cand, double *__restrict__ xpt_s,
                                                         double *__restrict__ xpt_sz, int64_t ldx,
                                                         double *__restrict__ stats, double *__restrict__ logpj,
                                                         int64_t ldl, uint16_t *__restrict__ nz_idx,
                                                         double *__restrict__ nz_val, int32_t *__restrict__ dense_rows,
                                                         int32_t *__restrict__ dense_count, double *__restrict__ blocks,
                                                         int64_t ldb) {
    static_assert(!LIST || (LACC && !LPJ), "lists ride on the statistics form of the kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_ndense, s_dbase;
    __shared__ int32_t s_dense[LIST ? GSC_DENSE_CAP : 1];
    const double thr = LIST ? T.c0[8 * (int64_t)H + 1] : 0.0;
    if (LIST && threadIdx.x == 0) s_ndense = 0;
    // 1 / sigma^2 from the host, or (0 there) from the ninth table row an M-step on the device has left
    const double inv_s2 = (inv_s2_host != 0.0) ? inv_s2_host : T.c0[8 * (int64_t)H];
    // [ 8 tables (H) | per datapoint: ac (16) Gc Pc ass aszsz as (16) asz (16) | state masks (S x u16) ]
    double *s_tab = reinterpret_cast<double *>(smem);
    double *s_c0 = s_tab, *s_c1 = s_tab + H, *s_gm = s_tab + 2 * H, *s_il = s_tab + 3 * H, *s_kl = s_tab + 4 * H;
    double *s_ilam = s_tab + 5 * H, *s_mu = s_tab + 6 * H, *s_lpi = s_tab + 7 * H;
    const int HH = Hp * Hp;
    const int dp_stride = 16 + 4 * HH + 2 * 16;
    double *s_dp = s_tab + 8 * H;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: keep it scalar
    const int j = lane & 15, row = lane >> 4;
    for (int h = tid; h < H; h += 256) {
        s_c0[h] = T.c0[h]; s_c1[h] = T.c1[h]; s_gm[h] = T.gm[h]; s_il[h] = T.il[h]; s_kl[h] = T.kl[h];
        s_ilam[h] = T.ilam[h]; s_mu[h] = T.mu[h]; s_lpi[h] = T.lpi[h];
    }
    // the state masks sit in LDS behind the per-datapoint arrays: the multi-cause loop then issues no
    // vector-memory operation at all (see the deferred pair atomics below)
    uint16_t *s_masks = reinterpret_cast<uint16_t *>(s_dp + ROWS * dp_stride);
    for (int s = tid; s < S; s += 256) s_masks[s] = masks[s];
    // LACC: [3][4 wavefronts][H] accumulators (column sums of xpt_s, of xpt_sz, the singletons' diagonal of sum xpt_szsz) behind the masks (8-byte aligned); the four datapoint rows of a wavefront add
    // into the same slots (LDS atomics, 4-way same-address: 16 instructions per four datapoints, nothing beside the
    // ~3300 the rest of them costs) -- per-row private slots cost 32 KB and the third workgroup per CU
    double *s_acc = reinterpret_cast<double *>(smem + ((8 * (size_t)(8 * H + ROWS * dp_stride) + 2 * (size_t)S + 7) & ~(size_t)7));
    if (LACC)
        for (int e = tid; e < 3 * 4 * H; e += 256) s_acc[e] = 0.0;
    double *acc_mine = s_acc + (size_t)wave * H + j;                    // + 16 i: latent j + 16 i; + 4 * H: next quantity
    __shared__ double s_E[128];                           // 2^(j/128): pm_exp_tab's table
    if (tid < 128) s_E[tid] = pm_powtab_dev[256 + tid];
    const double *etab = s_E - 256;
    double *s_ac = s_dp + (wave * 4 + row) * dp_stride;   // a at the candidates
    double *s_Gc = s_ac + 16, *s_Pc = s_Gc + HH;
    double *s_ass = s_Pc + HH, *s_aszsz = s_ass + HH;
    double *s_as = s_aszsz + HH, *s_asz = s_as + 16;
    __syncthreads();

    const double tiny = 2.2250738585072014e-308, fmin_ = -1.7976931348623157e308;
    // The block sums are accumulated per XCD (pm_common.h: f64 atomics from eight XCDs on the same few thousand
    // lines bounce those lines between the L2s -- measured: a third of this kernel's time).  Copy 0 is the
    // caller-visible slot, copies 1..7 sit behind the documented layout; the launcher folds them.
    double *g_ss = pm_xcd_copy(stats, stats + (2 * (int64_t)H * H + 3 * H), 2 * (int64_t)H * H);
    double *g_szsz = g_ss + (int64_t)H * H;
    // the ONLY per-lane state carried across datapoints: the singleton diagonal of sum xpt_szsz.  (diag of
    // sum xpt_ss = column sum of xpt_s; the column sums of xpt_s / xpt_sz come from gsc_colsum_kernel.)  Phases
    // are ordered so that nothing per-latent is live across the multi-cause loop: the kernel's occupancy is set
    // by that loop's g x g algebra (two waves per SIMD) and not by VPL.
    double dszsz[LACC ? 1 : VPL];
#pragma unroll
    for (int i = 0; i < (LACC ? 1 : VPL); ++i) dszsz[i] = 0.0;
    const int rowbase = lane & 48;
    // The block sums of xpt_ss / xpt_szsz go to global memory as f64 atomics whose round trip is microseconds; any
    // vector-memory wait after them (vmcnt counts in issue order) exposes it.  So a datapoint's atomics are
    // DEFERRED: its LDS accumulators stay put and are sent at the top of the row's NEXT datapoint, after that
    // datapoint's last loads and right before its multi-cause loop, which hides the round trip.
    int myc_prev = 0;
    double nf_prev = 0.0;
    bool pend = false;
    // Entries of a datapoint's blocks below thr_p are not sent.  The pass is bound by its vector-memory traffic, and the sixty
    // f64 atomics per datapoint are the largest single item of it (a build without them runs 0.09 ms of 0.63 faster at config
    // 4); most datapoints put all their weight on one or two states, so that all but a few entries of their blocks are
    // ~1e-20 of the others.  thr_p = 2^-57 / N of the smallest diagonal entry of the PREVIOUS EM step's all-reduced sums (tables[8 H
    // + 2], pm_gsc_mstep_finish_f64; 0 -- everything is sent -- when the tables come from the host): what is dropped from any
    // entry of sum xpt_ss / sum xpt_szsz stays below N thr_p = 2^-57 of the smallest diagonal entry whatever N is, under the
    // rounding of the diagonal sums themselves -- nothing the inverses, the element-wise psi_sq update or sigma_sq can see.
    const double thr_p = (inv_s2_host != 0.0 || H <= 2) ? 0.0 : T.c0[8 * (int64_t)H + 2];

    // previous datapoint's blocks -> global sums (xpt_ss: upper triangle, mirrored by the host; candidates are sorted by
    // index, so ci <= ck for i <= k); accumulators cleared for the next one
    auto flush_pairs = [&](bool clear) {
        for (int p0 = 0; p0 < HH; p0 += 16) {
            const int p = p0 + j;
            const bool ok = p < HH;
            const int i = ok ? p / Hp : 0, k = ok ? p - i * Hp : 0;
            const int ci = __builtin_amdgcn_ds_bpermute((rowbase + i) << 2, myc_prev);
            const int ck = __builtin_amdgcn_ds_bpermute((rowbase + k) << 2, myc_prev);
            if (ok) {
                if (pend && !(PM_GSC_ABL & 16)) {
                    // xpt_ss is symmetric: upper triangle only; xpt_szsz = kappa kappa^T + Lambda^-1 is NOT once psi_sq has
                    // been through an M-step (gsc_et.py:660-675 leaves it non-symmetric): both triangles, as they are
                    // The diagonals stay at home: diag(sum xpt_ss) IS the column sum of xpt_s (s_h^2 = s_h; nobody reads the
                    // diagonal of the block sum), and with the LDS accumulators (LACC) the diagonal of xpt_szsz joins the singletons'
                    // there -- twelve of a datapoint's atomics, and the ones that always clear the threshold.
                    const double vss = s_ass[p] * nf_prev, vzz = s_aszsz[p] * nf_prev;
                    if (k > i && vss > thr_p) pm_atomic_add(g_ss + (int64_t)ci * H + ck, PM_Q(vss, 0));
                    if (LACC && k == i) atomicAdd(&s_acc[2 * 4 * H + wave * H + ci], vzz);
                    else if (__builtin_fabs(vzz) > thr_p) pm_atomic_add(g_szsz + (int64_t)ci * H + ck, PM_Q(vzz, 2));
                }
                if (clear) {
                    s_ass[p] = 0.0;
                    s_aszsz[p] = 0.0;
                }
            }
        }
    };

    const int64_t groups = (N + ROWS - 1) / ROWS;
    // The scores row of the NEXT datapoint is requested before this one's outputs are stored (`apre`): vector-memory
    // operations complete in issue order as far as s_waitcnt can tell, so a row requested BEHIND the sixteen stores of the
    // previous datapoint cannot be waited for without waiting for those stores' acknowledgements too (SQ_WAIT_ANY: 65 % of the
    // wave cycles; a build without the stores ran 0.07 ms faster).
    double apre[VPL];
    double yn_pre = 0.0;
    auto prefetch = [&](int64_t g) {
        const int64_t n = g * ROWS + wave * 4 + row;
        const int64_t nn = (g < groups && n < N) ? n : N - 1;
        const double *ar = scores + nn * lds;
#pragma unroll
        for (int i = 0; i < VPL; ++i) apre[i] = (j + 16 * i < H) ? ar[j + 16 * i] : 0.0;
        yn_pre = ynorm2[nn];
    };
    int stamp_dp = 0;
    (void)stamp_dp;
    prefetch(blockIdx.x);
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int64_t n = grp * ROWS + wave * 4 + row;
        const bool live = n < N;
        const int64_t nn = live ? n : N - 1;
        const double *arow = scores + nn * lds;
        const double yn = yn_pre;
        GSC_STAMP(0);

        // ---- candidates: top-H' scores, then sorted by latent index (gsc_et.py:726-728)
        int myc = 0;
        if (do_select) {
            // keys are doubles whose low 10 mantissa bits carry the latent index (one v_max_f64 instead of a
            // three-instruction 64-bit integer maximum; as in bsc_rows16.hip): equal scores resolve towards the
            // larger index, like the reference's argsort; scores are finite here (clamped above), -inf = taken
            double key[VPL];
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int h = j + 16 * i;
                double kx = -INFINITY;
                if (h < H) {  // singleton log-posterior without prior (gsc_et.py:795-805)
                    const double ai = apre[i];
                    const double bb = ai - s_gm[h];
                    double v = s_c0[h] - yn * inv_s2 + s_c1[h] * ai + bb * bb * s_il[h];
                    if (v != v || v < fmin_) v = fmin_;
                    if (isinf(v)) v = 0.0;
                    uint64_t b = (uint64_t)__double_as_longlong(v);
                    b = (b & ~0x3FFull) | ((b >> 63) ? 0x3FFull - (uint64_t)h : (uint64_t)h);
                    kx = __longlong_as_double((long long)b);
                }
                key[i] = kx;
            }
            uint64_t mine = 0;                                // this lane's own selected latents, bit i
            for (int r = 0; r < ((PM_GSC_ABL & 32) ? 1 : Hp); ++r) {
                double m = key[0];
#pragma unroll
                for (int i = 1; i < VPL; ++i) m = __builtin_fmax(m, key[i]);
                m = g_row_max_f64(m);
#pragma unroll
                for (int i = 0; i < VPL; ++i)
                    if (key[i] == m) { key[i] = -INFINITY; mine |= 1ull << i; }
            }
            // rank of each selected latent among the selected = number of selected latents with a smaller index:
            // latent h = j + 16 i precedes h' = j' + 16 i' iff i < i' or (i == i' and j < j').
            int before_slot = 0;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const bool sel = (mine >> i) & 1ull;
                // selected lanes of this row in slot i: ballot restricted to the row
                const unsigned long long bal = __ballot(sel);
                const unsigned rowbits = (unsigned)((bal >> (row * 16)) & 0xFFFFull);
                const int lower = __builtin_popcount(rowbits & ((1u << j) - 1u));
                if (sel) s_as[before_slot + lower] = (double)(j + 16 * i);   // s_as reused as index scratch
                before_slot += __builtin_popcount(rowbits);
            }
            g_sync();
#if PM_GSC_LAUNDER
            {
                int jq = j;
                asm volatile("" : "+v"(jq));      // (see below: no spilled LDS address, no scratch reload here)
                if (jq < Hp) myc = (int)s_as[jq];
            }
#else
            if (j < Hp) myc = (int)s_as[j];
#endif
            g_sync();
#if PM_GSC_LAUNDER
            {   // (the lane's slot, opaque to the optimiser here: `cand + j` is otherwise a loop-invariant 64-bit address
                // that gets SPILLED at three wavefronts per SIMD -- and a scratch reload is a vector-memory operation:
                // its `s_waitcnt vmcnt(0)` drains the previous datapoint's stores before the candidate store may issue)
                int jq = j;
                asm volatile("" : "+v"(jq));
                if (live && jq < Hp) cand[n * Hp + jq] = myc;
            }
#else
            if (live && j < Hp) cand[n * Hp + j] = myc;
#endif
        } else {
            if (j < Hp) myc = cand[nn * Hp + j];
        }

        // ---- candidate blocks -> LDS, accumulators cleared.  Candidate k of this datapoint sits in lane
        // rowbase + k: ds_bpermute fetches it (uniform trip counts: every source lane stays active)
        GSC_STAMP(1);
#if PM_GSC_LAUNDER
        {
            int jq = j;
            asm volatile("" : "+v"(jq));
            if (jq < Hp) {
                s_ac[jq] = arow[myc];
                s_as[jq] = 0.0;
                s_asz[jq] = 0.0;
            }
        }
#else
        if (j < Hp) {
            s_ac[j] = arow[myc];
            s_as[j] = 0.0;
            s_asz[j] = 0.0;
        }
#endif
        for (int p0 = 0; p0 < HH; p0 += 16) {
            const int p = p0 + j;
            const bool ok = p < HH;
            const int i = ok ? p / Hp : 0, k = ok ? p - i * Hp : 0;
            const int ci = __builtin_amdgcn_ds_bpermute((rowbase + i) << 2, myc);
            const int ck = __builtin_amdgcn_ds_bpermute((rowbase + k) << 2, myc);
            if (ok) {
                s_Gc[p] = (PM_GSC_ABL & 128) ? (ci == ck ? 250.0 : 1.0) : gram[(int64_t)ci * H + ck];
                s_Pc[p] = (PM_GSC_ABL & 128) ? (ci == ck ? 1.1 : 0.0) : psi[(int64_t)ci * H + ck];
            }
        }
        // every load of this datapoint has landed before the atomics below are issued -- said explicitly, so that
        // the compiler has no reason to drain the counter (and with it the atomics) further down
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), other counters untouched
        GSC_STAMP(2);
        g_sync();
        flush_pairs(true);
        g_sync();
        GSC_STAMP(3);

        // ---- multi-cause states
        double Z = (j == 0) ? exp(-yn * inv_s2 * beta) : 0.0;       // null state (not clamped upstream: libm)
        double Zm = 0.0;                                            // (LPJ + blocks: the multi-cause states' weights alone)
        if (LPJ && live && j == 0) logpj[n * ldl] = -yn * inv_s2;
        for (int s0 = 0; s0 < ((PM_GSC_ABL & 1) ? 0 : S); s0 += 16) {
            const int s = s0 + j;
            const bool valid = s < S;
            const unsigned mask = valid ? s_masks[s] : 0u;
            int pos[GMAX];
            int g = 0;
#pragma unroll
            for (int q = 0; q < GMAX; ++q) pos[q] = 0;
            {
                unsigned m = mask;
#pragma unroll
                for (int q = 0; q < GMAX; ++q) {
                    if (m) {
                        pos[q] = __builtin_ctz(m);
                        m &= m - 1;
                        g = q + 1;
                    }
                }
            }
            double Gm[GMAX][GMAX], Lm[GMAX][GMAX], av[GMAX], muv[GMAX];
            double prior = 0.0;
#pragma unroll
            for (int r = 0; r < GMAX; ++r) {
                const bool lr = r < g;
                const int cr = __builtin_amdgcn_ds_bpermute((rowbase + pos[r]) << 2, myc);
                av[r] = lr ? s_ac[pos[r]] : 0.0;
                muv[r] = lr ? s_mu[cr] : 0.0;
                prior += lr ? s_lpi[cr] : 0.0;
#pragma unroll
                for (int c = 0; c < GMAX; ++c) {
                    const bool lc = lr && (c < g);
                    Gm[r][c] = lc ? s_Gc[pos[r] * Hp + pos[c]] : 0.0;
                    Lm[r][c] = lc ? s_Pc[pos[r] * Hp + pos[c]] : ((r == c) ? 1.0 : 0.0);
                }
            }
            LogDet ldet;
            sym_inverse<GMAX>(Lm, ldet);                       // Lm = Psi_a^-1, |det Psi_a|
#pragma unroll
            for (int r = 0; r < GMAX; ++r)
#pragma unroll
                for (int c = 0; c < GMAX; ++c)
                    Lm[r][c] = (r < g && c < g) ? Lm[r][c] + Gm[r][c] * inv_s2 : ((r == c) ? 1.0 : 0.0);
            sym_inverse<GMAX>(Lm, ldet);                       // Lm = Lambda^-1, * |det Lambda|
            const double C_det = ldet.value();
            double bvec[GMAX], r2 = yn, quad = 0.0;
#pragma unroll
            for (int r = 0; r < GMAX; ++r) {
                double gmu = 0.0;
#pragma unroll
                for (int c = 0; c < GMAX; ++c) gmu += Gm[r][c] * muv[c];
                bvec[r] = av[r] - gmu;
                r2 += muv[r] * (gmu - 2.0 * av[r]);
            }
            double kap[GMAX];
#pragma unroll
            for (int r = 0; r < GMAX; ++r) {
                double lb = 0.0;
#pragma unroll
                for (int c = 0; c < GMAX; ++c) lb += Lm[r][c] * bvec[c];
                quad += bvec[r] * lb;
                kap[r] = lb * inv_s2 + muv[r];
            }
            const double lp = -C_det - r2 * inv_s2 + quad * inv_s2 * inv_s2 + prior;
            if (LPJ && live && valid) logpj[n * ldl + 1 + H + s] = lp;
            const double p = gsc_weight(lp * beta, etab);
            if (valid) Z += p;
            if (LPJ && valid) Zm += p;
#pragma unroll
            for (int r = 0; r < GMAX; ++r) {
                if (r < g) {                                   // g == 0 for the padding lanes of the last trip
                    atomicAdd(&s_as[pos[r]], p);
                    atomicAdd(&s_asz[pos[r]], p * kap[r]);
#pragma unroll
                    for (int c = 0; c < GMAX; ++c) {
                        if (c < g) {
                            atomicAdd(&s_ass[pos[r] * Hp + pos[c]], p);
                            atomicAdd(&s_aszsz[pos[r] * Hp + pos[c]], p * (kap[r] * kap[c] + Lm[r][c]));
                        }
                    }
                }
            }
        }

        GSC_STAMP(4);
        if (LPJ && blocks && live) {
            // compute_posterior_hprime (gsc_et.py:260-398): this datapoint's un-normalised sums over the multi-cause states, as
            // they stand in LDS -- [E[s s^T] (H'^2) | E[sz sz^T] (H'^2) | E[s] (H') | E[sz] (H') | sum of the weights]
            g_sync();
            double *bo = blocks + n * ldb;
            for (int p0 = 0; p0 < HH; p0 += 16)
                if (p0 + j < HH) {
                    bo[p0 + j] = s_ass[p0 + j];
                    bo[HH + p0 + j] = s_aszsz[p0 + j];
                }
            if (j < Hp) {
                bo[2 * HH + j] = s_as[j];
                bo[2 * HH + Hp + j] = s_asz[j];
            }
            const double zm = g_row_sum(Zm);
            if (j == 0) bo[2 * HH + 2 * Hp] = zm;
        }
        // ---- singletons (gsc_et.py:752-809 with the prior): the scores row is read again (it is still in L2 /
        // the vector cache) rather than held in registers across the loop above
        int64_t row_off = nn * lds;
        asm volatile("" : "+v"(row_off));
        const double *arow2 = scores + row_off;
        double xs[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            double p = 0.0;
            if (h < H) {
                // the un-clamped singleton log-posterior (the score clamps are selection-only)
                const double ai = arow2[h];
                const double bb = ai - s_gm[h];
                const double lp = s_c0[h] - yn * inv_s2 + s_c1[h] * ai + bb * bb * s_il[h] + s_lpi[h];
                if (LPJ && live) logpj[n * ldl + 1 + h] = lp;
                p = (PM_GSC_ABL & 2) ? lp : gsc_weight(lp * beta, etab);
                Z += p;
            }
            xs[i] = p;
        }
        Z = g_row_sum(Z);
        const double nf = 1.0 / (Z + tiny);
        g_sync();

        GSC_STAMP(5);
        // ---- expectations of this datapoint; sums over datapoints (the block sums: see flush_pairs)
        double xsz[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            double kap = 0.0, il = 0.0;
            if (h < H) {      // (the score once more from the vector cache: eight registers less across the phase)
                kap = (arow2[h] - s_gm[h]) * s_kl[h] + s_mu[h];
                il = s_ilam[h];
            }
            xsz[i] = xs[i] * kap;
            // singles contribute to the diagonals of sum xpt_ss / xpt_szsz (multi-cause diagonal terms went
            // through the block atomics above)
            if (LACC) {
                if (live && h < H) atomicAdd(&acc_mine[8 * H + 16 * i], xs[i] * (kap * kap + il) * nf);
            } else if (live) {
                dszsz[i] += xs[i] * (kap * kap + il) * nf;
            }
        }
        for (int k = 0; k < Hp; ++k) {
            const int c = __builtin_amdgcn_ds_bpermute((rowbase + k) << 2, myc);
            const double as = s_as[k], asz = s_asz[k];
            const bool mine = (c & 15) == j;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                if (mine && (c >> 4) == i) {
                    xs[i] += as;
                    xsz[i] += asz;
                }
        }
        GSC_STAMP(6);
        prefetch(grp + gridDim.x);                          // (ahead of the stores below: see `apre`; unconditional, so
                                                            // that the row is not carried across the loop when unused)
        unsigned long long sigb[LIST ? VPL : 1];            // ballots of "significant" per slot (scalar registers)
        int nsig = 0;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int h = j + 16 * i;
            bool sig = false;
            if (live && h < H) {
                const double vs = xs[i] * nf, vz = xsz[i] * nf;
                if (!(PM_GSC_ABL & 4)) {
                    xpt_s[n * ldx + h] = vs;
                    xpt_sz[n * ldx + h] = vz;
                }
                if (LACC && !(PM_GSC_ABL & 8)) {
                    atomicAdd(&acc_mine[16 * i], vs);
                    atomicAdd(&acc_mine[4 * H + 16 * i], vz);
                }
                sig = LIST && (__builtin_fabs(vz) > thr || vs > thr);
            }
            if (LIST) {
                sigb[i] = __ballot(sig);
                nsig += __builtin_popcount((unsigned)((sigb[i] >> (row * 16)) & 0xFFFFull));
            }
        }
        GSC_STAMP(7);
        if (LIST && live && !(PM_GSC_ABL & 64)) {
            const bool sparse = nsig <= PM_BSC_NZ_MAX;
            uint16_t *li = nz_idx + n * PM_BSC_NZ_MAX;
            if (sparse) {
                double *lv = nz_val + n * PM_BSC_NZ_MAX;
                double *ls = nz_val + (N + n) * PM_BSC_NZ_MAX;      // second plane: xpt_s at the listed entries (pm_gsc_list_pairs_f64)
                int before = 0;
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const unsigned rowbits = (unsigned)((sigb[i] >> (row * 16)) & 0xFFFFull);
                    if ((rowbits >> j) & 1u) {
                        const int slot = before + __builtin_popcount(rowbits & ((1u << j) - 1u));
                        li[slot] = (uint16_t)(j + 16 * i);
                        lv[slot] = xsz[i] * nf;
                        ls[slot] = xs[i] * nf;
                    }
                    before += __builtin_popcount(rowbits);
                }
                if (j >= nsig) li[j] = 0xFFFF;
            } else {
                li[j] = 0xFFFF;
                if (j == 0) s_dense[atomicAdd(&s_ndense, 1)] = (int32_t)n;
            }
        }
        myc_prev = myc;
        nf_prev = nf;
        pend = live;
        g_sync();
        GSC_STAMP(8);
#ifdef PM_GSC_STAMPS
        ++stamp_dp;
#endif
    }
    flush_pairs(false);

    // flush the per-latent sums: lanes of different rows / waves own the same latent, so fold them in LDS first and send ONE
    // global atomic per latent and block -- a per-lane flush puts 256 * VPL atomics per block on H addresses and serialises
    // the grid's tail on them.
    __syncthreads();
    double *g_cs = stats + 2 * (int64_t)H * H;          // [column sums of xpt_s | of xpt_sz | singleton diagonal of sum xpt_szsz]
    if (LIST) {       // the workgroup's dense datapoints -> the global list (order among workgroups: as the atomics land)
        const int nd = s_ndense;
        if (tid == 0) s_dbase = nd ? atomicAdd(dense_count, nd) : 0;
        __syncthreads();
        for (int e = tid; e < nd; e += 256) dense_rows[s_dbase + e] = s_dense[e];
    }
    if (LACC) {       // [column sums of xpt_s | of xpt_sz | singleton diagonal]: the four wavefronts' slots, then one atomic each
        for (int e = tid; e < 3 * H; e += 256) {
            const int q = e / H, h = e - q * H;
            const double *src = s_acc + (size_t)q * 4 * H + h;
            const double v = (src[0] + src[H]) + (src[2 * (size_t)H] + src[3 * (size_t)H]);
            if (v != 0.0) pm_atomic_add(g_cs + e, PM_Q(v, q));           // (categories 0 | 1 | 2 = xpt_s | xpt_sz | xpt_szsz sums)
        }
        return;
    }
    for (int h = tid; h < H; h += 256) s_c0[h] = 0.0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (LACC ? 1 : VPL); ++i) {
        const int h = j + 16 * i;
        const double c_d = g_col_sum(dszsz[i]);             // rows of the wave first
        if (row == 0 && h < H) atomicAdd(&s_c0[h], PM_Q(c_d, 2));
    }
    __syncthreads();
    double *g_dszsz = g_cs + 2 * H;
    for (int h = tid; h < H; h += 256)
        if (s_c0[h] != 0.0) pm_atomic_add(g_dszsz + h, s_c0[h]);
}

// Column sums of xpt_s and xpt_sz (N,H) into stats[2 H^2 ..) : one block per slab of rows, threads over columns.
__global__ __launch_bounds__(256) void gsc_colsum_kernel(const double *__restrict__ xpt_s,
                                                         const double *__restrict__ xpt_sz, int64_t ldx, int64_t N,
                                                         int H, int64_t rows_per_block, double *__restrict__ g_cs,
                                                         double *__restrict__ g_csz) {
    __shared__ double s_part[2][256];
    const int tid = threadIdx.x;
    const int cols = H < 256 ? H : 256;                     // columns walked at once
    const int lanes_r = 256 / cols;                         // row-parallel groups inside the block (H <= 128)
    const int c = tid % cols, rr = tid / cols;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
    for (int h0 = 0; h0 < H; h0 += cols) {
        const int h = h0 + c;
        double a = 0.0, b = 0.0;
        if (h < H && rr < lanes_r) {
            // four rows per trip: eight independent loads in flight per thread (one per trip ran at 3.3 TB/s)
            double a1 = 0.0, a2 = 0.0, a3 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
            int64_t r = r0 + rr;
            for (; r + 3 * lanes_r < r1; r += 4 * lanes_r) {
                const double *ps = xpt_s + r * ldx + h, *pz = xpt_sz + r * ldx + h;
                const int64_t st = (int64_t)lanes_r * ldx;
                const double x0 = ps[0], x1 = ps[st], x2 = ps[2 * st], x3 = ps[3 * st];
                const double z0 = pz[0], z1 = pz[st], z2 = pz[2 * st], z3 = pz[3 * st];
                a += x0; a1 += x1; a2 += x2; a3 += x3;
                b += z0; b1 += z1; b2 += z2; b3 += z3;
            }
            for (; r < r1; r += lanes_r) {
                a += xpt_s[r * ldx + h];
                b += xpt_sz[r * ldx + h];
            }
            a = (a + a1) + (a2 + a3);
            b = (b + b1) + (b2 + b3);
        }
        s_part[0][tid] = a;
        s_part[1][tid] = b;
        __syncthreads();
        if (tid < cols && h0 + tid < H) {
            double sa = 0.0, sb = 0.0;
            for (int q = 0; q < lanes_r; ++q) {
                sa += s_part[0][q * cols + tid];
                sb += s_part[1][q * cols + tid];
            }
            if (sa != 0.0) pm_atomic_add(g_cs + h0 + tid, PM_Q(sa, 0));
            if (sb != 0.0) pm_atomic_add(g_csz + h0 + tid, PM_Q(sb, 1));
        }
        __syncthreads();
    }
}

static int allow_lds_gsc(const void *kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

#ifdef PM_GSC_STAMPS
extern "C" int pm_gsc_stamps_get(void *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pm_gsc_stamps), sizeof(unsigned long long) * 32 * 12 * 10);
}
#endif
extern "C" int64_t pm_gsc_stats_len(int64_t H) { return 2 * H * H + 3 * H + (PM_XCD_COPIES - 1) * 2 * H * H; }

static size_t gsc_shmem(int64_t H, int64_t Hprime, int64_t S) {
    return sizeof(double) * (8 * H + ROWS * (48 + 4 * Hprime * Hprime) + (S + 3) / 4);
}
// ... with the per-wavefront accumulators of the column sums behind it (LACC); used when three workgroups still fit a CU
static size_t gsc_shmem_lacc(int64_t H, int64_t Hprime, int64_t S) {
    return gsc_shmem(H, Hprime, S) + sizeof(double) * (3 * 4 * H + 1);
}

extern "C" int pm_gsc_supported(int64_t H, int64_t Hprime, int64_t gamma) {
    if (H <= 0 || H > 512 || Hprime <= 0 || Hprime > PM_MAX_HPRIME || Hprime > H || gamma < 1 || gamma > 8) return 0;
    int64_t S = 0, c = Hprime;                       // multi-cause states: sum_{g=2..gamma} C(H', g)
    for (int64_t g = 2; g <= gamma && g <= Hprime; ++g) {
        c = c * (Hprime - g + 1) / g;
        S += c;
    }
    return gsc_shmem(H, Hprime, S) <= 64 * 1024 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// The H- and H x H-sized tail of the M-step (gsc_et.py:640-713, scalar sigma_sq) and the tables of the NEXT E-step, in
// one workgroup: with them on the device the next E-step is launched before the host has seen this step's result.
//   pi   = clip(sum_s / N, 5e-5, 1 - 5e-5)                                   (gsc_et.py:640-646)
//   mu   = sum_sz / (sum_s + DBL_EPSILON)                                    (:654)
//   psi  = (mu mu^T o sum_ss + sum_zz - 2 mu[:,None] o xs_xsz) o (sum_ss + eps I)^-1 + eps I     (:660-675)
//   s2   = (sum |y|^2 - trace(xsz_xsz . W^T W)) / N / D + eps                (:703-713)
//   tables: c0, 2 mu / s2, G_hh mu, 1 / (lam s2^2), 1 / (lam s2), 1 / lam, mu, logit(pi), [1 / s2]   (GscTables)
// A parameter that is not learned is taken from `old`.  Every sum is formed in a fixed order.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(1024) void gsc_mstep_finish_kernel(
    const double *__restrict__ xs_xsz, const double *__restrict__ xsz_xsz, const double *__restrict__ sum_ss,
    const double *__restrict__ sum_zz, const double *__restrict__ ss_inv, const double *__restrict__ sum_s,
    const double *__restrict__ sum_sz, const double *__restrict__ sum_yy, const double *__restrict__ gram,
    const double *__restrict__ old, double N, double D, int H, int learn, double *__restrict__ params,
    double *__restrict__ tables) {
    __shared__ double s_mu[256], s_psid[256], s_red[1024];
    __shared__ double s_s2;
    const double eps = 1e-5;
    const int tid = threadIdx.x;
    const double *old_pi = old, *old_mu = old + H, *old_psi = old + 2 * H, *old_s2 = old + 2 * H + (int64_t)H * H;
    double *p_pi = params, *p_mu = params + H, *p_psi = params + 2 * H, *p_s2 = params + 2 * H + (int64_t)H * H;
    if (tid < H) {
        double pi = old_pi[tid];
        if (learn & 1) {
            pi = sum_s[tid] / N;
            pi = pi <= 5e-5 ? 5e-5 : pi;
            pi = pi >= 1.0 - 5e-5 ? 1.0 - 5e-5 : pi;
        }
        p_pi[tid] = pi;
        const double mu = (learn & 2) ? sum_sz[tid] / (sum_s[tid] + 2.220446049250313e-16) : old_mu[tid];
        p_mu[tid] = mu;
        s_mu[tid] = mu;
    }
    __syncthreads();
    double tr = 0.0;
    for (int e = tid; e < H * H; e += 1024) {
        const int i = e / H, j = e - i * H;
        double v = old_psi[e];
        if (learn & 4) {
            v = (s_mu[i] * s_mu[j] * sum_ss[e] + sum_zz[e] - 2.0 * (s_mu[i] * xs_xsz[e])) * ss_inv[e];
            if (i == j) v += eps;
        }
        p_psi[e] = v;
        if (i == j) s_psid[i] = v;
        tr += xsz_xsz[e] * gram[(int64_t)j * H + i];
    }
    s_red[tid] = tr;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) s_red[tid] += s_red[tid + w];
        __syncthreads();
    }
    if (tid == 0) {
        const double s2 = (learn & 8) ? (sum_yy[0] - s_red[0]) / N / D + eps : old_s2[0];
        p_s2[0] = s2;
        s_s2 = s2;
    }
    __syncthreads();
    if (tid < H) {
        const double s2 = s_s2, mu = s_mu[tid], psid = s_psid[tid], gd = gram[(int64_t)tid * H + tid], pi = p_pi[tid];
        const double lam = gd / s2 + 1.0 / psid;
        tables[tid] = -(log(psid) + log(lam)) - mu * mu * gd / s2;
        tables[H + tid] = 2.0 * mu / s2;
        tables[2 * H + tid] = gd * mu;
        tables[3 * H + tid] = 1.0 / (lam * s2 * s2);
        tables[4 * H + tid] = 1.0 / (lam * s2);
        tables[5 * H + tid] = 1.0 / lam;
        tables[6 * H + tid] = mu;
        tables[7 * H + tid] = log(pi) - log(1.0 - pi);
        tables[8 * H + tid] = 1.0 / s2;
    }
    // tables[8 H + 1]: the list threshold of the next E-step (gsc_estep_kernel, LIST) from THIS step's all-reduced column sums
    __syncthreads();
    s_red[tid] = tid < H ? fabs(sum_sz[tid]) : INFINITY;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) s_red[tid] = fmin(s_red[tid], s_red[tid + w]);
        __syncthreads();
    }
    // (2^-57 / N of it: whatever the lists drop from a column over all N datapoints of all ranks stays below 2^-57 of that
    // column's scale -- round-5 advisor finding: the fixed 2^-75 assumed N <= 2^18)
    const double n_all = N > 1.0 ? N : 1.0;
    if (tid == 0 && H > 1) tables[8 * H + 1] = ldexp(s_red[0], -57) / n_all;
    // tables[8 H + 2]: below this an entry of a datapoint's pair blocks is not sent (gsc_estep_kernel, thr_p): 2^-57 / N of the smallest
    // diagonal entry of sum xpt_ss (= sum xpt_s) and sum xpt_szsz
    __syncthreads();
    s_red[tid] = tid < H ? fmin(fabs(sum_s[tid]), fabs(sum_zz[(int64_t)tid * H + tid])) : INFINITY;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) s_red[tid] = fmin(s_red[tid], s_red[tid + w]);
        __syncthreads();
    }
    if (tid == 0 && H > 2) tables[8 * H + 2] = ldexp(s_red[0], -57) / n_all;
}
}  // namespace

extern "C" int pm_gsc_mstep_finish_f64(const double *xs_xsz, const double *xsz_xsz, const double *sum_ss,
                                       const double *sum_zz, const double *ss_inv, const double *sum_s,
                                       const double *sum_sz, const double *sum_yy, const double *gram, const double *old,
                                       double N, int64_t D, int64_t H, int learn, double *params, double *tables,
                                       void *stream) {
    if (!xs_xsz || !xsz_xsz || !sum_ss || !sum_zz || !ss_inv || !sum_s || !sum_sz || !sum_yy || !gram || !old || !params ||
        !tables || !(N > 0.0) || D <= 0 || H <= 0)
        return PM_EINVAL;
    if (H > 256) return PM_ERANGE;
    hipLaunchKernelGGL(gsc_mstep_finish_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), xs_xsz, xsz_xsz,
                       sum_ss, sum_zz, ss_inv, sum_s, sum_sz, sum_yy, gram, old, N, (double)D, (int)H, learn, params, tables);
    return (int)hipGetLastError();
}

static int gsc_estep_launch(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                            const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                            int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                            int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx,
                            double *stats, double *logpj, int64_t ldl, void *stream, uint16_t *nz_idx = nullptr,
                            double *nz_val = nullptr, int32_t *dense_rows = nullptr, int32_t *dense_count = nullptr,
                            double *blocks = nullptr, int64_t ldb = 0) {
    if (N == 0) return PM_OK;
    if (blocks && (!logpj || ldb < 2 * Hprime * Hprime + 2 * Hprime + 1)) return PM_EINVAL;
    if (logpj && ldl < 1 + H + S) return PM_EINVAL;
    if (!scores || !gram || !psi_sq || !ynorm2 || !tables || !cand || !xpt_s || !xpt_sz || !stats || N < 0 || H <= 0 ||
        Hprime <= 0 || S < 0 || lds < H || ldx < H || (S > 0 && !state_masks) || !(sigma_sq >= 0.0))
        return PM_EINVAL;
    if (!pm_gsc_supported(H, Hprime, gamma)) return PM_ERANGE;
    GscTables T{tables, tables + H, tables + 2 * H, tables + 3 * H, tables + 4 * H, tables + 5 * H, tables + 6 * H,
                tables + 7 * H};
    const bool lacc = gsc_shmem_lacc(H, Hprime, S) <= 53 * 1024;
    const size_t shmem = lacc ? gsc_shmem_lacc(H, Hprime, S) : gsc_shmem(H, Hprime, S);
    if (gsc_shmem(H, Hprime, S) > 64 * 1024) return PM_ERANGE;
    // ONE resident round of workgroups (three per CU for the tuned instantiations), each walking its share of the
    // datapoints: 2048 workgroups -- 2.7 rounds, the last one two thirds full, and 2048 table loads / accumulator flushes --
    // ran 0.717 ms at config 4, 768 run 0.677 (1536: 0.70, 3072: 0.71)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    int64_t groups = (N + ROWS - 1) / ROWS;
    if (groups > 3 * (int64_t)cus) groups = 3 * (int64_t)cus;
    if (nz_idx) {      // LIST: a workgroup's dense rows wait in GSC_DENSE_CAP slots of LDS
        if (!lacc || logpj || gamma > 3 || H <= 64 || H > 256 || !nz_val || !dense_rows || !dense_count) return PM_ERANGE;
        const int64_t per_wg = GSC_DENSE_CAP / ROWS, need = ((N + ROWS - 1) / ROWS + per_wg - 1) / per_wg;
        if (groups < need) groups = need;
        if (groups > INT32_MAX) return PM_ERANGE;
    }
    dim3 grid((unsigned)groups), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double inv_s2 = sigma_sq > 0.0 ? 1.0 / sigma_sq : 0.0;   // 0: tables[8 H] holds it (pm_gsc_mstep_finish_f64)
#define PM_LAUNCH_LA(V, G, L, A)                                                                                        \
    do {                                                                                                               \
        if (int e = allow_lds_gsc(reinterpret_cast<const void *>(gsc_estep_kernel<V, G, L, A>), shmem)) return e;       \
        hipLaunchKernelGGL((gsc_estep_kernel<V, G, L, A>), grid, block, shmem, s, scores, lds, gram, psi_sq, ynorm2, T, \
                           state_masks, (int)S, beta, inv_s2, N, (int)H, (int)Hprime, do_select, cand, xpt_s, xpt_sz,  \
                           ldx, stats, logpj, ldl, nullptr, nullptr, nullptr, nullptr, blocks, ldb);                   \
    } while (0)
#define PM_LAUNCH_LIST(V, G)                                                                                           \
    do {                                                                                                               \
        if (int e = allow_lds_gsc(reinterpret_cast<const void *>(gsc_estep_kernel<V, G, false, true, true>), shmem))   \
            return e;                                                                                                  \
        hipLaunchKernelGGL((gsc_estep_kernel<V, G, false, true, true>), grid, block, shmem, s, scores, lds, gram,      \
                           psi_sq, ynorm2, T, state_masks, (int)S, beta, inv_s2, N, (int)H, (int)Hprime, do_select,    \
                           cand, xpt_s, xpt_sz, ldx, stats, logpj, ldl, nz_idx, nz_val, dense_rows, dense_count,       \
                           nullptr, 0);                                                                                \
    } while (0)
#define PM_LAUNCH(V, G)                         \
    do {                                        \
        if (logpj) {                            \
            PM_LAUNCH_LA(V, G, true, false);    \
        } else if (lacc) {                      \
            PM_LAUNCH_LA(V, G, false, true);    \
        } else {                                \
            PM_LAUNCH_LA(V, G, false, false);   \
        }                                       \
    } while (0)
#define PM_BY_G(V)                            \
    do {                                      \
        if (gamma <= 2) PM_LAUNCH(V, 2);      \
        else if (gamma == 3) PM_LAUNCH(V, 3); \
        else if (gamma == 4) PM_LAUNCH(V, 4); \
        else if (gamma <= 6) PM_LAUNCH(V, 6); \
        else PM_LAUNCH(V, 8);                 \
    } while (0)
    if (nz_idx) {
        if (H <= 128) {
            if (gamma <= 2) PM_LAUNCH_LIST(8, 2);
            else PM_LAUNCH_LIST(8, 3);
        } else {
            if (gamma <= 2) PM_LAUNCH_LIST(16, 2);
            else PM_LAUNCH_LIST(16, 3);
        }
    } else if (H <= 16) PM_BY_G(1);
    else if (H <= 32) PM_BY_G(2);
    else if (H <= 64) PM_BY_G(4);
    else if (H <= 128) PM_BY_G(8);
    else if (H <= 256) PM_BY_G(16);
    else PM_BY_G(32);
#undef PM_BY_G
#undef PM_LAUNCH
#undef PM_LAUNCH_LA
#undef PM_LAUNCH_LIST
    {
        const int64_t rows_per_block = 512;
        const int64_t blocks = (N + rows_per_block - 1) / rows_per_block;
        double *g_cs = stats + 2 * H * H;
        const int64_t HH2 = 2 * H * H;
        hipLaunchKernelGGL(pm_fold_copies_kernel, dim3((unsigned)((HH2 + 255) / 256)), dim3(256), 0, s, stats,
                           stats + HH2 + 3 * H, HH2);
        if (!lacc || logpj)
            hipLaunchKernelGGL(gsc_colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, xpt_s, xpt_sz, ldx, N, (int)H,
                               rows_per_block, g_cs, g_cs + H);
    }
    return (int)hipGetLastError();
}

// The E-step kernel's statistics buffer [U_ss | U_zz | cs | csz | dzz] -> the M-step's packed layout
// [sum xpt_ss (H,H) | sum xpt_szsz (H,H) | sum xpt_s (H) | sum xpt_sz (H) | sum |y|^2]: xpt_ss mirrored from its upper
// triangle with the column sums of xpt_s on the diagonal (s_h^2 = s_h), xpt_szsz as accumulated (both triangles) plus the
// singletons' diagonal -- one launch instead of eight small tensor operations per EM step (gsc_et.py:603-610, 662-671).
namespace {
__global__ __launch_bounds__(256) void gsc_pack_kernel(const double *__restrict__ stats, int H,
                                                       const double *__restrict__ yy, double *__restrict__ out) {
    const int64_t HH = (int64_t)H * H;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double *U_ss = stats, *U_zz = stats + HH, *cs = stats + 2 * HH, *csz = cs + H, *dzz = csz + H;
    if (e < HH) {
        const int i = (int)(e / H), j = (int)(e - (int64_t)i * H);
        out[e] = i < j ? U_ss[e] : (i > j ? U_ss[(int64_t)j * H + i] : cs[i]);
        out[HH + e] = U_zz[e] + (i == j ? dzz[i] : 0.0);
    }
    if (e < H) {
        out[2 * HH + e] = cs[e];
        out[2 * HH + H + e] = csz[e];
    }
    if (e == 0) out[2 * HH + 2 * H] = yy[0];
}
}  // namespace

extern "C" int pm_gsc_pack_stats_f64(const double *stats, int64_t H, const double *sum_ynorm2, double *out, void *stream) {
    if (!stats || !sum_ynorm2 || !out || H <= 0) return PM_EINVAL;
    if (H > 512) return PM_ERANGE;
    const int64_t HH = H * H;
    hipLaunchKernelGGL(gsc_pack_kernel, dim3((unsigned)((HH + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       stats, (int)H, sum_ynorm2, out);
    return (int)hipGetLastError();
}

extern "C" int pm_gsc_estep_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                                const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                                int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                                int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx,
                                double *stats, void *stream) {
    return gsc_estep_launch(scores, lds, gram, psi_sq, ynorm2, tables, state_masks, S, gamma, beta, sigma_sq, N, H, Hprime,
                            do_select, cand, xpt_s, xpt_sz, ldx, stats, nullptr, 0, stream);
}

// pm_gsc_estep_f64 that also splits the rows of xpt_sz into listed (sparse) and dense ones for the M-step's contraction
// (gsc_estep_kernel, LIST).  `tables` must carry the threshold in tables[8 H + 1] (pm_gsc_mstep_finish_f64 writes it; 0 =
// every row dense); *dense_count must be 0 at launch; nz_idx / nz_val (N x PM_BSC_NZ_MAX), dense_rows (N).
extern "C" int pm_gsc_lists_supported(int64_t H, int64_t Hprime, int64_t gamma, int64_t D) {
    if (!pm_gsc_supported(H, Hprime, gamma) || gamma > 3 || H <= 64 || H > 256) return 0;
    int64_t S = 0, c = Hprime;
    for (int64_t g = 2; g <= gamma && g <= Hprime; ++g) {
        c = c * (Hprime - g + 1) / g;
        S += c;
    }
    // (the sparse product needs H <= 256, the gathered GEMM whole 128 x 128 tiles of the (D + 2 H) x H output)
    return (gsc_shmem_lacc(H, Hprime, S) <= 53 * 1024 && (D + 2 * H) % 128 == 0 && H % 128 == 0) ? 1 : 0;
}

extern "C" int pm_gsc_estep_lists_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                                      const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                                      int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                                      int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx,
                                      double *stats, uint16_t *nz_idx, double *nz_val, int32_t *dense_rows,
                                      int32_t *dense_count, void *stream) {
    if (!nz_idx || !nz_val || !dense_rows || !dense_count) return PM_EINVAL;
    return gsc_estep_launch(scores, lds, gram, psi_sq, ynorm2, tables, state_masks, S, gamma, beta, sigma_sq, N, H, Hprime,
                            do_select, cand, xpt_s, xpt_sz, ldx, stats, nullptr, 0, stream, nz_idx, nz_val, dense_rows,
                            dense_count);
}

// ---------------------------------------------------------------------------------------------
// xs^T xsz and xsz^T xsz (gsc_et.py:603-610: the two H x H products of first moments) over the LISTED datapoints, from the
// lists alone: a listed datapoint has a handful of entries above the threshold, its share of both products is their outer
// product -- ~25 multiply-adds per datapoint instead of 2 x H x H -- so that the sparse product behind it streams the D
// columns of Y only instead of all of [Y | xpt_s | xpt_sz] (half the bytes at config 4).  A workgroup owns ONE of the two
// products (`kind`), `rows_c` rows of it (all H at H = 128: the accumulator is 128 KB of LDS) and a group of datapoints; a
// wavefront loads the lists of four datapoints per request (sixteen lanes each), then spreads the k x k pairs of one list
// over its 64 lanes: lane p takes (s, t) = (p / k, p % k), fetches both entries with ds_bpermute and adds ONE product.  Dense
// datapoints (empty lists) contribute through pm_gemm_tn_acc_rows_f64.  Round 4 measured this form and kept the simpler
// one (no gain while the contraction was not the longest stream of its phase); round 5: DESIGN.md 4.5b.
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ double bperm_f64(int byte_addr, double v) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v)),
                            __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)));
}

__global__ __launch_bounds__(1024) void gsc_list_pairs_kernel(const uint16_t *__restrict__ nz_idx,
                                                               const double *__restrict__ nz_vs,
                                                               const double *__restrict__ nz_vz, int64_t N, int H,
                                                               int rows_c, int nchunks, int64_t rows_per_group,
                                                               double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double pacc[];          // [rows_c][H]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15;
    const int kc = blockIdx.x % (2 * nchunks), kind = kc & 1, chunk = kc >> 1;
    const int64_t grp = blockIdx.x / (2 * nchunks);
    const int c0 = chunk * rows_c, per = rows_c * H;
    for (int e = tid; e < per; e += 1024) pacc[e] = 0.0;
    __syncthreads();
    const int64_t lo = grp * rows_per_group, hi = (lo + rows_per_group < N) ? lo + rows_per_group : N;
    constexpr int PU = 8;                                   // requests in flight per wavefront (4 lists each)
    for (int64_t n0 = lo; n0 < hi; n0 += 64 * PU) {
        int idx[PU];
        double vl[PU], vz[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int64_t n = n0 + 64 * u + (tid >> 4);
            const bool live = n < hi;
            const int64_t e = (live ? n : lo) * PM_BSC_NZ_MAX + j;
            idx[u] = live ? (int)nz_idx[e] : 0xFFFF;
            vz[u] = nz_vz[e];
            vl[u] = kind ? vz[u] : nz_vs[e];                // the left operand's entries: xs or xsz
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const unsigned long long bal = __ballot(idx[u] != 0xFFFF);
            for (int q = 0; q < 4; ++q) {                   // (uniform)
                const int cnt = __builtin_popcount((unsigned)((bal >> (16 * q)) & 0xFFFFull));   // valid slots lead
                if (cnt == 0) continue;
                const int npairs = cnt * cnt, magic = 65536 / cnt + 1;       // p / cnt = p * magic >> 16 for p < 256
                for (int p0 = 0; p0 < npairs; p0 += 64) {
                    const int p = p0 + lane;
                    const int sl = (p * magic) >> 16, tl = p - sl * cnt;
                    const int as = (16 * q + (sl & 15)) << 2, at = (16 * q + (tl & 15)) << 2;
                    const int is = __builtin_amdgcn_ds_bpermute(as, idx[u]), it = __builtin_amdgcn_ds_bpermute(at, idx[u]);
                    const double a = bperm_f64(as, vl[u]), z = bperm_f64(at, vz[u]);
                    const int r = is - c0;
                    if (p < npairs && r >= 0 && r < rows_c) atomicAdd(&pacc[r * H + it], a * z);
                }
            }
        }
    }
    __syncthreads();
    // (per-XCD copies of the outputs + a fold launch were measured and dropped: 128 groups adding 16 K doubles each are not
    // what this kernel waits for -- 0.112 vs 0.114 ms on the first version)
    const int64_t HH = (int64_t)H * H;
    double *dst = out + (kind ? HH : 0) + (int64_t)c0 * H;
    for (int e = tid; e < per; e += 1024) {
        const double u = pacc[e];
        if (u != 0.0) pm_atomic_add(dst + e, u);
    }
}
}  // namespace

extern "C" int pm_gsc_list_pairs_f64(const uint16_t *nz_idx, const double *nz_val_s, const double *nz_val, int64_t N,
                                     int64_t H, double *out, void *stream) {
    if (!nz_idx || !nz_val_s || !nz_val || !out || N < 0 || H <= 0) return PM_EINVAL;
    if (H > 256 || H % 64 != 0) return PM_ERANGE;
    if (N == 0) return PM_OK;
    const int rows_c = (int)(H <= 128 ? H : 16384 / H), nchunks = (int)(H / rows_c);      // rows_c x H doubles <= 128 KB
    if (H % rows_c != 0) return PM_ERANGE;
#ifndef PM_PAIRS_WGS
#define PM_PAIRS_WGS 256
#endif
    int64_t groups = PM_PAIRS_WGS / (2 * nchunks);
    if (groups < 1) groups = 1;
    int64_t rpg = (N + groups - 1) / groups;
    rpg = (rpg + 63) / 64 * 64;
    groups = (N + rpg - 1) / rpg;
    const size_t shmem = sizeof(double) * (size_t)rows_c * (size_t)H;
    if (int e = (int)hipFuncSetAttribute(reinterpret_cast<const void *>(gsc_list_pairs_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
        return e;
    hipLaunchKernelGGL(gsc_list_pairs_kernel, dim3((unsigned)(2 * nchunks * groups)), dim3(1024), shmem,
                       static_cast<hipStream_t>(stream), nz_idx, nz_val_s, nz_val, N, (int)H, rows_c, nchunks, rpg, out);
    return (int)hipGetLastError();
}


extern "C" int pm_gsc_estep_lpj_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                                    const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                                    int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                                    int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx,
                                    double *stats, double *logpj, int64_t ldl, void *stream) {
    if (!logpj) return PM_EINVAL;
    return gsc_estep_launch(scores, lds, gram, psi_sq, ynorm2, tables, state_masks, S, gamma, beta, sigma_sq, N, H, Hprime,
                            do_select, cand, xpt_s, xpt_sz, ldx, stats, logpj, ldl, stream);
}

// pm_gsc_estep_lpj_f64 that also hands out every datapoint's un-normalised sums over the multi-cause states (what
// GSC.compute_posterior_hprime returns, gsc_et.py:260-398), blocks (N, ldb >= 2 H'^2 + 2 H' + 1):
// [sum_s p_s 1 1^T (H' x H') | sum_s p_s (kappa kappa^T + Lambda^-1) (H' x H') | sum_s p_s (H') | sum_s p_s kappa (H') | sum_s p_s],
// over the candidates in the order of `cand`; p_s = exp(beta lp_s) clamped as the reference clamps it.
extern "C" int pm_gsc_estep_lpj_blocks_f64(const double *scores, int64_t lds, const double *gram, const double *psi_sq,
                                           const double *ynorm2, const double *tables, const uint16_t *state_masks, int64_t S,
                                           int64_t gamma, double beta, double sigma_sq, int64_t N, int64_t H, int64_t Hprime,
                                           int do_select, int32_t *cand, double *xpt_s, double *xpt_sz, int64_t ldx,
                                           double *stats, double *logpj, int64_t ldl, double *blocks, int64_t ldb,
                                           void *stream) {
    if (!logpj || !blocks) return PM_EINVAL;
    return gsc_estep_launch(scores, lds, gram, psi_sq, ynorm2, tables, state_masks, S, gamma, beta, sigma_sq, N, H, Hprime,
                            do_select, cand, xpt_s, xpt_sz, ldx, stats, logpj, ldl, stream, nullptr, nullptr, nullptr, nullptr,
                            blocks, ldb);
}

// component_scores (gsc_et.py:752-809): the singleton log-posterior of every latent WITHOUT the prior, with the
// reference's clamps (NaN and values below the smallest double -> that double, +-inf -> 0), from the scores a = W^T y
// (or (Sigma^-1 W)^T y for diagonal / full noise, as pm_gsc_estep_f64 takes them).
namespace {
__global__ __launch_bounds__(256) void gsc_component_scores_kernel(const double *__restrict__ scores, int64_t lds,
                                                                    const double *__restrict__ ynorm2,
                                                                    const double *__restrict__ tables, double inv_s2,
                                                                    int64_t N, int H, double *__restrict__ out,
                                                                    int64_t ldo) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * H) return;
    const int64_t n = e / H;
    const int h = (int)(e - n * H);
    const double fmin_ = -1.7976931348623157e308;
    const double ai = scores[n * lds + h];
    const double bb = ai - tables[2 * (int64_t)H + h];
    double v = tables[h] - ynorm2[n] * inv_s2 + tables[(int64_t)H + h] * ai + bb * bb * tables[3 * (int64_t)H + h];
    if (v != v || v < fmin_) v = fmin_;
    if (isinf(v)) v = 0.0;
    out[n * ldo + h] = v;
}
}  // namespace

extern "C" int pm_gsc_component_scores_f64(const double *scores, int64_t lds, const double *ynorm2, const double *tables,
                                           double sigma_sq, int64_t N, int64_t H, double *out, int64_t ldo,
                                           void *stream) {
    if (N == 0) return PM_OK;
    if (!scores || !ynorm2 || !tables || !out || N < 0 || H <= 0 || lds < H || ldo < H || !(sigma_sq > 0.0)) return PM_EINVAL;
    const int64_t blocks = (N * H + 255) / 256;
    if (blocks > INT32_MAX) return PM_ERANGE;
    hipLaunchKernelGGL(gsc_component_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       scores, lds, ynorm2, tables, 1.0 / sigma_sq, N, (int)H, out, ldo);
    return (int)hipGetLastError();
}

namespace {
// Deterministic mode inside an EM loop: the quanta of the NEXT E-step and of its M-step's contraction from parameters that exist
// on the device only (the M-step's own solution; the host sets the quanta of an E-step it launches from host-side parameters
// -- gsc_et.py GSC._det_quanta, whose bounds these are).  Posterior means |kappa| <= zb = |mu|_max + min(|y|_max / min_h |W_h|,
// |Psi|_inf |W_a| (|y|_max + |W_a| sqrt(gamma) |mu|_max) / sigma^2), |W_a| <= sqrt(gamma) max_h |W_h|; second moments
// kappa^2 + Lambda^-1 <= zb^2 + max_h psi_hh.  One workgroup; writes this unit's own quanta and leaves them, with the
// contraction's (PM_DET_GEMM, PM_DET_WP_SPARSE: their symbols' addresses come in as arguments), and a copy of all in `out`.
__device__ double det_magic(double bound) {          // 1.5 * 2^(e + 1), 2^e >= bound (DeviceCAModel._magic)
    if (!(bound > 0.0) || !(bound < INFINITY)) return 0.0;
    int e;
    const double m = frexp(bound, &e);                // bound = m 2^e, m in [0.5, 1)
    if (m == 0.5) --e;
    return ldexp(1.5, e + 1);
}

__global__ __launch_bounds__(256) void gsc_det_quanta_kernel(const double *__restrict__ gram, int64_t ldg,
                                                               const double *__restrict__ psi, const double *__restrict__ tables,
                                                               int H, double sqrt_gamma, double ymax, double ynmax, double n,
                                                               double *__restrict__ out, double *__restrict__ gsc_M,
                                                               double *__restrict__ gemm_M, double *__restrict__ sparse_M) {
    // (four wavefronts: the kernel runs beside the scores GEMM and must find room on a CU.  15 us on an idle device, ~130 us
    // beside that GEMM -- s_setprio changes nothing -- and still done before it: the next pass waits for the GEMM, not for this)
    __shared__ double s_red[5][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double w2max = 0.0, w2min = INFINITY, mumax = 0.0, prow = 0.0, pdiag = 0.0;
    for (int h = tid; h < H; h += 256) {
        const double g = gram[(int64_t)h * ldg + h];
        w2max = fmax(w2max, g);
        w2min = fmin(w2min, g);
        mumax = fmax(mumax, fabs(tables[6 * (int64_t)H + h]));
        pdiag = fmax(pdiag, fabs(psi[(int64_t)h * H + h]));
    }
    // |Psi|_inf: a wavefront per row, lanes along it (coalesced), eight rows' loads in flight -- a thread per row read its row
    // element by element: 140 us for a 128 x 128 matrix, on the critical path of the EM loop
    for (int h0 = wave; h0 < H; h0 += 4 * 8) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int h = h0 + 4 * j;
            r[j] = 0.0;
            if (h < H)
                for (int k = lane; k < H; k += 64) r[j] += fabs(psi[(int64_t)h * H + k]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            for (int o = 32; o > 0; o >>= 1) r[j] += __shfl_xor(r[j], o);
            prow = fmax(prow, r[j]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        w2max = fmax(w2max, __shfl_xor(w2max, o));
        w2min = fmin(w2min, __shfl_xor(w2min, o));
        mumax = fmax(mumax, __shfl_xor(mumax, o));
        pdiag = fmax(pdiag, __shfl_xor(pdiag, o));
    }
    if (lane == 0) {
        s_red[0][wave] = w2max;
        s_red[1][wave] = -w2min;
        s_red[2][wave] = mumax;
        s_red[3][wave] = prow;
        s_red[4][wave] = pdiag;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            for (int q = 0; q < 5; ++q) s_red[q][0] = fmax(s_red[q][0], s_red[q][w]);
        const double wmax = sqrt(s_red[0][0]), wmin = sqrt(fmax(-s_red[1][0], 0.0));
        const double inv_s2 = tables[8 * (int64_t)H];
        mumax = s_red[2][0];
        const double wa = sqrt_gamma * wmax;
        const double zb = mumax + fmin(ynmax / fmax(wmin, 1e-300), s_red[3][0] * wa * (ynmax + wa * sqrt_gamma * mumax) * inv_s2);
        const double M0 = det_magic(n), M1 = det_magic(n * zb), M2 = det_magic(n * (zb * zb + s_red[4][0]));
        const double Mg = det_magic(n * fmax(fmax(ymax, 1.0), zb) * zb);
        for (int i = 0; i < 16; ++i) out[i] = 0.0;
        out[0] = M0;
        out[1] = M1;
        out[2] = M2;
        out[8] = Mg;
        for (int i = 0; i < 8; ++i) {      // (the contraction's two kernels add into the same accumulators: one quantum)
            if (gsc_M) gsc_M[i] = i == 0 ? M0 : i == 1 ? M1 : i == 2 ? M2 : 0.0;
            if (gemm_M) gemm_M[i] = i == 0 ? Mg : 0.0;
            if (sparse_M) sparse_M[i] = i == 0 ? Mg : 0.0;
        }
    }
}

// A row list built with an atomic counter (the dense rows of gsc_estep_kernel<LIST>: in the order the workgroups finished) put
// into ASCENDING order, in place: the gathered GEMM behind it then sums its K-slices in an order that does not depend on the
// schedule.  Three small launches: every listed row sets its byte in a flag array of N bytes (plain stores; zero on entry);
// a workgroup per chunk of 8192 rows counts the chunk's set bytes; the same grid again adds the counts in front of its chunk,
// scans its own and writes the rows back in order, clearing the flags as it goes.  [Measured before: ONE workgroup with the
// rows as bits of an LDS bitmap set by ds_or_b64 -- 50 us for 40 000 rows; one workgroup compacting the byte flags -- 61 us:
// a single CU is slow at anything 200 KB wide.]
__global__ __launch_bounds__(256) void row_list_flag_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ count,
                                                             unsigned char *__restrict__ flags, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < *count) {
        const int r = rows[i];
        if (r >= 0 && r < N) flags[r] = 1;
    }
}

// (a chunk = 1024 flag words = 8192 rows, one workgroup: lane-contiguous 8-byte loads)
__global__ __launch_bounds__(1024) void row_list_count_kernel(const unsigned long long *__restrict__ flags8, int words,
                                                               int *__restrict__ counts) {
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, w = blockIdx.x * 1024 + tid;
    int c = w < words ? __popcll(flags8[w] & 0x0101010101010101ull) : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((tid & 63) == 0) s_wave[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) t += s_wave[k];
        counts[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void row_list_write_kernel(int32_t *__restrict__ rows, unsigned long long *__restrict__ flags8,
                                                               int words, const int *__restrict__ counts) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x * 1024 + tid;
    int before = 0;                                          // rows of the chunks in front of this one
    for (int k = tid; k < (int)blockIdx.x; k += 1024) before += counts[k];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if (lane == 0) s_wave[wave] = before;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) t += s_wave[k];
        s_base = t;
    }
    __syncthreads();
    unsigned long long b = w < words ? flags8[w] & 0x0101010101010101ull : 0ull;
    const int mine = __popcll(b);
    int incl = mine;                                         // inclusive scan over the wavefront, then over the 16 wavefronts
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    const int base = s_base;
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int o = base + incl - mine;
    for (int k = 0; k < wave; ++k) o += s_wave[k];
    if (b) flags8[w] = 0ull;
    while (b) {
        rows[o++] = (w << 3) + (__builtin_ctzll(b) >> 3);
        b &= b - 1ull;
    }
}
}  // namespace

extern "C" double *prosper_det_addr_gsc(void);
extern "C" double *prosper_det_addr_gemm(void);
extern "C" double *prosper_det_addr_bsc_wp_sparse(void);

extern "C" int pm_gsc_det_quanta_f64(const double *gram, int64_t ldg, const double *psi_sq, const double *tables, int64_t H,
                                     int64_t gamma, double ymax, double ynmax, double n, double *quanta16, void *stream) {
    if (!gram || !psi_sq || !tables || !quanta16 || H <= 0 || ldg < H || gamma <= 0) return PM_EINVAL;
    if (H > INT32_MAX) return PM_ERANGE;
#ifndef PM_DETERMINISTIC
    return -2;
#else
    static double *gsc_M = prosper_det_addr_gsc(), *gemm_M = prosper_det_addr_gemm(), *sparse_M = prosper_det_addr_bsc_wp_sparse();
    if (!gsc_M || !gemm_M || !sparse_M) return PM_EINVAL;
    hipLaunchKernelGGL(gsc_det_quanta_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), gram, ldg, psi_sq, tables,
                       (int)H, sqrt((double)gamma), ymax, ynmax, n, quanta16, gsc_M, gemm_M, sparse_M);
    return (int)hipGetLastError();
#endif
}

extern "C" int pm_sort_row_list_i32(int32_t *rows, const int32_t *count, int64_t N, unsigned char *flags, void *stream) {
    if (!rows || !count || !flags || N < 0 || (reinterpret_cast<uintptr_t>(flags) & 7)) return PM_EINVAL;
    if (N > INT32_MAX - 1024) return PM_ERANGE;
    if (N == 0) return PM_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(row_list_flag_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rows, count, flags, (int)N);
    const int words = (int)((N + 7) / 8), chunks = (words + 1023) / 1024;
    unsigned long long *f8 = reinterpret_cast<unsigned long long *>(flags);
    int *counts = reinterpret_cast<int *>(flags + (size_t)words * 8);
    hipLaunchKernelGGL(row_list_count_kernel, dim3((unsigned)chunks), dim3(1024), 0, s, f8, words, counts);
    hipLaunchKernelGGL(row_list_write_kernel, dim3((unsigned)chunks), dim3(1024), 0, s, rows, f8, words, counts);
    return (int)hipGetLastError();
}

PM_DET_SETTER(gsc)
