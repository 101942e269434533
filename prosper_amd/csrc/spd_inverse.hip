// In-place Gauss-Jordan inverse of a symmetric positive definite H x H matrix (H <= 256) in ONE
// workgroup: the M-step's W_new = Wq^-1 . Wp (np.linalg.lstsq at bsc_et.py:380; Wq is a sum of
// second moments).  rocSOLVER's potrf + potrs take ~0.9 ms of ~40 small launches for this size; here
// the matrix lives in registers (1024 threads x 8x8 elements, thread (ti,tj) owns A[i][j] with
// i = ti + 32 a, j = tj + 32 b), each of the H elimination steps broadcasts the pivot row and column
// through a double-buffered LDS pair (one barrier per step) and applies the rank-1 update.
// No pivoting (SPD); the smallest / largest pivot are reported so the caller can detect a
// numerically singular matrix and fall back to the host's LAPACK lstsq, as the reference would.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "prosper_hip.h"

#ifndef PM_SPD_ABL
#define PM_SPD_ABL 0   // timing ablations for scratch/spd_bench.hip; 0 = the real kernel
#endif

namespace {

constexpr int TS = 32, EL = 8, NMAX = TS * EL;  // 256

// upper (H,H): upper triangle incl. diagonal of the matrix; diag_add (H) added to the diagonal (may be null).
// Symmetric sweep: sweeping index k maps A -> B with B_kk = -1/A_kk, B_ik = B_ki = A_ik/A_kk,
// B_ij = A_ij - A_ik A_kj / A_kk; after all k, B = -A^-1.  Symmetry is preserved, so a thread keeps only the
// 8x8-block pairs x <= y of its residue class: 36 elements (a whole 256 x 256 f64 matrix would fill the
// CU's entire register file).
__global__ __launch_bounds__(1024) void spd_inverse_kernel(const double *__restrict__ upper, int64_t ldu,
                                                            const double *__restrict__ diag_add, int n,
                                                            double *__restrict__ full, double *__restrict__ inv,
                                                            int64_t ldo, double *__restrict__ pivots,
                                                            int64_t stride_in, int64_t stride_out,
                                                            const double *__restrict__ guard, int guard_n, double guard_tol2,
                                                            const double *__restrict__ refined, int64_t guard_stride) {
    // warm start (pm_spd_inverse_warm_f64): `inv` already holds a Newton-Schulz refinement of the previous inverse
    // that started from a residual ||I - A X0||_F^2 = sum(guard[0 .. guard_n)) -- summed in a fixed order, so every
    // rank holding the same matrices decides alike.  Small enough: nothing to do.
    if (guard) {
        __shared__ double s_g[NMAX];
        __shared__ int s_skip;
        guard += blockIdx.x * guard_stride;              // (matrix blockIdx.x of a batch)
        if (threadIdx.x < NMAX) s_g[threadIdx.x] = (int)threadIdx.x < guard_n ? guard[threadIdx.x] : 0.0;
        __syncthreads();
        if (threadIdx.x < 64) {
            // fixed summation order (four consecutive entries per lane, then a butterfly): every rank holding the same
            // matrices gets the same bits -- and not 256 dependent LDS reads by one thread (11 us of this path)
            const int l = threadIdx.x;
            double r2 = ((s_g[4 * l] + s_g[4 * l + 1]) + s_g[4 * l + 2]) + s_g[4 * l + 3];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) r2 += __shfl_xor(r2, off);
            if (l == 0) s_skip = (r2 < guard_tol2) ? 1 : 0;           // (NaN compares false: the sweep runs)
        }
        __syncthreads();
        if (s_skip) return;      // (ns_finish_kernel, which took the same decision, has written inv and the pivots)
    }
    // one workgroup per matrix of a batch (blockIdx.x): independent inverses run on different CUs at once
    upper += blockIdx.x * stride_in;
    inv += blockIdx.x * stride_out;
    if (full) full += blockIdx.x * stride_out;
    if (diag_add) diag_add += (int64_t)blockIdx.x * n;
    pivots += 2 * blockIdx.x;
    __shared__ double s_c[2][NMAX];
    const int tid = threadIdx.x;
    const int ti = tid >> 5, tj = tid & 31;
    double a[EL][EL];  // only x <= y is live
#pragma unroll
    for (int x = 0; x < EL; ++x)
#pragma unroll
        for (int y = 0; y < EL; ++y) {
            if (y < x) continue;
            const int i = ti + TS * x, j = tj + TS * y;
            double v = (i == j) ? 1.0 : 0.0;  // identity padding keeps the live block's inverse intact
            if (i < n && j < n) {
                v = (i <= j) ? upper[(int64_t)i * ldu + j] : upper[(int64_t)j * ldu + i];
                if (i == j && diag_add) v += diag_add[i];
                if (full) {
                    full[(int64_t)i * ldo + j] = v;
                    full[(int64_t)j * ldo + i] = v;
                }
            }
            a[x][y] = v;
        }
    double pmin = INFINITY, pmax = 0.0;
    // k = 32 kx + km; the outer loop is unrolled so that every register-array index below is static
#pragma unroll
    for (int kx = 0; kx < EL; ++kx) {
        const int kend = min(TS, n - TS * kx);             // <= 0 once past the matrix
        for (int km = 0; km < kend; ++km) {
            const int k = TS * kx + km;
            double *cb = s_c[k & 1];
            // publish column k: c[i] = A[min(i,k)][max(i,k)].  For i in block-row x <= kx the element sits at
            // block (x, kx) of the threads with tj == km; for x > kx at block (kx, x) of the threads with ti == km.
            if (PM_SPD_ABL != 3 && tj == km) {
#pragma unroll
                for (int x = 0; x <= kx; ++x) cb[ti + TS * x] = a[x][kx];
            }
            if (PM_SPD_ABL != 3 && ti == km) {
#pragma unroll
                for (int y = kx + 1; y < EL; ++y) cb[tj + TS * y] = a[kx][y];
            }
            if (PM_SPD_ABL != 4) __syncthreads();
            const double p = cb[k];
            const double ip = (PM_SPD_ABL == 1) ? p * 0.5 : 1.0 / p;
            pmin = fmin(pmin, p);
            pmax = fmax(pmax, p);
            double ci[EL];
#pragma unroll
            for (int x = 0; x < EL; ++x) ci[x] = (PM_SPD_ABL == 2) ? p + x : cb[ti + TS * x];
            // rank-1 update of everything, then the few threads holding row / column k overwrite those entries
#pragma unroll
            for (int y = 0; y < EL; ++y) {
                const double cjy = ((PM_SPD_ABL == 2) ? p - y : cb[tj + TS * y]) * ip;
#pragma unroll
                for (int x = 0; x <= y; ++x) a[x][y] = fma(-ci[x], cjy, a[x][y]);
                asm volatile("" ::: "memory");             // keep the LDS reads of later columns from being hoisted
            }
            if (tj == km) {                                // column k: B_ik = A_ik / p
#pragma unroll
                for (int x = 0; x <= kx; ++x) a[x][kx] = ci[x] * ip;
            }
            if (ti == km) {                                // row k: B_kj = A_kj / p, and B_kk = -1/p
#pragma unroll
                for (int y = kx; y < EL; ++y) a[kx][y] = (tj == km && y == kx) ? -ip : cb[tj + TS * y] * ip;
            }
        }
    }
#pragma unroll
    for (int x = 0; x < EL; ++x)
#pragma unroll
        for (int y = 0; y < EL; ++y) {
            if (y < x) continue;
            const int i = ti + TS * x, j = tj + TS * y;
            if (i < n && j < n && (x < y || i <= j)) {   // diagonal blocks are held in full: emit one triangle
                inv[(int64_t)i * ldo + j] = -a[x][y];
                inv[(int64_t)j * ldo + i] = -a[x][y];
            }
        }
    if (tid == 0 && pivots) {
        pivots[0] = pmin;
        pivots[1] = pmax;
    }
}

// ---- warm start: Newton-Schulz refinement of the previous EM step's inverse ---------------------------------------
// The sweep above is a chain of n dependent pivots (publish -> barrier -> read -> divide -> update, 1.1 us each: 0.3 ms
// at n = 256 however the work is spread -- ablations in scratch/spd_bench.hip: its LDS reads cost 10 us, the division
// 18 us, the f64 VALU work 80 us).  Between two EM steps the second-moment matrix moves little, so the previous step's
// inverse X0 is an excellent approximate inverse: with R0 = I - A X0,
//     X_{k+1} = X_k + X_k R_k,    R_{k+1} = R_k^2        =>   ||R_3|| <= ||R_0||^8
// and every step is two small GEMMs on the matrix cores of 64 CUs (v_mfma_f64_16x16x4_f64 straight from L2), no
// dependent chain.  The sweep kernel still runs behind it and returns at once if ||R_0||_F < 0.1 (guard above); the
// caller's solve adds a step of iterative refinement against A itself (DeviceCAModel._apply_inverse).
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NS_T = 16;      // a workgroup owns one 16 x 16 output tile; its four wavefronts split K

// acc0 += sum_k Pt0[k][i0 + .] * Q[k][j0 + .] (and acc1 with Pt1, if given) over this wavefront's quarter of k
// (row-major, leading dimension ld; rows / columns >= n count as 0).  Fragment maps of v_mfma_f64_16x16x4_f64:
// A lane l = (i = l & 15, k = l >> 4), B lane l = (k = l >> 4, j = l & 15).  All of a wavefront's operands are
// requested before the first MFMA: the kernel is one L2 round trip long, not one per K-step.
template <bool TWO, bool UPPER>
__device__ __forceinline__ void ns_tile(const double *__restrict__ Pt0, const double *__restrict__ Pt1,
                                        const double *__restrict__ Q, const double *__restrict__ diag_add, int i0, int j0,
                                        int n, int64_t ldp, int64_t ld, int lane, int wave, d4 &acc0, d4 &acc1) {
    constexpr int KSTEPS = NMAX / 4 / 4;                       // 16 K-steps of 4 per wavefront
    const int i = i0 + (lane & 15), j = j0 + (lane & 15), kq = lane >> 4;
    const int ic = min(i, n - 1), jc = min(j, n - 1);
    double a0[KSTEPS], a1[KSTEPS], b[KSTEPS];
#pragma unroll
    for (int u = 0; u < KSTEPS; ++u) {
        const int k = (wave * KSTEPS + u) * 4 + kq;
        const int kc = min(k, n - 1);
        if (UPPER) {      // Pt0 is the upper triangle of a symmetric matrix (+ diag_add on the diagonal)
            a0[u] = Pt0[(int64_t)min(kc, ic) * ldp + max(kc, ic)];
            if (kc == ic && diag_add) a0[u] += diag_add[ic];
        } else {
            a0[u] = Pt0[(int64_t)kc * ldp + ic];
        }
        if (TWO) a1[u] = Pt1[(int64_t)kc * ld + ic];
        b[u] = Q[(int64_t)kc * ld + jc];
        const bool kv = k < n;
        a0[u] = (kv && i < n) ? a0[u] : 0.0;
        if (TWO) a1[u] = (kv && i < n) ? a1[u] : 0.0;
        b[u] = (kv && j < n) ? b[u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < KSTEPS; ++u) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b[u], acc0, 0, 0, 0);
        if (TWO) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b[u], acc1, 0, 0, 0);
    }
}

// GENERAL (non-symmetric) matrices -- GSC's sum xpt_szsz from the second EM step on (gsc_et.py:625 inverts it as it is;
// psi_sq stops being symmetric at gsc_et.py:660-675) -- take the LEFT-sided iteration, whose products need no transposed
// copy of X:  R' = I - X B,  X <- X + R' X,  R' <- R' R'  (with L' = R'^T kept beside R' as in the symmetric form).
// The tile below is acc0 += sum_k P[k][i] Q0[k][j] (and acc1 with Q1): one left operand, two right ones; PT / QT read the
// operand through transposed indices (P[k][i] := Pm[i][k], Q0[k][j] := Q0m[j][k] + [j == k] qdiag[j]): the residual kernel
// forms X0 B for B = A^T out of the row-major X0 and A themselves, i.e. the iteration converges to (A^T)^-1 = (A^-1)^T --
// the left operand GSC's W_new^T = (A^-1)^T Wp^T wants.
template <bool TWOQ, bool PT, bool QT>
__device__ __forceinline__ void ns_tile_g(const double *__restrict__ Pm, const double *__restrict__ Q0,
                                          const double *__restrict__ Q1, const double *__restrict__ qdiag, int i0, int j0,
                                          int n, int64_t ldp, int64_t ldq, int lane, int wave, d4 &acc0, d4 &acc1) {
    constexpr int KSTEPS = NMAX / 4 / 4;
    const int i = i0 + (lane & 15), j = j0 + (lane & 15), kq = lane >> 4;
    const int ic = min(i, n - 1), jc = min(j, n - 1);
    double a[KSTEPS], b0[KSTEPS], b1[KSTEPS];
#pragma unroll
    for (int u = 0; u < KSTEPS; ++u) {
        const int k = (wave * KSTEPS + u) * 4 + kq;
        const int kc = min(k, n - 1);
        a[u] = PT ? Pm[(int64_t)ic * ldp + kc] : Pm[(int64_t)kc * ldp + ic];
        b0[u] = QT ? Q0[(int64_t)jc * ldq + kc] : Q0[(int64_t)kc * ldq + jc];
        if (QT && qdiag && kc == jc) b0[u] += qdiag[jc];
        if (TWOQ) b1[u] = Q1[(int64_t)kc * ldq + jc];
        const bool kv = k < n;
        a[u] = (kv && i < n) ? a[u] : 0.0;
        b0[u] = (kv && j < n) ? b0[u] : 0.0;
        if (TWOQ) b1[u] = (kv && j < n) ? b1[u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < KSTEPS; ++u) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b0[u], acc0, 0, 0, 0);
        if (TWOQ) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b1[u], acc1, 0, 0, 0);
    }
}

// the four wavefronts' partial tiles summed in a fixed order; every wavefront returns the total
__device__ __forceinline__ d4 ns_reduce(d4 acc, double (*s_t)[4][64], int lane, int wave) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s_t[wave][r][lane] = acc[r];
    __syncthreads();
    d4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = ((s_t[0][r][lane] + s_t[1][r][lane]) + s_t[2][r][lane]) + s_t[3][r][lane];
    return t;
}

// A = U + U^T - diag(U) + diag(diag_add) -> full;  R = I - A X (A, X symmetric),  L = R^T,
// partial[tile] = sum of the tile's R^2
__global__ __launch_bounds__(256) void ns_residual_kernel(const double *__restrict__ upper, int64_t ldu,
                                                          const double *__restrict__ diag_add,
                                                          const double *__restrict__ X, int n, int64_t ld,
                                                          double *__restrict__ full, double *__restrict__ R,
                                                          double *__restrict__ L, double *__restrict__ partial,
                                                          int64_t stride_in, int64_t stride_x, int64_t stride_full,
                                                          int64_t stride_work, unsigned general_mask,
                                                          double *__restrict__ rowabs) {
    __shared__ double s_t[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.y * NS_T, j0 = blockIdx.x * NS_T;
    const bool gen = (general_mask >> blockIdx.z) & 1u;  // this matrix is a general one, given in full: see ns_tile_g
    upper += blockIdx.z * stride_in;                     // matrix blockIdx.z of a batch
    if (diag_add) diag_add += (int64_t)blockIdx.z * n;
    X += blockIdx.z * stride_x;
    if (full) full += blockIdx.z * stride_full;
    R += blockIdx.z * stride_work;
    L += blockIdx.z * stride_work;
    partial += blockIdx.z * stride_work;
    if (rowabs) rowabs += blockIdx.z * stride_work;
    if (full) {   // this tile of the assembled matrix: one element per thread
        const int i = i0 + (threadIdx.x >> 4), j = j0 + (threadIdx.x & 15);
        if (i < n && j < n) {
            double v = (i <= j || gen) ? upper[(int64_t)i * ldu + j] : upper[(int64_t)j * ldu + i];
            if (i == j && diag_add) v += diag_add[i];
            full[(int64_t)i * ld + j] = v;
        }
    }
    d4 t = {0, 0, 0, 0}, unused = {0, 0, 0, 0};
    if (gen)     // t = X0 A^T: the left residual of the transposed problem
        ns_tile_g<false, true, true>(X, upper, nullptr, diag_add, i0, j0, n, ld, ldu, lane, wave, t, unused);
    else
        ns_tile<false, true>(upper, nullptr, X, diag_add, i0, j0, n, ldu, ld, lane, wave, t, unused);
    t = ns_reduce(t, s_t, lane, wave);
    if (wave != 0) return;
    const int j = j0 + (lane & 15);
    double sq = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + (lane >> 4) + 4 * r;
        const bool in = i < n && j < n;
        if (in) {
            const double v = (i == j ? 1.0 : 0.0) - t[r];
            R[(int64_t)i * ld + j] = v;
            L[(int64_t)j * ld + i] = v;
            sq += v * v;
        }
        if (rowabs) {       // this tile's share of sum_j |(A X0)_ij| (ns_scale_kernel: ||A X0||_inf bounds the spectral radius)
            double ra = in ? fabs(t[r]) : 0.0;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) ra += __shfl_xor(ra, off);
            if ((lane & 15) == 0 && i < n) rowabs[(int64_t)blockIdx.x * n + i] = ra;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    if (lane == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = sq;
}

// The LONG warm path (pm_spd_inverse_warm_long_f64): a start X0 whose residual is not small -- the second-moment matrix
// of a data-truncation step whose kept set jumped, a temperature step -- still converges once it is SCALED: A and X0 are
// symmetric positive definite, so A X0 has real positive eigenvalues, all <= ||A X0||_inf; with alpha = 1 / ||A X0||_inf the
// residual I - alpha A X0 has its spectrum in [0, 1) and Newton-Schulz converges from ANY such start, quadratically once
// below ~0.5: (1 - 1/x)^(2^k) for x = ||A X0||_inf / lambda_min(A X0).  Every workgroup derives alpha from the residual
// kernel's row sums in the same fixed order, then rescales its tile: X' = alpha X0, R' = I - alpha A X0 = alpha R + (1 - alpha) I.
__global__ __launch_bounds__(256) void ns_scale_kernel(const double *__restrict__ X0, int64_t stride_x0,
                                                       double *__restrict__ Xs, double *__restrict__ R,
                                                       double *__restrict__ L, const double *__restrict__ rowabs, int n,
                                                       int tiles, int64_t ld, int64_t stride_work) {
    __shared__ double s_m[256];
    const int tid = threadIdx.x;
    X0 += blockIdx.z * stride_x0;
    Xs += blockIdx.z * stride_work;
    R += blockIdx.z * stride_work;
    L += blockIdx.z * stride_work;
    rowabs += blockIdx.z * stride_work;
    double rs = 0.0;
    if (tid < n)
        for (int t = 0; t < tiles; ++t) rs += rowabs[(int64_t)t * n + tid];
    s_m[tid] = rs;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) s_m[tid] = fmax(s_m[tid], s_m[tid + off]);
        __syncthreads();
    }
    const double nrm = s_m[0];
    const double alpha = (nrm > 0.0 && nrm < INFINITY) ? 1.0 / nrm : 1.0;      // (NaN / inf: the guard rejects the result)
    const int i = blockIdx.y * NS_T + (tid >> 4), j = blockIdx.x * NS_T + (tid & 15);
    if (i < n && j < n) {
        const double r = alpha * R[(int64_t)i * ld + j] + (i == j ? 1.0 - alpha : 0.0);
        R[(int64_t)i * ld + j] = r;
        L[(int64_t)j * ld + i] = r;
        Xs[(int64_t)i * ld + j] = alpha * X0[(int64_t)i * ld + j];
    }
}

// Xn = X + X R,  Rn = R R (= L^T R),  Ln = Rn^T  (LAST: only Xn)
template <bool LAST>
__global__ __launch_bounds__(256) void ns_step_kernel(const double *__restrict__ X, const double *__restrict__ R,
                                                      const double *__restrict__ L, int n, int64_t ld,
                                                      double *__restrict__ Xn, double *__restrict__ Rn,
                                                      double *__restrict__ Ln, int64_t stride_x, int64_t stride_work,
                                                      unsigned general_mask, double *__restrict__ partial_out) {
    __shared__ double s_t[2][4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.y * NS_T, j0 = blockIdx.x * NS_T;
    const bool gen = (general_mask >> blockIdx.z) & 1u;
    X += blockIdx.z * stride_x;                          // matrix blockIdx.z of a batch
    R += blockIdx.z * stride_work;
    L += blockIdx.z * stride_work;
    Xn += blockIdx.z * stride_work;
    if (!LAST) {
        Rn += blockIdx.z * stride_work;
        Ln += blockIdx.z * stride_work;
    }
    d4 xr = {0, 0, 0, 0}, rr = {0, 0, 0, 0};
    if (gen)     // left-sided: R' X = L'^T X and R' R' = L'^T R'
        ns_tile_g<!LAST, false, false>(L, X, R, nullptr, i0, j0, n, ld, ld, lane, wave, xr, rr);
    else
        ns_tile<!LAST, false>(X, L, R, nullptr, i0, j0, n, ld, ld, lane, wave, xr, rr);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s_t[0][wave][r][lane] = xr[r];
        if (!LAST) s_t[1][wave][r][lane] = rr[r];
    }
    __syncthreads();
    if (wave != 0) return;
    const int j = j0 + (lane & 15);
    double sq = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + (lane >> 4) + 4 * r;
        if (i < n && j < n) {
            const double dx = ((s_t[0][0][r][lane] + s_t[0][1][r][lane]) + s_t[0][2][r][lane]) + s_t[0][3][r][lane];
            Xn[(int64_t)i * ld + j] = X[(int64_t)i * ld + j] + dx;
            if (!LAST) {
                const double v = ((s_t[1][0][r][lane] + s_t[1][1][r][lane]) + s_t[1][2][r][lane]) + s_t[1][3][r][lane];
                Rn[(int64_t)i * ld + j] = v;
                Ln[(int64_t)j * ld + i] = v;
                sq += v * v;
            }
        }
    }
    if (!LAST && partial_out) {      // ||R_k||_F^2 of the LAST residual formed: what the guard of the result reads
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
        if (lane == 0) partial_out[blockIdx.z * stride_work + blockIdx.y * gridDim.x + blockIdx.x] = sq;
    }
}

// The warm path's last launch before the (guarded) sweep: every workgroup sums the start residual's per-tile squares in the
// SAME fixed order the sweep kernel uses and, if the refinement is accepted, writes its 16 x 16 tile of
// inv = (X + X^T) / 2 (the next step's start and the solve both treat it as symmetric); workgroup (0, 0) adds the
// conditioning estimate: 1 / X_ii is the pivot row i would get if it were eliminated LAST (its Schur complement against
// all other rows), a lower bound of the pivot the sweep would report for it, and every pivot is at most the matrix's
// own diagonal entry -- pivots = [min_i 1 / X_ii, max_i A_ii].  (One workgroup doing all of this inside the sweep
// kernel took 60 us: 64 K transposed reads through one CU.)
__global__ __launch_bounds__(256) void ns_finish_kernel(const double *__restrict__ X, int n, int64_t stride_x,
                                                        const double *__restrict__ partial, int n_partial,
                                                        double tol2, const double *__restrict__ upper, int64_t ldu,
                                                        int64_t stride_in, const double *__restrict__ diag_add,
                                                        double *__restrict__ inv, int64_t ldo, int64_t stride_out,
                                                        double *__restrict__ pivots, double *__restrict__ accepted,
                                                        int64_t accepted_stride, unsigned general_mask,
                                                        const double *__restrict__ partial0) {
    __shared__ double s_g[NMAX], s_lo[NMAX], s_hi[NMAX];
    __shared__ int s_skip, s_small;
    const int tid = threadIdx.x;
    X += blockIdx.z * stride_x;
    partial += blockIdx.z * stride_x;
    partial0 += blockIdx.z * stride_x;
    s_g[tid] = tid < n_partial ? partial[tid] : 0.0;
    s_lo[tid] = tid < n_partial ? partial0[tid] : 0.0;
    __syncthreads();
    if (tid < 64) {
        double r2 = ((s_g[4 * tid] + s_g[4 * tid + 1]) + s_g[4 * tid + 2]) + s_g[4 * tid + 3];
        double q2 = ((s_lo[4 * tid] + s_lo[4 * tid + 1]) + s_lo[4 * tid + 2]) + s_lo[4 * tid + 3];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            r2 += __shfl_xor(r2, off);
            q2 += __shfl_xor(q2, off);
        }
        if (tid == 0) {
            s_skip = (r2 < tol2) ? 1 : 0;
            s_small = (q2 < 1.0) ? 1 : 0;
        }
    }
    __syncthreads();
    // the decision itself, for the caller: 1 = the refinement stands (inv is exact to rounding) and the START residual was
    // below 1 in the Frobenius norm (the short path would do next time too), 2 = it stands from a start further away (the
    // long path's doing, or luck), 0 = the sweep behind this kernel computes inv (exact to cond(A) eps, as
    // pm_spd_inverse_f64)
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && accepted)
        accepted[blockIdx.z * accepted_stride] = s_skip ? (s_small ? 1.0 : 2.0) : 0.0;
    if (!s_skip) return;
    const int i = blockIdx.y * NS_T + (tid >> 4), j = blockIdx.x * NS_T + (tid & 15);
    const bool gen = (general_mask >> blockIdx.z) & 1u;      // (a general matrix: X = (A^T)^-1 as it stands)
    if (i < n && j < n)
        inv[blockIdx.z * stride_out + (int64_t)i * ldo + j] =
            gen ? X[(int64_t)i * n + j] : 0.5 * (X[(int64_t)i * n + j] + X[(int64_t)j * n + i]);
    if (blockIdx.x == 0 && blockIdx.y == 0 && pivots) {
        double lo = INFINITY, hi = 0.0;
        if (tid < n) {
            const double x = X[(int64_t)tid * n + tid];
            lo = (x > 0.0) ? 1.0 / x : -1.0;             // (a non-positive diagonal: "not positive definite")
            hi = upper[blockIdx.z * stride_in + (int64_t)tid * ldu + tid] + (diag_add ? diag_add[(int64_t)blockIdx.z * n + tid] : 0.0);
        }
        s_lo[tid] = lo;
        s_hi[tid] = hi;
        __syncthreads();
        if (tid < 64) {
            lo = fmin(fmin(s_lo[tid], s_lo[tid + 64]), fmin(s_lo[tid + 128], s_lo[tid + 192]));
            hi = fmax(fmax(s_hi[tid], s_hi[tid + 64]), fmax(s_hi[tid + 128], s_hi[tid + 192]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo = fmin(lo, __shfl_xor(lo, off));
                hi = fmax(hi, __shfl_xor(hi, off));
            }
            if (tid == 0) {
                pivots[2 * blockIdx.z] = lo;
                pivots[2 * blockIdx.z + 1] = hi;
            }
        }
    }
}

}  // namespace

extern "C" int pm_spd_inverse_f64(const double *upper, int64_t ldu, const double *diag_add, int64_t n, double *full,
                                  double *inv, int64_t ldo, double *pivots, void *stream) {
    if (!upper || !inv || n <= 0 || ldu < n || ldo < n) return PM_EINVAL;
    if (n > NMAX) return PM_ERANGE;
    hipLaunchKernelGGL(spd_inverse_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), upper, ldu,
                       diag_add, (int)n, full, inv, ldo, pivots, (int64_t)0, (int64_t)0, (const double *)nullptr, 0, 0.0,
                       (const double *)nullptr, (int64_t)0);
    return (int)hipGetLastError();
}

extern "C" int64_t pm_spd_inverse_warm_work_len(int64_t n) {
    const int64_t tiles = (n + NS_T - 1) / NS_T;
    // (two (X, R, L) triples | the start residual's per-tile squares | the last residual's | row sums of |A X0| per tile
    // column | the accepted flag)
    return n > 0 ? 6 * n * n + 2 * tiles * tiles + tiles * n + 1 : 0;
}

static int launch_warm(const double *upper, int64_t ldu, int64_t stride_in, const double *diag_add, int64_t n,
                       const double *prev_inv, int64_t stride_prev, double *work, double *full, double *inv,
                       int64_t stride_out, double *pivots, int64_t batch, hipStream_t s, unsigned general_mask = 0u,
                       double *accepted = nullptr, int nfull = 3, bool scaled = false) {
    const int tiles = (int)((n + NS_T - 1) / NS_T);
    const int64_t nn = n * n, wl = pm_spd_inverse_warm_work_len(n);
    // work, per matrix: two (X, R, L) triples, then the per-tile sums of squares of the start residual and of the last one
    // formed, the row sums of |A X0|, the accepted flag
    double *X[2] = {work, work + 3 * nn}, *R[2] = {work + nn, work + 4 * nn}, *L[2] = {work + 2 * nn, work + 5 * nn};
    double *partial0 = work + 6 * nn, *partial = partial0 + tiles * tiles, *rowabs = partial + tiles * tiles;
    const dim3 grid((unsigned)tiles, (unsigned)tiles, (unsigned)batch);
    hipLaunchKernelGGL(ns_residual_kernel, grid, dim3(256), 0, s, upper, ldu, diag_add, prev_inv, (int)n, n, full, R[0], L[0],
                       partial0, stride_in, stride_prev, stride_out, wl, general_mask, scaled ? rowabs : (double *)nullptr);
    const double *Xsrc = prev_inv;
    int64_t xstride = stride_prev;
    if (scaled) {
        hipLaunchKernelGGL(ns_scale_kernel, grid, dim3(256), 0, s, prev_inv, stride_prev, X[0], R[0], L[0],
                           (const double *)rowabs, (int)n, tiles, n, wl);
        Xsrc = X[0];
        xstride = wl;
    }
    // nfull steps that also square the residual, then one that only updates X: the result's residual is R_nfull^2.  The
    // guard reads ||R_nfull||_F < 1e-8 -- the LAST residual formed, not the start (round 6: on an annealing ramp the start
    // residual is a few per cent in every direction, 0.3-0.7 in the Frobenius norm, and converges all the same) -- so the
    // refined inverse is exact to rounding whenever it is accepted and the caller's solve needs no refinement pass of its
    // own (two H x H x D products saved per EM step).  Three + one steps: every start with ||R_0||_2 < 0.075.
    int b = 0;
    for (int k = 0; k < nfull; ++k) {
        hipLaunchKernelGGL(ns_step_kernel<false>, grid, dim3(256), 0, s, Xsrc, (const double *)R[b], (const double *)L[b],
                           (int)n, n, X[1 - b], R[1 - b], L[1 - b], xstride, wl, general_mask,
                           k == nfull - 1 ? partial : (double *)nullptr);
        Xsrc = X[1 - b];
        xstride = wl;
        b = 1 - b;
    }
    hipLaunchKernelGGL(ns_step_kernel<true>, grid, dim3(256), 0, s, Xsrc, (const double *)R[b], (const double *)L[b], (int)n,
                       n, X[1 - b], (double *)nullptr, (double *)nullptr, xstride, wl, general_mask, (double *)nullptr);
    const double *Xfin = X[1 - b];
    hipLaunchKernelGGL(ns_finish_kernel, grid, dim3(256), 0, s, Xfin, (int)n, wl, (const double *)partial, tiles * tiles,
                       1e-16, upper, ldu, stride_in, diag_add, inv, n, stride_out, pivots,
                       accepted ? accepted : work + wl - 1, accepted ? (int64_t)1 : wl, general_mask,
                       (const double *)partial0);
    hipLaunchKernelGGL(spd_inverse_kernel, dim3((unsigned)batch), dim3(1024), 0, s, upper, ldu, diag_add, (int)n,
                       (double *)nullptr, inv, n, pivots, stride_in, stride_out, (const double *)partial, tiles * tiles, 1e-16,
                       Xfin, wl);
    return (int)hipGetLastError();
}

extern "C" int pm_spd_inverse_warm_f64(const double *upper, int64_t ldu, const double *diag_add, int64_t n,
                                       const double *prev_inv, int64_t ldp, double *work, double *full, double *inv,
                                       int64_t ldo, double *pivots, void *stream) {
    if (!upper || !inv || !prev_inv || !work || !full || !pivots || n <= 0 || ldu < n || ldo < n || ldp < n) return PM_EINVAL;
    if (n > NMAX) return PM_ERANGE;
    if (ldp != n || ldo != n) return PM_EINVAL;          // the work matrices share one leading dimension with them
    return launch_warm(upper, ldu, 0, diag_add, n, prev_inv, 0, work, full, inv, 0, pivots, 1, static_cast<hipStream_t>(stream),
                       0u, pivots + 2);
}

extern "C" int pm_spd_inverse_warm_long_f64(const double *upper, int64_t ldu, const double *diag_add, int64_t n,
                                            const double *prev_inv, int64_t ldp, double *work, double *full, double *inv,
                                            int64_t ldo, double *pivots, void *stream) {
    if (!upper || !inv || !prev_inv || !work || !full || !pivots || n <= 0 || ldu < n || ldo < n || ldp < n) return PM_EINVAL;
    if (n > NMAX) return PM_ERANGE;
    if (ldp != n || ldo != n) return PM_EINVAL;
    // eight + one steps from the scaled start: (1 - 1/x)^256 < 1e-8 for x = ||A X0||_inf / lambda_min(A X0) up to ~14
    return launch_warm(upper, ldu, 0, diag_add, n, prev_inv, 0, work, full, inv, 0, pivots, 1, static_cast<hipStream_t>(stream),
                       0u, pivots + 2, 8, true);
}

extern "C" int pm_spd_inverse_warm_batch_f64(const double *upper, int64_t ldu, int64_t stride_in, const double *diag_add,
                                             int64_t n, const double *prev_inv, int64_t stride_prev, double *work,
                                             double *inv, int64_t stride_out, double *pivots, int64_t batch, void *stream) {
    if (batch == 0) return PM_OK;
    if (!upper || !inv || !prev_inv || !work || !pivots || n <= 0 || ldu < n || batch < 0 ||
        stride_in < ldu * (n - 1) + n || stride_out < n * n || stride_prev < n * n)
        return PM_EINVAL;
    if (n > NMAX || batch > 65535) return PM_ERANGE;
    return launch_warm(upper, ldu, stride_in, diag_add, n, prev_inv, stride_prev, work, nullptr, inv, stride_out, pivots, batch,
                       static_cast<hipStream_t>(stream));
}

extern "C" int pm_inverse_warm_batch_f64(const double *mats, int64_t ldu, int64_t stride_in, const double *diag_add, int64_t n,
                                         const double *prev_inv, int64_t stride_prev, double *work, double *inv,
                                         int64_t stride_out, double *pivots, double *accepted, int64_t batch,
                                         uint32_t general_mask, void *stream) {
    if (batch == 0) return PM_OK;
    if (!mats || !inv || !prev_inv || !work || !pivots || !accepted || n <= 0 || ldu < n || batch < 0 ||
        stride_in < ldu * (n - 1) + n || stride_out < n * n || stride_prev < n * n)
        return PM_EINVAL;
    if (n > NMAX || batch > 32) return PM_ERANGE;
    return launch_warm(mats, ldu, stride_in, diag_add, n, prev_inv, stride_prev, work, nullptr, inv, stride_out, pivots, batch,
                       static_cast<hipStream_t>(stream), general_mask, accepted);
}

extern "C" int pm_spd_inverse_batch_f64(const double *upper, int64_t ldu, int64_t stride_in, const double *diag_add,
                                        int64_t n, double *full, double *inv, int64_t ldo, int64_t stride_out,
                                        double *pivots, int64_t batch, void *stream) {
    if (batch == 0) return PM_OK;
    if (!upper || !inv || !pivots || n <= 0 || ldu < n || ldo < n || batch < 0 || stride_in < ldu * (n - 1) + n ||
        stride_out < ldo * (n - 1) + n)
        return PM_EINVAL;
    if (n > NMAX || batch > 65535) return PM_ERANGE;
    hipLaunchKernelGGL(spd_inverse_kernel, dim3((unsigned)batch), dim3(1024), 0, static_cast<hipStream_t>(stream), upper,
                       ldu, diag_add, (int)n, full, inv, ldo, pivots, stride_in, stride_out, (const double *)nullptr, 0, 0.0,
                       (const double *)nullptr, (int64_t)0);
    return (int)hipGetLastError();
}
