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
                                                            int64_t stride_in, int64_t stride_out) {
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

}  // namespace

extern "C" int pm_spd_inverse_f64(const double *upper, int64_t ldu, const double *diag_add, int64_t n, double *full,
                                  double *inv, int64_t ldo, double *pivots, void *stream) {
    if (!upper || !inv || n <= 0 || ldu < n || ldo < n) return PM_EINVAL;
    if (n > NMAX) return PM_ERANGE;
    hipLaunchKernelGGL(spd_inverse_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), upper, ldu,
                       diag_add, (int)n, full, inv, ldo, pivots, (int64_t)0, (int64_t)0);
    return (int)hipGetLastError();
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
                       ldu, diag_add, (int)n, full, inv, ldo, pivots, stride_in, stride_out);
    return (int)hipGetLastError();
}
