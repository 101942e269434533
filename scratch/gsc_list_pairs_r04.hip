// Round 4, measured and not kept: the two H x H products xs^T xsz / xsz^T xsz of GSC's M-step over the LISTED rows from the
// lists alone (outer products of the lists), so that the sparse product streams the D columns of Y only instead of all of
// [Y | xs | xsz].  Config 4, N = 200k: sparse stream 0.20 -> 0.125 ms, this kernel 0.072 ms (first version, sixteen lanes per
// datapoint: 0.112) -- the EM iteration did not move (1.39-1.42 ms either way), so the simpler form stayed
// (pm_wp_sparse_t_f64 over all D + 2 H columns).  Needs a third list array (the xs values of the listed entries) from the
// E-step kernel.  Not compiled into the library.
// xs^T xsz and xsz^T xsz (gsc_et.py:603-610: the two H x H products of first moments) over the LISTED rows, from the lists
// alone: a listed row has a handful of entries, its share of both products is their outer product -- ~25 multiply-adds per
// datapoint instead of 2 x H x H.  A workgroup owns ONE of the two products (`kind`), `rows_c` rows of it (all H of them at
// H = 128: the accumulator is 128 KB of LDS) and a group of datapoints.  A wavefront loads the lists of four datapoints per
// request (sixteen lanes each), then spreads the k x k pairs of one list over its 64 lanes: lane p takes (s, t) = (p / k,
// p % k), fetches both entries with ds_bpermute and adds ONE product -- seven LDS operations per datapoint (k <= 8).  (First
// version: sixteen lanes per datapoint, entry s broadcast per trip, both products, row chunks: 56 LDS operations per
// datapoint, 0.11 ms, all of it the LDS pipe.)  Dense rows (empty lists) contribute through pm_gemm_tn_acc_rows_f64.
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

