// Stand-alone timing harness for the fused BSC E-step kernels at config 2 (D=1024 H=256 H'=8 gamma=4): loads a build
// of libprosper_hip.so (argv[1]), runs pm_bsc_estep_fused_f64 / pm_bsc_estep_fused8_f64 on synthetic data and prints
// milliseconds per launch (HIP events).  Used with ablation / stamp builds of the library (scratch/f8_variants.sh);
// correctness is the job of tests/, not of this program.
//   f8_bench <lib.so> <tile 4|8> [N=196608] [reps=20] [stamps 0|1] [mstats 0|1] [part 0|1|2]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../include/prosper_hip.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef int (*fused_fn)(const double *, int64_t, const double *, int64_t, const double *, const double *, const double *,
                        const double *, const uint16_t *, const uint16_t *, const int32_t *, int64_t, int64_t,
                        const pm_bsc_estep_params *, int64_t, int64_t, int64_t, int64_t, int, int32_t *, double *, int64_t,
                        double *, double *, int64_t, double *, int64_t, int, void *);
typedef int (*fused4_fn)(const double *, int64_t, const double *, int64_t, const double *, const double *, const double *,
                         const double *, const uint16_t *, const uint16_t *, const int32_t *, int64_t, int64_t,
                         const pm_bsc_estep_params *, int64_t, int64_t, int64_t, int64_t, int, int32_t *, double *, int64_t,
                         double *, double *, int64_t, double *, int64_t, void *);
typedef int (*gemm_fn)(const double *, int64_t, const double *, int64_t, double *, int64_t, int64_t, int64_t, int64_t, void *);
typedef int (*stamps_fn)(unsigned long long *, int);

static uint64_t rs = 88172645463325252ull;
static inline double urand() {
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (double)(rs >> 11) * (1.0 / 9007199254740992.0);
}
static inline double nrand() {
    const double u = urand() + 1e-300, v = urand();
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

__global__ void gen_data(double *Y, const double *Wgt /* H x D */, int64_t N, int D, int H, uint64_t seed) {
    // y = s . Wgt + noise, s ~ Bernoulli(4/H); one thread per element; cheap hash RNG
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int64_t n = i / D;
    const int d = (int)(i - n * D);
    double acc = 0.0;
    for (int h = 0; h < H; ++h) {
        uint64_t z = seed + (uint64_t)n * 0x9E3779B97F4A7C15ull + (uint64_t)h * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        if ((z & 0xFFFF) < (uint64_t)(65536.0 * 4.0 / H)) acc += Wgt[(int64_t)h * D + d];
    }
    uint64_t z = seed * 31 + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    const double u1 = ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    z = z * 0xD1342543DE82EF95ull + 1; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    const double u2 = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    Y[i] = acc + sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
__global__ void row_sqnorm(const double *Y, double *out, int64_t N, int D) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int d = 0; d < D; ++d) s += Y[n * D + d] * Y[n * D + d];
    out[n] = s;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const char *libpath = argv[1];
    const int tile = atoi(argv[2]);
    const int64_t N = argc > 3 ? atoll(argv[3]) : 196608;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int want_stamps = argc > 5 ? atoi(argv[5]) : 0;
    const int want_mstats = argc > 6 ? atoi(argv[6]) : 0;
    const int D = 1024, H = 256, Hp = 8, gamma = 4;
    void *lib = dlopen(libpath, RTLD_NOW);
    if (!lib) {
        fprintf(stderr, "%s\n", dlerror());
        return 1;
    }
    fused_fn fused = (fused_fn)dlsym(lib, "pm_bsc_estep_fused8_f64");
    fused4_fn fused4 = (fused4_fn)dlsym(lib, "pm_bsc_estep_fused_f64");
    const int part = argc > 7 ? atoi(argv[7]) : 0;
    gemm_fn gemm = (gemm_fn)dlsym(lib, "pm_gemm_nt_f64");
    if (!fused || !gemm) return 1;

    // state table: combinations of Hp positions of size 2..gamma, in generate_state_matrix order
    std::vector<uint16_t> masks, parents;
    std::vector<int32_t> size_off;
    for (int g = 2; g <= gamma; ++g) {
        size_off.push_back((int32_t)masks.size());
        std::vector<int> idx(g);
        for (int i = 0; i < g; ++i) idx[i] = i;
        while (true) {
            uint16_t m = 0;
            for (int i = 0; i < g; ++i) m |= (uint16_t)(1u << idx[i]);
            masks.push_back(m);
            int i = g - 1;
            while (i >= 0 && idx[i] == Hp - g + i) --i;
            if (i < 0) break;
            ++idx[i];
            for (int k = i + 1; k < g; ++k) idx[k] = idx[k - 1] + 1;
        }
    }
    size_off.push_back((int32_t)masks.size());
    const int S = (int)masks.size();
    parents.assign(S, 0xFFFF);
    for (int s = 0; s < S; ++s) {
        const unsigned m = masks[s];
        const int k = 31 - __builtin_clz(m);
        const unsigned rest = m & ~(1u << k);
        if (__builtin_popcount(rest) >= 2)
            for (int q = 0; q < S; ++q)
                if (masks[q] == rest) parents[s] = (uint16_t)q;
    }
    while ((int)size_off.size() < gamma) size_off.push_back(S);

    std::vector<double> Wgt((size_t)H * D), W0((size_t)H * D);
    for (auto &v : Wgt) v = nrand();
    for (size_t i = 0; i < W0.size(); ++i) W0[i] = Wgt[i] + 0.1 * nrand();
    double *dWgt, *dW, *dY, *dG, *dyn, *dlogpj, *dlse;
    int32_t *dcand;
    uint16_t *dmasks, *dparents;
    const int K = 1 + H + S, Kpad = (K + 15) / 16 * 16;
    CK(hipMalloc(&dWgt, sizeof(double) * H * D));
    CK(hipMalloc(&dW, sizeof(double) * H * D));
    CK(hipMalloc(&dY, sizeof(double) * N * D));
    CK(hipMalloc(&dG, sizeof(double) * H * H));
    CK(hipMalloc(&dyn, sizeof(double) * N));
    CK(hipMalloc(&dlogpj, sizeof(double) * N * Kpad));
    CK(hipMalloc(&dlse, sizeof(double) * N));
    CK(hipMalloc(&dcand, sizeof(int32_t) * N * Hp));
    double *dexpect = nullptr, *dstats = nullptr;
    const int64_t nstats = (int64_t)H * D + (int64_t)H * H + 2 * H + 4;
    if (want_mstats) {
        CK(hipMalloc(&dexpect, sizeof(double) * N * H));
        CK(hipMalloc(&dstats, sizeof(double) * nstats));
        CK(hipMemset(dstats, 0, sizeof(double) * nstats));
    }
    CK(hipMalloc(&dmasks, 2 * S));
    CK(hipMalloc(&dparents, 2 * S));
    CK(hipMemcpy(dWgt, Wgt.data(), sizeof(double) * H * D, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, W0.data(), sizeof(double) * H * D, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmasks, masks.data(), 2 * S, hipMemcpyHostToDevice));
    CK(hipMemcpy(dparents, parents.data(), 2 * S, hipMemcpyHostToDevice));
    gen_data<<<(unsigned)((N * D + 255) / 256), 256>>>(dY, dWgt, N, D, H, 12345);
    row_sqnorm<<<(unsigned)((N + 255) / 256), 256>>>(dY, dyn, N, D);
    CK(hipMemset(dG, 0, sizeof(double) * H * H));
    if (gemm(dW, D, dW, D, dG, H, H, H, D, nullptr)) return 1;
    CK(hipDeviceSynchronize());

    pm_bsc_estep_params P;
    const double pi = 4.0 / H, sigma = 1.0;
    P.pil_bar = log(pi / (1 - pi));
    P.ecoef = -1.0 / (2 * sigma * sigma);
    P.prior_scale = 1.0;
    P.mu_sqnorm = 0.0;
    auto launch = [&]() {
        if (tile != 8)
            return fused4(dY, D, dW, D, dG, dyn, nullptr, nullptr, dmasks, dparents, size_off.data(), S, gamma, &P, N, D, H,
                          Hp, 3, dcand, dlogpj, Kpad, dlse, dexpect, H, dstats, D, nullptr);
        return fused(dY, D, dW, D, dG, dyn, nullptr, nullptr, dmasks, dparents, size_off.data(), S, gamma, &P, N, D, H, Hp,
                     3, dcand, dlogpj, Kpad, dlse, dexpect, H, dstats, D, part, nullptr);
    };
    for (int i = 0; i < 10; ++i)
        if (int e = launch()) {
            fprintf(stderr, "launch failed: %d\n", e);
            return 1;
        }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms(reps);
    // back-to-back block first (sustained clock), then individually timed launches
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float tot;
    CK(hipEventElapsedTime(&tot, e0, e1));
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    const double flop = 2.0 * N * D * H;
    printf("%s tile %d N %lld: back-to-back %.4f ms/launch (%.1f TF/s, %.3f of 78.6) | single median %.4f min %.4f\n", libpath,
           tile, (long long)N, tot / reps, flop / (tot / reps) * 1e-9, flop / (tot / reps) * 1e-9 / 78.6, ms[reps / 2], ms[0]);
    // checksum so that nothing is optimised away / to eyeball variants against each other
    std::vector<double> hl(1024);
    CK(hipMemcpy(hl.data(), dlse, sizeof(double) * 1024, hipMemcpyDeviceToHost));
    double cs = 0;
    for (double v : hl) cs += v;
    printf("  lse checksum %.10e\n", cs);

    if (want_stamps) {
        stamps_fn rd = (stamps_fn)dlsym(lib, tile == 8 ? "pm_f8_read_stamps" : "pm_fused_read_stamps");
        if (!rd) {
            fprintf(stderr, "no stamps in this build\n");
            return 0;
        }
        const int nb = (int)std::min<int64_t>(8192, (N + 63) / 64);
        std::vector<unsigned long long> st((size_t)nb * 8);
        if (rd(st.data(), nb)) return 1;
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb; ++b) t0 = std::min(t0, st[(size_t)b * 8]);
        double sum[8] = {0}, span = 0;
        double clk = 0;
        int nclk = 0;
        for (int b = 0; b < nb; ++b) {
            const unsigned long long *s = &st[(size_t)b * 8];
            for (int i = 1; i < 5; ++i) sum[i] += (double)(s[i] - s[i - 1]) / 100.0;
            span = std::max(span, (double)(s[4] - t0) / 100.0);
            if (s[6] && s[1] > s[0]) {
                clk += (double)s[6] / ((double)(s[1] - s[0]) * 10.0);   // cycles per ns -> GHz
                ++nclk;
            }
        }
        printf("  stamps (mean us per workgroup): K-loop %.1f | tables %.1f | row passes %.1f | tail %.1f ; span %.1f us ; clock %.3f GHz\n",
               sum[1] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb, span, nclk ? clk / nclk : 0.0);
        typedef int (*es_fn)(unsigned long long *);
        es_fn res = (es_fn)dlsym(lib, "pm_f8_read_estamps");
        if (res) {
            unsigned long long es[16][32];
            if (!res(&es[0][0])) {
                printf("  row-pass phases of workgroup 300 (us since its pass 0 started; slots: start, B1, select, B2, fetch+singles, energies, lse):\n");
                for (int w = 0; w < 16; ++w) {
                    printf("   wave %d:", w);
                    for (int r = 0; r < 4; ++r) {
                        for (int k = 0; k < 7; ++k) printf(" %5.1f", (double)((long long)es[w][r * 8 + k] - (long long)es[0][0]) / 100.0);
                        printf(" |");
                    }
                    printf("\n");
                }
            }
        }
        // timeline of the first CU's workgroups (same HW_ID)
        const unsigned long long id0 = st[7];
        int shown = 0;
        for (int b = 0; b < nb && shown < 14; ++b) {
            const unsigned long long *s = &st[(size_t)b * 8];
            if (s[7] != id0) continue;
            printf("  wg %5d start %8.1f kend %8.1f tables %8.1f end %8.1f\n", b, (double)(s[0] - t0) / 100.0,
                   (double)(s[1] - t0) / 100.0, (double)(s[2] - t0) / 100.0, (double)(s[4] - t0) / 100.0);
            ++shown;
        }
    }
    return 0;
}
