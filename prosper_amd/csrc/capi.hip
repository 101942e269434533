// ABI bookkeeping for libprosper_hip.so (see include/prosper_hip.h).
#include <hip/hip_runtime.h>

#include "prosper_hip.h"

extern "C" int pm_version(void) { return 1014; }  // 1014 pm_sort_row_list_i32 takes a flag array (two launches, no LDS atomics); 1013 GSC's deterministic quanta from device-side parameters (pm_gsc_det_quanta_f64), pm_sort_row_list_i32; 1012 deferred statistics of data-truncation steps (pm_bsc_estep_fused8_defer_f64, pm_bsc_defer_apply_f64); 1011 the list-writing BSC pass keeps only overflowed datapoints' dense rows (pm_bsc_wp_sparse_expand_f64); 1007 fused scores GEMM + E-step (pm_bsc_estep_fused_f64); 1001 spd inverse, 1002 MMCA (pm_mca_params.signed_w), 1003 DSC, 1004 fused MCA pass + column moments, 1005 TSC flags in pm_dsc_params, weighted row norms, 1006 batched spd inverse, per-XCD scratch in the MCA / GSC statistics

extern "C" const char *pm_error_string(int code) {
    if (code == PM_OK) return "ok";
    if (code == PM_EINVAL) return "invalid argument (null pointer, non-positive dimension or short leading dimension)";
    if (code == PM_ERANGE) return "dimension outside the supported range";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown error";
}

// ---- the deterministic build (pm_common.h, PM_DETERMINISTIC) --------------------------------------------------------------
extern "C" int prosper_det_set_bsc_fused8(const double *, void *);
extern "C" int prosper_det_set_bsc_wp_sparse(const double *, void *);
extern "C" int prosper_det_set_gsc(const double *, void *);
extern "C" int prosper_det_set_gemm(const double *, void *);
extern "C" int prosper_det_set_mca(const double *, void *);
extern "C" int prosper_det_set_dsc(const double *, void *);
extern "C" int prosper_det_set_bsc_rows16(const double *, void *);
extern "C" int prosper_det_set_bsc_fused(const double *, void *);
extern "C" int prosper_det_set_bsc_kernels(const double *, void *);

extern "C" int pm_det_build(void) {
#ifdef PM_DETERMINISTIC
    return 1;
#else
    return 0;
#endif
}

extern "C" int pm_det_set_quanta(int unit, const double *M8, void *stream) {
    if (!M8) return PM_EINVAL;
    switch (unit) {
        case PM_DET_BSC_FUSED8: return prosper_det_set_bsc_fused8(M8, stream);
        case PM_DET_WP_SPARSE: return prosper_det_set_bsc_wp_sparse(M8, stream);
        case PM_DET_GSC: return prosper_det_set_gsc(M8, stream);
        case PM_DET_GEMM: return prosper_det_set_gemm(M8, stream);
        case PM_DET_MCA: return prosper_det_set_mca(M8, stream);
        case PM_DET_DSC: return prosper_det_set_dsc(M8, stream);
        case PM_DET_BSC_ROWS16: return prosper_det_set_bsc_rows16(M8, stream);
        case PM_DET_BSC_FUSED: return prosper_det_set_bsc_fused(M8, stream);
        case PM_DET_BSC_KERNELS: return prosper_det_set_bsc_kernels(M8, stream);
        default: return PM_EINVAL;
    }
}
