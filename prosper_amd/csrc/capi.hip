// ABI bookkeeping for libprosper_hip.so (see include/prosper_hip.h).
#include <hip/hip_runtime.h>

#include "prosper_hip.h"

extern "C" int pm_version(void) { return 1009; }  // 1007 fused scores GEMM + E-step (pm_bsc_estep_fused_f64); 1001 spd inverse, 1002 MMCA (pm_mca_params.signed_w), 1003 DSC, 1004 fused MCA pass + column moments, 1005 TSC flags in pm_dsc_params, weighted row norms, 1006 batched spd inverse, per-XCD scratch in the MCA / GSC statistics

extern "C" const char *pm_error_string(int code) {
    if (code == PM_OK) return "ok";
    if (code == PM_EINVAL) return "invalid argument (null pointer, non-positive dimension or short leading dimension)";
    if (code == PM_ERANGE) return "dimension outside the supported range";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown error";
}
