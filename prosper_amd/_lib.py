"""ctypes binding of libprosper_hip.so (C ABI: include/prosper_hip.h).

There is no CPU fallback: if the library is missing or a call fails this raises, so a
GPU box can never silently run something else than the HIP kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprosper_hip.so")

c_dp = C.c_void_p      # device pointers travel as integers (torch.Tensor.data_ptr())
i64 = C.c_int64


class McaParams(C.Structure):
    """struct pm_mca_params"""
    _fields_ = [("pil_bar", C.c_double), ("pre1", C.c_double), ("beta", C.c_double), ("inv_rho", C.c_double),
                ("signed_w", C.c_double)]


class DscParams(C.Structure):
    """struct pm_dsc_params"""
    _fields_ = [("K", C.c_int32), ("K0", C.c_int32), ("values", C.c_double * 8), ("logpi", C.c_double * 8),
                ("pre1", C.c_double), ("ecoef", C.c_double), ("pscale", C.c_double),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


class EStepParams(C.Structure):
    """struct pm_bsc_estep_params"""
    _fields_ = [("pil_bar", C.c_double), ("ecoef", C.c_double),
                ("prior_scale", C.c_double), ("mu_sqnorm", C.c_double)]


# name -> (restype, argtypes); every symbol include/prosper_hip.h declares
SIGNATURES = {
    "pm_version": (C.c_int, []),
    "pm_error_string": (C.c_char_p, [C.c_int]),
    "pm_gemm_nt_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, i64, c_dp]),
    "pm_gemm_nn_small_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, i64, c_dp]),
    "pm_gemm_nt_small_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, i64, c_dp]),
    "pm_gemm_tn_acc_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, i64, c_dp]),
    "pm_row_sqnorm_f64": (C.c_int, [c_dp, i64, i64, i64, c_dp, c_dp]),
    "pm_col_moments_f64": (C.c_int, [c_dp, i64, i64, i64, c_dp, c_dp, c_dp]),
    "pm_row_wsqnorm_f64": (C.c_int, [c_dp, i64, i64, i64, c_dp, c_dp, c_dp]),
    "pm_spd_inverse_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, c_dp, i64, c_dp, c_dp]),
    "pm_spd_inverse_warm_work_len": (i64, [i64]),
    "pm_spd_inverse_warm_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, c_dp, c_dp, c_dp, i64, c_dp, c_dp]),
    "pm_spd_inverse_warm_long_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, c_dp, c_dp, c_dp, i64, c_dp, c_dp]),
    "pm_spd_inverse_warm_batch_f64": (C.c_int, [c_dp, i64, i64, c_dp, i64, c_dp, i64, c_dp, c_dp, i64, c_dp, i64, c_dp]),
    "pm_inverse_warm_batch_f64": (C.c_int, [c_dp, i64, i64, c_dp, i64, c_dp, i64, c_dp, c_dp, i64, c_dp, c_dp, i64, C.c_uint32,
                                            c_dp]),
    "pm_spd_inverse_batch_f64": (C.c_int, [c_dp, i64, i64, c_dp, i64, c_dp, c_dp, i64, i64, c_dp, i64, c_dp]),
    "pm_kth_hist_f64": (C.c_int, [c_dp, i64, c_dp, C.c_int, C.c_int, c_dp, c_dp]),
    "pm_kth_scan": (C.c_int, [c_dp, c_dp, C.c_int, C.c_int, c_dp]),
    "pm_kth_value_f64": (C.c_int, [c_dp, c_dp, c_dp]),
    "pm_kth_round_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_dp]),
    "pm_kth_final_f64": (C.c_int, [c_dp, c_dp, C.c_int, C.c_int, C.c_int, c_dp, c_dp]),
    "pm_kth_round_k_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i64, c_dp]),
    "pm_kth_final_z_f64": (C.c_int, [c_dp, c_dp, C.c_int, C.c_int, C.c_int, c_dp, C.c_int, c_dp]),
    "pm_bsc_select_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, c_dp, c_dp]),
    "pm_bsc_estep_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, i64,
                                   C.POINTER(EStepParams), i64, i64, i64, c_dp, i64, c_dp, c_dp]),
    "pm_bsc_stats_len": (i64, [i64, i64]),
    "pm_bsc_stats_offset_wq": (i64, [i64, i64]),
    "pm_bsc_stats_offset_qdiag": (i64, [i64, i64]),
    "pm_bsc_stats_offset_mus": (i64, [i64, i64]),
    "pm_bsc_stats_offset_scalars": (i64, [i64, i64]),
    "pm_bsc_mstep_rows_f64": (C.c_int, [c_dp, i64, c_dp, C.c_double, c_dp, c_dp, i64, c_dp, c_dp, i64,
                                        C.POINTER(EStepParams), i64, i64, i64, i64, c_dp, i64, c_dp, c_dp]),
    "pm_bsc_rows16_supported": (C.c_int, [i64, i64, i64]),
    "pm_bsc_rows16_nz_supported": (C.c_int, [i64, i64, i64]),
    "pm_bsc_select_estep_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                          C.POINTER(C.c_int32), i64, i64, C.POINTER(EStepParams), i64, i64, i64,
                                          C.c_int, c_dp, c_dp, i64, c_dp, c_dp]),
    "pm_bsc_fused_supported": (C.c_int, [i64, i64, i64, i64]),
    "pm_bsc_fused_occupancy": (C.c_int, [i64, i64, i64, i64]),
    "pm_bsc_estep_fused_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                         C.POINTER(C.c_int32), i64, i64, C.POINTER(EStepParams), i64, i64, i64, i64,
                                         C.c_int, c_dp, c_dp, i64, c_dp, c_dp, i64, c_dp, i64, c_dp]),
    "pm_bsc_fused8_supported": (C.c_int, [i64, i64, i64, i64]),
    "pm_bsc_fused8_whole_shard": (C.c_int, [i64, i64, i64, i64]),
    "pm_bsc_fused8_main_rows": (i64, [i64, i64]),
    "pm_bsc_estep_fused8_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                          C.POINTER(C.c_int32), i64, i64, C.POINTER(EStepParams), i64, i64, i64, i64,
                                          C.c_int, c_dp, c_dp, i64, c_dp, c_dp, i64, c_dp, i64, C.c_int, c_dp]),
    "pm_bsc_estep_fused8_nz_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                             C.POINTER(C.c_int32), i64, i64, C.POINTER(EStepParams), i64, i64, i64, i64,
                                             C.c_int, c_dp, c_dp, i64, c_dp, c_dp, i64, c_dp, i64, c_dp, c_dp, C.c_int,
                                             c_dp]),
    "pm_bsc_estep_fused8_defer_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                                C.POINTER(C.c_int32), i64, i64, C.POINTER(EStepParams), i64, i64, i64, i64,
                                                C.c_int, c_dp, c_dp, i64, c_dp, c_dp, i64, c_dp, i64, c_dp, c_dp, c_dp,
                                                C.c_int, c_dp]),
    "pm_bsc_defer_apply_f64": (C.c_int, [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, i64, c_dp, C.POINTER(EStepParams),
                                         i64, i64, i64, i64, c_dp]),
    "pm_bsc_wp_sparse_f64": (C.c_int, [c_dp, c_dp, c_dp, i64, c_dp, i64, i64, i64, c_dp]),
    "pm_bsc_expand_lists_gated_f64": (C.c_int, [c_dp, c_dp, c_dp, i64, i64, i64, c_dp, c_dp]),
    "pm_bsc_wp_sparse_expand_f64": (C.c_int, [c_dp, c_dp, c_dp, i64, c_dp, c_dp, i64, i64, i64, i64, c_dp]),
    "pm_gemm_tn_acc_gated_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, i64, c_dp, c_dp]),
    "pm_gemm_tn_acc_rows_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, c_dp, c_dp, i64, i64, c_dp]),
    "pm_bsc_mstep_rows16_f64": (C.c_int, [c_dp, i64, c_dp, C.c_double, c_dp, c_dp, i64,
                                          C.POINTER(EStepParams), i64, i64, i64, i64, c_dp, i64, c_dp, c_dp]),
    "pm_bsc_mstep_rows16_nz_f64": (C.c_int, [c_dp, i64, c_dp, C.c_double, c_dp, c_dp, i64,
                                             C.POINTER(EStepParams), i64, i64, i64, i64, c_dp, i64, c_dp, c_dp, c_dp,
                                             c_dp]),
    "pm_mca_select_scores_f64": (C.c_int, [c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, i64, c_dp]),
    "pm_mca_estep_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, i64, c_dp, c_dp, c_dp, i64,
                                   C.POINTER(McaParams), i64, i64, i64, i64, c_dp, i64, c_dp, c_dp, c_dp]),
    "pm_mca_stats_len": (i64, [i64, i64]),
    "pm_mca_mstep_rows_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, C.c_double, c_dp, i64, c_dp, c_dp, c_dp, c_dp, i64,
                                        C.POINTER(McaParams), i64, i64, i64, i64, c_dp, i64, c_dp, c_dp]),
    "pm_mca_w_update_f64": (C.c_int, [c_dp, c_dp, i64, i64, C.c_double, c_dp, c_dp, c_dp]),
    "pm_mca_tables_f64": (C.c_int, [c_dp, i64, i64, C.c_double, c_dp, c_dp, c_dp]),
    "pm_mca_estep_mstats_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, i64, c_dp, c_dp, c_dp, c_dp, i64,
                                          C.POINTER(McaParams), i64, i64, i64, i64, c_dp, i64, c_dp, c_dp,
                                          c_dp, i64, c_dp, c_dp]),
    "pm_mca_estep_mstats_defer_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, i64, c_dp, c_dp, c_dp, c_dp, i64,
                                                C.POINTER(McaParams), i64, i64, i64, i64, c_dp, i64, c_dp, c_dp,
                                                c_dp, i64, c_dp, c_dp, c_dp, c_dp]),
    "pm_mca_defer_apply_work_len": (i64, [i64, i64]),
    "pm_mca_defer_apply_f64": (C.c_int, [c_dp, c_dp, c_dp, i64, c_dp, c_dp, c_dp, c_dp, i64, c_dp, c_dp, i64, i64, i64, i64,
                                         c_dp]),
    "pm_xsc_select_supported": (C.c_int, [i64, i64, C.c_int]),
    "pm_xsc_select_f64": (C.c_int, [c_dp, i64, c_dp, C.POINTER(DscParams), i64, i64, i64, c_dp, c_dp]),
    "pm_dsc_select_scores_f64": (C.c_int, [c_dp, i64, c_dp, C.POINTER(DscParams), i64, i64, c_dp, i64, c_dp]),
    "pm_tsc_select_scores_f64": (C.c_int, [c_dp, i64, c_dp, i64, i64, c_dp, i64, c_dp]),
    "pm_dsc_estep_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, c_dp, i64, c_dp, C.POINTER(DscParams), i64, i64, i64,
                                   c_dp, i64, c_dp, c_dp]),
    "pm_dsc_stats_len": (i64, [i64, i64]),
    "pm_dsc_mstep_rows_f64": (C.c_int, [c_dp, i64, c_dp, C.c_double, c_dp, c_dp, i64, c_dp, C.POINTER(DscParams),
                                        i64, i64, i64, i64, c_dp, i64, c_dp, c_dp]),
    "pm_dsc_mstep_rows_nz_f64": (C.c_int, [c_dp, i64, c_dp, C.c_double, c_dp, c_dp, i64, c_dp, C.POINTER(DscParams),
                                           i64, i64, i64, i64, c_dp, i64, c_dp, c_dp, c_dp, c_dp]),
    "pm_dsc_mstep_rows_cutp_f64": (C.c_int, [c_dp, i64, c_dp, C.c_double, c_dp, c_dp, c_dp, i64, c_dp, C.POINTER(DscParams),
                                             i64, i64, i64, i64, c_dp, i64, c_dp, c_dp, c_dp, c_dp]),
    "pm_dsc_rows16_supported": (C.c_int, [i64, i64, i64, i64, C.c_int]),
    "pm_dsc_estep_mstats_supported": (C.c_int, [i64, i64, i64, i64, C.c_int]),
    "pm_dsc_estep_mstats_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, c_dp, i64, c_dp, C.POINTER(DscParams), i64, i64, i64,
                                          i64, c_dp, i64, c_dp, c_dp, i64, c_dp, c_dp, c_dp, c_dp]),
    "pm_wp_sparse_f64": (C.c_int, [c_dp, c_dp, c_dp, i64, c_dp, i64, c_dp, i64, i64, i64, c_dp]),
    "pm_wp_sparse_t_f64": (C.c_int, [c_dp, c_dp, c_dp, i64, c_dp, i64, i64, i64, i64, c_dp]),
    "pm_gsc_supported": (C.c_int, [i64, i64, i64]),
    "pm_gsc_stats_len": (i64, [i64]),
    "pm_gsc_pack_stats_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp]),
    "pm_gsc_mstep_finish_f64": (C.c_int, [c_dp] * 10 + [C.c_double, i64, i64, C.c_int, c_dp, c_dp, c_dp]),
    "pm_gsc_estep_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, i64, i64, C.c_double, C.c_double,
                                   i64, i64, i64, C.c_int, c_dp, c_dp, c_dp, i64, c_dp, c_dp]),
    "pm_gsc_lists_supported": (C.c_int, [i64, i64, i64, i64]),
    "pm_gsc_estep_lists_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, i64, i64, C.c_double, C.c_double,
                                         i64, i64, i64, C.c_int, c_dp, c_dp, c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "pm_gsc_list_pairs_f64": (C.c_int, [c_dp, c_dp, c_dp, i64, i64, c_dp, c_dp]),
    "pm_gsc_estep_lpj_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, i64, i64, C.c_double, C.c_double,
                                       i64, i64, i64, C.c_int, c_dp, c_dp, c_dp, i64, c_dp, c_dp, i64, c_dp]),
    "pm_gsc_estep_lpj_blocks_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, c_dp, c_dp, c_dp, i64, i64, C.c_double, C.c_double,
                                              i64, i64, i64, C.c_int, c_dp, c_dp, c_dp, i64, c_dp, c_dp, i64, c_dp, i64, c_dp]),
    "pm_col_sum_kept_f64": (C.c_int, [c_dp, i64, i64, i64, c_dp, C.c_double, c_dp, c_dp]),
    "pm_infer_topk_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, i64, i64, i64, i64, i64, c_dp, c_dp, c_dp, c_dp, i64, c_dp]),
    "pm_infer_topk_cols_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, i64, i64, i64, i64, i64, i64, c_dp, c_dp, c_dp, c_dp, i64,
                                         c_dp]),
    "pm_infer_topk_signed_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, i64, i64, i64, i64, i64, C.c_int, c_dp, c_dp, c_dp, c_dp,
                                           c_dp, c_dp, c_dp, c_dp]),
    "pm_det_build": (C.c_int, []),
    "pm_det_set_quanta": (C.c_int, [C.c_int, c_dp, c_dp]),
    "pm_gsc_det_quanta_f64": (C.c_int, [c_dp, C.c_int64, c_dp, c_dp, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double,
                                        c_dp, c_dp]),
    "pm_sort_row_list_i32": (C.c_int, [c_dp, c_dp, C.c_int64, c_dp, c_dp]),
    "pm_gsc_component_scores_f64": (C.c_int, [c_dp, i64, c_dp, c_dp, C.c_double, i64, i64, c_dp, i64, c_dp]),
}


class HipError(RuntimeError):
    pass


MIN_VERSION = 1014
_lib = None
_lib_det = None
LIB_PATH_DET = os.path.join(os.path.dirname(LIB_PATH), "libprosper_hip_det.so")
DET_UNITS = {"bsc_fused8": 0, "wp_sparse": 1, "gsc": 2, "gemm": 3, "mca": 4, "dsc": 5, "bsc_rows16": 6, "bsc_fused": 7,
             "bsc_kernels": 8}      # PM_DET_* of prosper_hip.h


def _open(path, what):
    # torch bundles its own libamdhip64.so.7; it must be the ONE HIP runtime in the process
    # (pointers and streams handed to the C ABI come from it).  Loading ours first would pull
    # /opt/rocm's copy of the same soname and leave torch and the kernels on different runtimes.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise HipError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or prosper_amd/csrc/build.sh (looked in %s)" % (what, path))
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    # (the statistics layouts and list conventions the host layer relies on changed under unchanged entry-point names more
    # than once: refuse a library older than the one this package was written against)
    if lib.pm_version() < MIN_VERSION:
        raise HipError("%s is version %d, this package needs >= %d: rebuild it (prosper_amd/csrc/build.sh)"
                       % (path, lib.pm_version(), MIN_VERSION))
    return lib


def load(det=False):
    """Load the shared library once and attach prototypes.  ``det``: the deterministic build (libprosper_hip_det.so: the same
    sources with order-independent reductions, include/prosper_hip.h) that ``model.deterministic = True`` runs on."""
    global _lib, _lib_det
    if det:
        if _lib_det is None:
            _lib_det = _open(LIB_PATH_DET, "libprosper_hip_det.so")
            if _lib_det.pm_det_build() != 1:
                raise HipError("%s is not a deterministic build" % LIB_PATH_DET)
        return _lib_det
    if _lib is None:
        _lib = _open(LIB_PATH, "libprosper_hip.so")
    return _lib


def check(code, what=""):
    if code != 0:
        msg = load().pm_error_string(code).decode()
        raise HipError("%s failed: %s (code %d)" % (what or "libprosper_hip call", msg, code))


def call(name, *args, det=False):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    check(getattr(load(det), name)(*args), name)
