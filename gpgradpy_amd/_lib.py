"""ctypes binding of libgpgrad_hip.so (the C ABI declared in include/gpgrad.h).

The HIP library is the product: if it is missing or cannot be loaded this module raises -- there is no
CPU fallback anywhere in this package.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C gpgradpy_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpgrad_hip.so")
if os.environ.get("GPG_LIB"):   # A/B measurements only (tools/build_variant.sh): another build of the SAME HIP library
    LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ["GPG_LIB"]))

GPG_KERNEL = {"SqExp": 0, "Ma5f2": 1, "RatQu": 2}
GPG_WELLCOND = {"base": 0, "precon": 1}
PROF_CATS = ("assembly", "potrf", "trsm", "gemm_panel", "gemm_trail", "reduce")
PROF_NCAT = len(PROF_CATS)

# every symbol include/gpgrad.h declares
ABI_SYMBOLS = (
    "gpg_create", "gpg_destroy", "gpg_last_error", "gpg_set_grad_mask", "gpg_set_data", "gpg_lkd", "gpg_lkd_grad", "gpg_lkd_batch", "gpg_lkd_grad_batch",
    "gpg_setup_eval", "gpg_predict", "gpg_predict_grad", "gpg_predict_var", "gpg_predict_hess", "gpg_get_matrix", "gpg_kern_rtensor", "gpg_kern_rtensor_grad_hp", "gpg_kern_rtensor_hess_x", "gpg_factor_apply", "gpg_dcov_quadform", "gpg_cond_fro", "gpg_set_noise", "gpg_abs_rowsum", "gpg_set_gradient_nugget", "gpg_lkd_alpha", "gpg_prof_enable", "gpg_prof_read", "gpg_set_panel", "gpg_set_lookahead", "gpg_set_factor_mode", "gpg_set_pair_mode", "gpg_factor_fallbacks", "gpg_overlap_fallbacks", "gpg_solve_fallbacks", "gpg_last_factor", "gpg_set_batch", "gpg_reserve_batch", "gpg_set_max_workgroups",
    "gpg_device_info", "gpg_multi_create", "gpg_multi_destroy", "gpg_multi_last_error", "gpg_multi_count", "gpg_multi_set_data",
    "gpg_multi_lkd_batch",
)


class GpgHp(C.Structure):
    _fields_ = [("theta", C.POINTER(C.c_double)), ("varK_mat", C.c_double), ("var_fval", C.c_double),
                ("var_fgrad", C.c_double), ("eta", C.c_double), ("wellcond", C.c_int),
                ("closed_form_varK", C.c_int), ("hp_kernel", C.c_double)]


class GpgLkdOut(C.Structure):
    _fields_ = [("ln_lkd", C.c_double), ("ln_det", C.c_double), ("beta", C.c_double), ("varK", C.c_double),
                ("rKr", C.c_double), ("info", C.c_int), ("pad_", C.c_int)]


class GpgError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpgError(f"{LIB_PATH} not found: build the HIP extension first "
                       "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    vp = C.c_void_p
    lib.gpg_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.gpg_create.restype = C.c_int
    lib.gpg_destroy.argtypes = [vp]
    lib.gpg_destroy.restype = None
    lib.gpg_last_error.argtypes = [vp]
    lib.gpg_last_error.restype = C.c_char_p
    lib.gpg_set_grad_mask.argtypes = [vp, C.POINTER(C.c_ubyte)]
    lib.gpg_set_grad_mask.restype = C.c_int
    lib.gpg_set_data.argtypes = [vp, dp, dp, dp]
    lib.gpg_set_data.restype = C.c_int
    lib.gpg_lkd.argtypes = [vp, C.POINTER(GpgHp), C.POINTER(GpgLkdOut)]
    lib.gpg_lkd.restype = C.c_int
    lib.gpg_lkd_grad.argtypes = [vp, C.POINTER(GpgHp), C.POINTER(GpgLkdOut), dp, dp]
    lib.gpg_lkd_grad.restype = C.c_int
    lib.gpg_lkd_batch.argtypes = [vp, C.c_int, dp, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(GpgLkdOut)]
    lib.gpg_lkd_batch.restype = C.c_int
    lib.gpg_lkd_grad_batch.argtypes = [vp, C.c_int, dp, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(GpgLkdOut), dp, dp]
    lib.gpg_lkd_grad_batch.restype = C.c_int
    lib.gpg_setup_eval.argtypes = [vp, C.POINTER(GpgHp), C.c_double, dp]
    lib.gpg_setup_eval.restype = C.c_int
    lib.gpg_predict.argtypes = [vp, C.c_int, dp, C.c_double, dp, dp, dp]
    lib.gpg_predict.restype = C.c_int
    lib.gpg_predict_grad.argtypes = [vp, C.c_int, dp, C.c_double, dp, dp, dp, dp, dp]
    lib.gpg_predict_grad.restype = C.c_int
    lib.gpg_predict_var.argtypes = [vp, C.c_int, dp, C.c_double, dp, dp]
    lib.gpg_predict_var.restype = C.c_int
    lib.gpg_predict_hess.argtypes = [vp, dp, C.c_double, dp, dp, dp, dp, dp, dp]
    lib.gpg_predict_hess.restype = C.c_int
    lib.gpg_get_matrix.argtypes = [vp, C.POINTER(GpgHp), C.c_int, dp]
    lib.gpg_get_matrix.restype = C.c_int
    lib.gpg_kern_rtensor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, C.c_int,
                                     C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte), dp]
    lib.gpg_kern_rtensor.restype = C.c_int
    lib.gpg_prof_enable.argtypes = [vp, C.c_uint]
    lib.gpg_prof_enable.restype = C.c_int
    lib.gpg_prof_read.argtypes = [vp, dp, C.POINTER(C.c_longlong), dp]
    lib.gpg_prof_read.restype = C.c_int
    lib.gpg_set_panel.argtypes = [vp, C.c_int]
    lib.gpg_set_panel.restype = C.c_int
    lib.gpg_set_lookahead.argtypes = [vp, C.c_int]
    lib.gpg_set_lookahead.restype = C.c_int
    lib.gpg_set_factor_mode.argtypes = [vp, C.c_int]
    lib.gpg_set_factor_mode.restype = C.c_int
    lib.gpg_set_batch.argtypes = [vp, C.c_int]
    lib.gpg_set_batch.restype = C.c_int
    lib.gpg_set_max_workgroups.argtypes = [vp, C.c_int]
    lib.gpg_set_max_workgroups.restype = C.c_int
    lib.gpg_reserve_batch.argtypes = [vp, C.c_int]
    lib.gpg_reserve_batch.restype = C.c_int
    lib.gpg_factor_apply.argtypes = [vp, C.c_int, dp, dp]
    lib.gpg_factor_apply.restype = C.c_int
    lib.gpg_dcov_quadform.argtypes = [vp, C.POINTER(GpgHp), dp, dp]
    lib.gpg_dcov_quadform.restype = C.c_int
    lib.gpg_cond_fro.argtypes = [vp, C.POINTER(GpgHp), C.POINTER(C.c_double), dp]
    lib.gpg_abs_rowsum.argtypes = [vp, C.POINTER(GpgHp), dp]
    lib.gpg_abs_rowsum.restype = C.c_int
    lib.gpg_kern_rtensor_grad_hp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, C.c_int, dp, dp]
    lib.gpg_kern_rtensor_grad_hp.restype = C.c_int
    lib.gpg_kern_rtensor_hess_x.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, C.c_int, C.POINTER(C.c_ubyte), dp]
    lib.gpg_kern_rtensor_hess_x.restype = C.c_int
    lib.gpg_set_noise.argtypes = [vp, dp]
    lib.gpg_set_noise.restype = C.c_int
    lib.gpg_lkd_alpha.argtypes = [vp, dp]
    lib.gpg_lkd_alpha.restype = C.c_int
    lib.gpg_set_gradient_nugget.argtypes = [vp, C.c_double]
    lib.gpg_set_gradient_nugget.restype = C.c_int
    lib.gpg_cond_fro.restype = C.c_int
    lib.gpg_factor_fallbacks.argtypes = [vp]
    lib.gpg_factor_fallbacks.restype = C.c_int
    lib.gpg_set_pair_mode.argtypes = [vp, C.c_int]
    lib.gpg_set_pair_mode.restype = C.c_int
    for name in ("gpg_overlap_fallbacks", "gpg_solve_fallbacks"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = C.c_int
    lib.gpg_last_factor.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.gpg_last_factor.restype = C.c_int
    lib.gpg_multi_create.argtypes = [C.POINTER(vp), C.c_int, ip, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.gpg_multi_create.restype = C.c_int
    lib.gpg_multi_destroy.argtypes = [vp]
    lib.gpg_multi_destroy.restype = None
    lib.gpg_multi_last_error.argtypes = [vp]
    lib.gpg_multi_last_error.restype = C.c_char_p
    lib.gpg_multi_count.argtypes = [vp]
    lib.gpg_multi_count.restype = C.c_int
    lib.gpg_multi_set_data.argtypes = [vp, dp, dp, dp]
    lib.gpg_multi_set_data.restype = C.c_int
    lib.gpg_multi_lkd_batch.argtypes = [vp, C.c_int, dp, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(GpgLkdOut), ip]
    lib.gpg_multi_lkd_batch.restype = C.c_int
    lib.gpg_device_info.argtypes = [C.c_int, C.c_char_p, C.c_int]
    lib.gpg_device_info.restype = C.c_int
    _lib = lib
    return lib


def as_dp(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_double))
