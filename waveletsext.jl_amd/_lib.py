"""ctypes binding of libwaveletsext_hip.so (the C ABI in include/waveletsext_hip.h).

The product path has NO CPU fallback: if the shared library is missing this module raises on
import of the symbol table, and every compute entry point fails with WxError(WX_EHIP) when no
HIP device is visible.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WX_HIP_LIB: another build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("WX_HIP_LIB") or os.path.join(_HERE, "csrc", "libwaveletsext_hip.so")

WX_OK, WX_EASSERT, WX_EARG, WX_EBOUNDS, WX_EHIP, WX_EUNSUPPORTED = 0, -1, -2, -3, -10, -11


class WxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libwaveletsext_hip status %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load the library once (RTLD_GLOBAL is not needed; HIP runtime comes in as a dependency)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libwaveletsext_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C waveletsext.jl_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


_P, _I, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64


def _declare(L):
    L.wx_version.restype = _I
    L.wx_last_error.restype = ctypes.c_char_p
    L.wx_device_count.restype = _I
    L.wx_build_info.restype = ctypes.c_char_p
    L.wx_debug_set_dispatch.argtypes = [_I]           # csrc/wx_debug.h: test-suite hook, not in the public header
    L.wx_debug_set_dispatch.restype = None
    sigs = {
        # name: argtypes (without the _f64/_f32 suffix)
        "wx_wpd1d": [_P, _P, _L, _I, _L, _P, _I, _P],
        "wx_wpt1d": [_P, _P, _L, _I, _P, _L, _L, _P, _I, _P],
        "wx_iwpt1d": [_P, _P, _L, _I, _P, _L, _L, _P, _I, _P],
        "wx_iwpd1d": [_P, _P, _L, _I, _I, _P, _L, _L, _P, _I, _P],
        "wx_getbasiscoef1d": [_P, _P, _L, _I, _P, _L, _L, _P],
    }
    sigs.update(_EXTRA_SIGS)
    for name, args in sigs.items():
        for suf in ("_f64", "_f32"):
            if hasattr(L, name + suf):
                fn = getattr(L, name + suf)
                fn.argtypes = args
                fn.restype = _I
    for name, args in _PLAIN_SIGS.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _I
    for name in ("wx_siwt_ncols", "wx_siwt_nnodes"):
        if hasattr(L, name):
            getattr(L, name).argtypes = [_I, _I]
            getattr(L, name).restype = _L


# filled in by the other host modules' families (SWT / ACWT / 2-D / JBB)
_EXTRA_SIGS = {
    "wx_wpd2d": [_P, _P, _L, _L, _I, _L, _P, _I, _P],
    "wx_wpt2d": [_P, _P, _L, _L, _I, _P, _L, _L, _P, _I, _P],
    "wx_iwpt2d": [_P, _P, _L, _L, _I, _P, _L, _L, _P, _I, _P],
    "wx_iwpd2d": [_P, _P, _L, _L, _I, _I, _P, _L, _L, _P, _I, _P],
    "wx_sdwt1d": [_P, _P, _L, _I, _L, _P, _I, _P],
    "wx_isdwt1d": [_P, _P, _L, _I, _L, _L, _P, _I, _P],
    "wx_swpt1d": [_P, _P, _L, _I, _L, _P, _I, _P],
    "wx_iswpt1d": [_P, _P, _L, _I, _L, _L, _P, _I, _P],
    "wx_swpd1d": [_P, _P, _L, _I, _L, _P, _I, _P],
    "wx_iswpd1d": [_P, _P, _L, _L, _I, _P, _L, _L, _L, _P, _I, _P],
    "wx_acdwt1d": [_P, _P, _L, _I, _L, _P, _I, _P],
    "wx_iacdwt1d": [_P, _P, _L, _I, _L, _P],
    "wx_acwpt1d": [_P, _P, _L, _I, _L, _P, _I, _P],
    "wx_iacwpt1d": [_P, _P, _L, _I, _L, _P],
    "wx_acwpd1d": [_P, _P, _L, _I, _L, _P, _I, _P],
    "wx_iacwpd1d": [_P, _P, _L, _L, _I, _P, _L, _L, _P],
    "wx_sdwt2d": [_P, _P, _L, _L, _I, _L, _P, _I, _P],
    "wx_swpt2d": [_P, _P, _L, _L, _I, _L, _P, _I, _P],
    "wx_swpd2d": [_P, _P, _L, _L, _I, _L, _P, _I, _P],
    "wx_isdwt2d": [_P, _P, _L, _L, _I, _L, _L, _P, _I, _P],
    "wx_iswpt2d": [_P, _P, _L, _L, _I, _L, _L, _P, _I, _P],
    "wx_iswpd2d": [_P, _P, _L, _L, _L, _I, _P, _L, _L, _L, _P, _I, _P],
    "wx_acdwt2d": [_P, _P, _L, _L, _I, _L, _P, _I, _P],
    "wx_acwpt2d": [_P, _P, _L, _L, _I, _L, _P, _I, _P],
    "wx_acwpd2d": [_P, _P, _L, _L, _I, _L, _P, _I, _P],
    "wx_iacdwt2d": [_P, _P, _L, _L, _I, _L, _P],
    "wx_iacwpt2d": [_P, _P, _L, _L, _I, _L, _P],
    "wx_iacwpd2d": [_P, _P, _L, _L, _L, _I, _P, _L, _L, _P],
    "wx_jbb_costs2d": [_P, _P, _L, _L, _L, _L, _I, _I, ctypes.c_double, _P, _P],
    "wx_getbasiscoef2d": [_P, _P, _L, _L, _I, _P, _L, _L, _P],
    "wx_getbasiscoef1d_trees": [_P, _P, _L, _I, _P, _L, _L, _P],
    "wx_getbasiscoef2d_trees": [_P, _P, _L, _L, _I, _P, _L, _L, _P],
    "wx_jbb_moments": [_P, _P, _P, _L, _L, _I, _P],
    "wx_jbb_costs": [_P, _P, _L, _L, _L, _I, _I, ctypes.c_double, _P, _P],
    "wx_acwpd_jbb_moments": [_P, _P, _P, _L, _I, _L, _P, _I, _I, _P],
    "wx_siwpd": [_P, _P, _P, _L, _I, _I, _L, _P, _I, _P],
    "wx_siwt_bestbasis": [_P, _P, _I, _I, _L, _P],
    "wx_isiwpd": [_P, _P, _P, _L, _I, _I, _L, _P, _I, _I, _P],
}
_PLAIN_SIGS = {
    "wx_treeselect_f64": [_P, _L, _L, _I, _P],
    "wx_treeselect_f32": [_P, _L, _L, _I, _P],
    "wx_treeselect_gap_f64": [_P, _L, _L, _I, _P, _P],
    "wx_treeselect_gap_f32": [_P, _L, _L, _I, _P, _P],
    "wx_treeselect2d_f64": [_P, _L, _L, _L, _I, _P],
    "wx_treeselect2d_f32": [_P, _L, _L, _L, _I, _P],
    "wx_shutdown": [],
    "wx_set_host_hugepages": [_I],
    "wx_energy_map_f64": [_P, _L, _L, _L, _P, _I, _P, _P, _P],
    "wx_energy_map_f32": [_P, _L, _L, _L, _P, _I, _P, _P, _P],
    "wx_class_mean_f64": [_P, _L, _L, _P, _I, _P, _P],
    "wx_class_mean_f32": [_P, _L, _L, _P, _I, _P, _P],
    "wx_class_var_f64": [_P, _L, _L, _P, _I, _P, _P, _P],
    "wx_class_var_f32": [_P, _L, _L, _P, _I, _P, _P, _P],
    "wx_class_median_mad_f64": [_P, _L, _L, _P, _I, _P, _P, _P],
    "wx_class_median_mad_f32": [_P, _L, _L, _P, _I, _P, _P, _P],
    "wx_emd_measure_f64": [_P, _L, _L, _P, _I, _P, _P],
    "wx_emd_measure_f32": [_P, _L, _L, _P, _I, _P, _P],
    "wx_pdf_energy_map_f64": [_P, _L, _L, _P, _I, _P, _P],
    "wx_pdf_energy_map_f32": [_P, _L, _L, _P, _I, _P, _P],
    "wx_signature_weights_f64": [_P, _L, _L, _P, _I, _P, _P],
    "wx_signature_weights_f32": [_P, _L, _L, _P, _I, _P, _P],
    "wx_emd_measure_weighted_f64": [_P, _P, _L, _L, _P, _I, _P, _P],
    "wx_emd_measure_weighted_f32": [_P, _P, _L, _L, _P, _I, _P, _P],
    "wx_noisest_f64": [_P, _L, _L, _L, _L, _L, _P, _P],
    "wx_noisest_f32": [_P, _L, _L, _L, _L, _L, _P, _P],
    "wx_threshold_f64": [_P, _P, _L, _L, _L, _I, _P, _L, _L, _P, _P],
    "wx_threshold_f32": [_P, _P, _L, _L, _L, _I, _P, _L, _L, _P, _P],
    "wx_dwt3d_f64": [_P, _P, _L, _L, _L, _I, _L, _P, _I, _P],
    "wx_dwt3d_f32": [_P, _P, _L, _L, _L, _I, _L, _P, _I, _P],
    "wx_idwt3d_f64": [_P, _P, _L, _L, _L, _I, _L, _P, _I, _P],
    "wx_idwt3d_f32": [_P, _P, _L, _L, _L, _I, _L, _P, _I, _P],
    "wx_denoiseall_dwt_f64": [_P, _P, _L, _I, _L, _P, _I, _I, ctypes.c_double, _I, _P, _P],
    "wx_denoiseall_dwt_f32": [_P, _P, _L, _I, _L, _P, _I, _I, ctypes.c_double, _I, _P, _P],
    "wx_denoiseall_sig_f64": [_P, _P, _L, _I, _L, _P, _I, _I, ctypes.c_double, _I, _P, _P],
    "wx_denoiseall_sig_f32": [_P, _P, _L, _I, _L, _P, _I, _I, ctypes.c_double, _I, _P, _P],
    "wx_iwpt1d_thresh_f64": [_P, _P, _L, _I, _P, _L, _L, _P, _I, _I, _P, _L, _L, ctypes.c_double, _P],
    "wx_iwpt1d_thresh_f32": [_P, _P, _L, _I, _P, _L, _L, _P, _I, _I, _P, _L, _L, ctypes.c_double, _P],
    "wx_surethreshold_f64": [_P, _L, _L, _L, _P, _P, _P],
    "wx_surethreshold_f32": [_P, _L, _L, _L, _P, _P, _P],
    "wx_relerrorthreshold_f64": [_P, _L, _L, _L, _P, _I, _P, _P],
    "wx_relerrorthreshold_f32": [_P, _L, _L, _L, _P, _I, _P, _P],
    "wx_bb_costs_f64": [_P, _P, _L, _L, _L, _I, _I, _P],
    "wx_bb_costs_f32": [_P, _P, _L, _L, _L, _I, _I, _P],
    "wx_bb_costs2d_f64": [_P, _P, _L, _L, _L, _L, _I, _I, _P],
    "wx_bb_costs2d_f32": [_P, _P, _L, _L, _L, _L, _I, _I, _P],
    "wx_treeselect_batch_f64": [_P, _L, _L, _L, _I, _L, _P, _P],
    "wx_treeselect_batch_f32": [_P, _L, _L, _L, _I, _L, _P, _P],
    "wx_comm_unique_id": [_P],
    "wx_comm_init": [_I, _I, _P, ctypes.POINTER(ctypes.c_void_p)],
    "wx_comm_destroy": [_P],
    "wx_allgather_out_f64": [_P, _P, _L, _P, _P],
    "wx_allgather_out_f32": [_P, _P, _L, _P, _P],
    "wx_allgatherv_out_f64": [_P, _P, _P, _I, _P, _P],
    "wx_allgatherv_out_f32": [_P, _P, _P, _I, _P, _P],
    "wx_allreduce_moments_f64": [_P, _L, _P, _P],
    "wx_allreduce_moments_f32": [_P, _L, _P, _P],
}


def check(rc):
    if rc == WX_OK:
        return
    msg = lib().wx_last_error().decode("utf-8", "replace")
    if rc == WX_EASSERT:
        raise AssertionError(msg)
    if rc == WX_EARG:
        from .filters import ArgumentError
        raise ArgumentError(msg)
    if rc == WX_EBOUNDS:
        raise IndexError(msg)
    raise WxError(rc, msg)


def device_count():
    return lib().wx_device_count()


def build_info():
    """what the loaded binary was built from (source digest, commit, compiler, flags): wx_build_info"""
    return lib().wx_build_info().decode("utf-8", "replace")


def shutdown():
    """release the library's cached device scratch (wx_shutdown)"""
    check(lib().wx_shutdown())


def set_host_hugepages(on):
    """MADV_HUGEPAGE advice on host result arrays of 64 MiB and more (on by default; the advice persists on the caller's address range:
    include/waveletsext_hip.h); returns the previous setting (wx_set_host_hugepages)"""
    return bool(lib().wx_set_host_hugepages(1 if on else 0))


def set_force_generic(on):
    lib().wx_debug_set_dispatch(2 if on == 2 else (1 if on else 0))
