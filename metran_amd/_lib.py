"""ctypes binding of ``libmetran_hip.so`` (C ABI declared in ``include/metran_hip.h``).

There is NO CPU fallback: if the shared library is missing, or no gfx950 device is
visible, every compute entry point raises ``MetranHipError``.  (``oracle/`` is test
infrastructure and is never imported from here.)
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint32, c_void_p

__all__ = ["MetranHipError", "lib", "library_path", "Problem", "Outputs", "check", "API"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "libmetran_hip.so"


class MetranHipError(Exception):
    """Raised for every failure of the HIP library (the reference raises bare ``Exception``
    with a logged message, e.g. /root/reference/metran/kalmanfilter.py:733-745)."""


c_dp = POINTER(c_double)
c_i64p = POINTER(c_int64)
c_u32p = POINTER(c_uint32)


class Problem(Structure):
    """``mk_problem`` (include/metran_hip.h)."""

    _fields_ = [
        ("n_instances", c_int64),
        ("n_records", c_int64),
        ("T", c_int64),
        ("N", c_int64),
        ("K", c_int64),
        ("warmup", c_int64),
        ("d_obs", c_void_p),
        ("d_phi", c_void_p),
        ("d_q", c_void_p),
        ("d_loadings", c_void_p),
        ("d_obsvar", c_void_p),
        ("d_x0", c_void_p),
        ("d_P0", c_void_p),
        ("obs_time_major", c_int64),
        ("d_scale", c_void_p),
        ("d_offset", c_void_p),
    ]


class Outputs(Structure):
    """``mk_outputs`` (include/metran_hip.h)."""

    _fields_ = [
        ("d_mle", c_void_p),
        ("d_sigmas", c_void_p),
        ("d_detfs", c_void_p),
        ("d_sigmacount", c_void_p),
        ("d_F", c_void_p),
        ("d_Pf", c_void_p),
        ("d_Xp", c_void_p),
        ("d_Pp", c_void_p),
        ("d_S", c_void_p),
        ("d_Ps", c_void_p),
        ("d_status", c_void_p),
        ("time_major", c_int64),
        ("d_sim_means", c_void_p),
        ("d_sim_vars", c_void_p),
        ("record_stride", c_int64),
        ("flags", c_int64),
    ]


# name -> (restype, argtypes); this table is what tests check against the header.
API = {
    "mk_abi_version": (c_int, []),
    "mk_last_error": (c_char_p, []),
    "mk_device_count": (c_int, [POINTER(c_int)]),
    "mk_create": (c_int, [c_int, POINTER(c_void_p)]),
    "mk_destroy": (c_int, [c_void_p]),
    "mk_set_stream": (c_int, [c_void_p, c_void_p]),
    "mk_sync": (c_int, [c_void_p]),
    "mk_observations_changed": (c_int, [c_void_p]),
    "mk_shape_supported": (c_int, [c_int64, c_int64]),
    "mk_shape_specialised": (c_int, [c_int64, c_int64]),
    "mk_generic_max_states": (c_int64, []),
    "mk_register_shape_module": (c_int, [c_char_p]),
    "mk_record_stride": (c_int64, [c_int64]),
    "mk_record_stride_sym": (c_int64, [c_int64]),
    "mk_tape_stride": (c_int64, [c_int64, c_int64]),
    "mk_tape_supported": (c_int, [c_int64, c_int64]),
    "mk_state_tape_stride": (c_int64, [c_int64, c_int64]),
    "mk_supported_shapes": (c_int, [c_i64p, c_int]),
    "mk_malloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "mk_free": (c_int, [c_void_p, c_void_p]),
    "mk_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "mk_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "mk_memset": (c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    "mk_params_from_alpha": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_double,
                                     c_void_p, c_void_p]),
    "mk_filter": (c_int, [c_void_p, POINTER(Problem), POINTER(Outputs)]),
    "mk_loglik": (c_int, [c_void_p, POINTER(Problem), c_void_p]),
    "mk_smooth": (c_int, [c_void_p, POINTER(Problem), POINTER(Outputs)]),
    "mk_filter_smooth": (c_int, [c_void_p, POINTER(Problem), POINTER(Outputs)]),
    "mk_smooth_dense": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "mk_simulate": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p]),
    "mk_decompose": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                             c_void_p]),
    "mk_sum": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mk_comm_set_library": (c_int, [c_char_p]),
    "mk_comm_unique_id": (c_int, [c_void_p]),
    "mk_comm_init_rank": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mk_set_communicator": (c_int, [c_void_p, c_void_p]),
    "mk_comm_destroy": (c_int, [c_void_p]),
    "mk_allreduce_sum": (c_int, [c_void_p, c_void_p, c_int64]),
    "mk_loglik_grad": (c_int, [c_void_p, POINTER(Problem), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mk_loglik_grad_phases": (c_int, [c_void_p, POINTER(Problem), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int]),
    "mk_adjoint_update_stride": (c_int64, [c_int64, c_int64]),
    "mk_set_adjoint_updates": (c_int, [c_void_p, c_void_p, c_int64]),
    "mk_lbfgs_direction": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "mk_lbfgs_trial": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mk_lbfgs_armijo": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_int64, c_void_p, POINTER(c_int), POINTER(c_int)]),
    "mk_lbfgs_update": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int64, POINTER(c_int)]),
    "mk_alpha_grad": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_double, c_void_p,
                              c_void_p, c_void_p]),
    "mk_standardize": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mk_mask_observations": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "mk_pack_observations": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mk_set_kernel_variant": (c_int, [c_void_p, c_int, c_int]),
    "mk_get_kernel_variant": (c_int, [c_void_p, c_int, POINTER(c_int)]),
    "mk_fa_correlation": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mk_fa_analyse": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p]),
    "mk_fa_minres": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p]),
    "mk_fa_rotate": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_double, c_int, c_double]),
    "mk_fa_eigh": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "mk_enable_timing": (c_int, [c_void_p, c_int]),
    "mk_last_kernel_ms": (c_int, [c_void_p, POINTER(c_float), POINTER(c_float)]),
    "mk_kernel_ms_totals": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64), POINTER(c_double), POINTER(c_int64)]),
}

ABI_VERSION = 7  # MK_ABI_VERSION of include/metran_hip.h
_lib = None


def library_path():
    return os.environ.get("METRAN_HIP_LIBRARY", os.path.join(_HERE, _LIBNAME))


def lib():
    """Load (once) and return the bound library; raises MetranHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise MetranHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C metran_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path
        )
    try:
        L = ctypes.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise MetranHipError("cannot load %s: %s" % (path, e)) from e
    for name, (res, args) in API.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise MetranHipError("%s does not export %s (stale build?)" % (path, name)) from e
        fn.restype = res
        fn.argtypes = args
    if L.mk_abi_version() != ABI_VERSION:
        raise MetranHipError("ABI version mismatch: library %d, binding %d" % (L.mk_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().mk_last_error()
        raise MetranHipError("libmetran_hip error %d: %s" % (rc, (msg or b"").decode("utf-8", "replace")))
