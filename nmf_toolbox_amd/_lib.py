"""ctypes binding of libnmfx.so -- the C ABI declared in include/nmfx.h.

There is no CPU fallback: if the library is missing it is an ImportError-style failure, and every
compute entry point raises NmfxError when no MI355X is usable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NMFX_LIB_VARIANT=<name>: an A/B build of the kernel switches (build.py --variant), for measurements only
LIB_PATH = os.path.join(_HERE, "libnmfx_%s.so" % os.environ["NMFX_LIB_VARIANT"] if os.environ.get("NMFX_LIB_VARIANT") else "libnmfx.so")

NMFX_OK, NMFX_ERR_INVALID, NMFX_ERR_NO_DEVICE, NMFX_ERR_HIP, NMFX_ERR_UNSUPPORTED, NMFX_ERR_NOMEM, NMFX_ERR_NEGATIVE = range(7)
DIV_EUCLIDEAN, DIV_KL, DIV_IS, DIV_AB, DIV_EUCLIDEAN_NOCOST = range(5)
F32, F64 = 0, 1
ABI_VERSION = 600   # the NMFX_VERSION of include/nmfx.h the structures below were written against (load() refuses any other library)

# every symbol include/nmfx.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "nmfx_nmf", "nmfx_cnmf", "nmfx_lnmf", "nmfx_nmfsc", "nmfx_cnmfsc", "nmfx_reconstruct", "nmfx_projfunc", "nmfx_last_error",
    "nmfx_device_count", "nmfx_version", "nmfx_engine_workspace_bytes", "nmfx_engine_packed_count",
    "nmfx_engine_create", "nmfx_engine_destroy", "nmfx_engine_init", "nmfx_engine_wstep_partial",
    "nmfx_engine_wstep_finish", "nmfx_engine_hstep", "nmfx_engine_cost_pass", "nmfx_engine_is_fused", "nmfx_engine_defer_hstep_finish", "nmfx_engine_hstep_finish", "nmfx_engine_cost_ptr", "nmfx_engine_copy_cost", "nmfx_engine_set_rank0",
    "nmfx_engine_iterate", "nmfx_engine_profile", "nmfx_engine_profile_ntags", "nmfx_engine_profile_tag_name",
    "nmfx_engine_profile_read", "nmfx_engine_tag_work", "nmfx_gemm_f32", "nmfx_constrainednmf", "nmfx_sort_dictionary",
    "nmfx_engine_set_constraint", "nmfx_nmfsc_dev", "nmfx_engine_wstep_partial_chunk", "nmfx_engine_packed_chunk",
    "nmfx_engine_between_allreduces", "nmfx_engine_between_allreduces_cost", "nmfx_projfunc_dev", "nmfx_nmfsc_profile", "nmfx_nmfsc_profile_ntags", "nmfx_nmfsc_profile_tag_name", "nmfx_nmfsc_profile_read", "nmfx_last_call_timing", "nmfx_sc_iteration_seconds", "nmfx_engine_cost_lag", "nmfx_engine_sumvv_local", "nmfx_engine_sumvv_set_global",
    "nmfx_minmax_dev", "nmfx_scale_dev", "nmfx_gemm64", "nmfx_engine_sync_master", "nmfx_engine_master_ptrs", "nmfx_engine_init_f64", "nmfx_last_call_exchange", "nmfx_rccl_library", "nmfx_abi_sizes",
]


class NmfxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class Problem(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("K_total", C.c_int32), ("T", C.c_int32), ("dtype", C.c_int32),
        ("V", C.c_void_p), ("W_init", C.c_void_p), ("H_init", C.c_void_p),
        ("divergence", C.c_int32), ("alpha", C.c_double), ("beta", C.c_double),
        ("num_sources", C.c_int32), ("K_s", C.c_void_p), ("W_sparsity", C.c_void_p), ("H_sparsity", C.c_void_p),
        ("W_fixed", C.c_void_p), ("H_fixed", C.c_void_p),
        ("maxiter", C.c_int32), ("tolerance", C.c_double), ("device", C.c_int32),
        ("sc_W_sparsity", C.c_double), ("sc_H_sparsity", C.c_double), ("path", C.c_int32),
        ("sc_stepsize_H0", C.c_double), ("sc_stepsize_W0", C.c_double), ("sc_resume", C.c_int32), ("n_gpus", C.c_int32), ("device_ids", C.c_void_p),
        ("multi_backend", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("W", C.c_void_p), ("H", C.c_void_p), ("cost", C.c_void_p), ("cost_len", C.c_int32), ("iters_run", C.c_int32),
        ("tries_H", C.c_void_p), ("tries_W", C.c_void_p), ("stepsize_H", C.c_double), ("stepsize_W", C.c_double),
        ("converged_early", C.c_int32),
    ]


class EngineDesc(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n_local", C.c_int64), ("K_total", C.c_int32), ("T", C.c_int32), ("divergence", C.c_int32),
        ("alpha", C.c_double), ("beta", C.c_double),
        ("lamW_col", C.c_void_p), ("lamH_row", C.c_void_p), ("fixW_col", C.c_void_p), ("fixH_row", C.c_void_p),
        ("device", C.c_int32), ("stream", C.c_void_p), ("col_offset", C.c_int64), ("path", C.c_int32),
        ("halo_left", C.c_int32), ("halo_right", C.c_int32), ("n_valid", C.c_int64), ("algorithm", C.c_int32), ("K_valid", C.c_int32), ("flags", C.c_int32),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)   # nmfx_allreduce_fn
REDUCE_SUM, REDUCE_MAX = 0, 1

_lib = None


def load():
    """Load libnmfx.so (building it is `python -m nmf_toolbox_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "nmf_toolbox_amd: %s not found. Build the HIP extension first (python -m nmf_toolbox_amd.build). "
            "There is no CPU fallback." % LIB_PATH)
    try:  # share ONE HIP runtime with torch when torch is around (see build.py)
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    # the library reads every field of the structures it is handed: a binding written against another header must not call in
    ver = lib.nmfx_version() if hasattr(lib, "nmfx_version") else -1
    if ver != ABI_VERSION:
        raise ImportError("nmf_toolbox_amd: %s is ABI version %d, this binding was written against %d -- rebuild (python -m nmf_toolbox_amd.build)" % (LIB_PATH, ver, ABI_VERSION))
    sz = (C.c_int32 * 3)()
    lib.nmfx_abi_sizes(C.byref(sz, 0), C.byref(sz, 4), C.byref(sz, 8))
    mine = (C.sizeof(Problem), C.sizeof(Result), C.sizeof(EngineDesc))
    if tuple(sz) != mine:
        raise ImportError("nmf_toolbox_amd: structure sizes differ between %s %r and this binding %r (nmfx_problem, nmfx_result, nmfx_engine_desc)" % (LIB_PATH, tuple(sz), mine))
    lib.nmfx_last_error.restype = C.c_char_p
    lib.nmfx_engine_profile_tag_name.restype = C.c_char_p
    lib.nmfx_engine_profile_tag_name.argtypes = [C.c_int32]
    lib.nmfx_nmfsc_profile_tag_name.restype = C.c_char_p
    lib.nmfx_nmfsc_profile_tag_name.argtypes = [C.c_int32]
    lib.nmfx_nmfsc_profile.argtypes = [C.c_int32]
    lib.nmfx_nmfsc_profile_read.argtypes = [C.c_void_p, C.c_void_p]
    lib.nmfx_engine_destroy.restype = None
    lib.nmfx_engine_destroy.argtypes = [C.c_void_p]
    for name in ("nmfx_nmf", "nmfx_cnmf", "nmfx_lnmf", "nmfx_nmfsc", "nmfx_cnmfsc"):
        getattr(lib, name).argtypes = [C.POINTER(Problem), C.POINTER(Result)]
    lib.nmfx_constrainednmf.argtypes = [C.POINTER(Problem), C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(Result), C.c_void_p]
    lib.nmfx_sort_dictionary.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.nmfx_engine_set_constraint.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.nmfx_nmfsc_dev.argtypes = [C.POINTER(Problem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.POINTER(Result)]
    lib.nmfx_reconstruct.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.nmfx_projfunc.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    lib.nmfx_projfunc_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    lib.nmfx_engine_workspace_bytes.argtypes = [C.POINTER(EngineDesc), C.POINTER(C.c_size_t)]
    lib.nmfx_engine_packed_count.argtypes = [C.POINTER(EngineDesc), C.POINTER(C.c_size_t)]
    lib.nmfx_engine_create.argtypes = [C.POINTER(EngineDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.nmfx_engine_is_fused.argtypes = [C.c_void_p]
    lib.nmfx_engine_defer_hstep_finish.argtypes = [C.c_void_p, C.c_int32]
    lib.nmfx_engine_hstep_finish.argtypes = [C.c_void_p]
    for name in ("nmfx_engine_init", "nmfx_engine_wstep_partial", "nmfx_engine_wstep_finish", "nmfx_engine_hstep", "nmfx_engine_cost_pass"):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.nmfx_engine_between_allreduces.argtypes = [C.c_void_p, C.c_int32]
    lib.nmfx_engine_between_allreduces_cost.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.nmfx_engine_wstep_partial_chunk.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.nmfx_engine_packed_chunk.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.nmfx_engine_cost_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.nmfx_engine_copy_cost.argtypes = [C.c_void_p, C.c_void_p]
    lib.nmfx_engine_set_rank0.argtypes = [C.c_void_p, C.c_int32]
    lib.nmfx_engine_iterate.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.nmfx_engine_profile.argtypes = [C.c_void_p, C.c_int32]
    lib.nmfx_engine_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.nmfx_engine_tag_work.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.nmfx_gemm_f32.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_size_t]
    lib.nmfx_last_call_timing.argtypes = [C.POINTER(C.c_double)] * 5
    lib.nmfx_sc_iteration_seconds.argtypes = [C.POINTER(C.c_double), C.c_int32]
    lib.nmfx_sc_iteration_seconds.restype = C.c_int32
    lib.nmfx_engine_cost_lag.argtypes = [C.c_void_p]
    lib.nmfx_engine_sumvv_local.argtypes = [C.c_void_p, C.c_void_p]
    lib.nmfx_engine_sumvv_set_global.argtypes = [C.c_void_p, C.c_void_p]
    lib.nmfx_minmax_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.nmfx_scale_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p]
    lib.nmfx_gemm64.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
    lib.nmfx_engine_sync_master.argtypes = [C.c_void_p]
    lib.nmfx_last_call_exchange.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.nmfx_rccl_library.restype = C.c_char_p
    lib.nmfx_rccl_library.argtypes = [C.POINTER(C.c_int32)]
    lib.nmfx_engine_init_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.nmfx_engine_master_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    _lib = lib
    return lib


def check(status):
    if status != NMFX_OK:
        raise NmfxError(status, load().nmfx_last_error().decode("utf-8", "replace"))


def device_count():
    return int(load().nmfx_device_count())


def last_call_timing():
    """seconds / bytes of the last blocking factorisation on this thread: dict(ingest_s, iterate_s, egress_s, host_bytes_in, host_bytes_out)"""
    v = [C.c_double() for _ in range(5)]
    check(load().nmfx_last_call_timing(*[C.byref(x) for x in v]))
    return dict(zip(("ingest_s", "iterate_s", "egress_s", "host_bytes_in", "host_bytes_out"), (x.value for x in v)))
