"""
ctypes binding of libsdfs_hip.so (C ABI: include/sdfs_hip.h).

There is no CPU fallback: importing this module raises if the shared library is
missing, and creating an operator raises if no HIP device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("SDFS_LIB_NAME", "libsdfs_hip.so"))   # SDFS_LIB_NAME: diagnostic builds

SDFS_MODEL_SSY, SDFS_MODEL_GCY = 0, 1
SDFS_ALGO_SA, SDFS_ALGO_NEWTON, SDFS_ALGO_ANDERSON = 0, 1, 2
SDFS_ERR_UNSUPPORTED = -3
SDFS_ERR_NUMERIC = -4
SDFS_MAX_KERNELS = 16


class SdfsError(RuntimeError):
    pass


class sdfs_opts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int64),
                ("inner_rtol", C.c_double), ("inner_atol", C.c_double),
                ("inner_max_iter", C.c_int64),
                ("history", C.c_int32), ("mixing_freq", C.c_int32),
                ("beta", C.c_double), ("ridge", C.c_double),
                ("check_every", C.c_int32), ("use_graph", C.c_int32),
                ("record_errors", C.c_int32), ("krylov_f32", C.c_int32), ("t_f32", C.c_int32)]


class sdfs_kernel_counter(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("alg_bytes", C.c_double), ("alg_flops", C.c_double)]


class sdfs_counters(C.Structure):
    _fields_ = [("nkernels", C.c_int32), ("reserved", C.c_int32),
                ("k", sdfs_kernel_counter * SDFS_MAX_KERNELS)]


# every symbol include/sdfs_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I64 = C.POINTER(C.c_int64)
SYMBOLS = {
    "sdfs_create": (C.c_int, [C.c_int, C.c_int, _I64, _D, C.c_int, C.POINTER(_D), _I64, C.c_int,
                              C.c_int, C.POINTER(_P)]),
    "sdfs_create_sharded": (C.c_int, [C.c_int, C.c_int, _I64, _D, C.c_int, C.POINTER(_D), _I64, C.c_int,
                                      C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int64,
                                      C.c_int64, C.POINTER(_P)]),
    "sdfs_create_continuous": (C.c_int, [C.c_int, C.c_int, _I64, _D, C.c_int, C.POINTER(_D), _D, _D, C.c_int64,
                                         C.c_int, C.POINTER(_P)]),
    "sdfs_create_dense": (C.c_int, [C.c_int64, _D, C.c_double, C.c_double, C.c_int, C.POINTER(_P)]),
    "sdfs_lin_interp": (C.c_int, [C.c_int, C.c_int, _I64, C.POINTER(_D), _D, _D, C.c_int64, _D]),
    "sdfs_destroy": (None, [_P]),
    "sdfs_last_error": (C.c_char_p, [_P]),
    "sdfs_default_opts": (C.c_int, [C.POINTER(sdfs_opts)]),
    "sdfs_grid_size": (C.c_int64, [_P]),
    "sdfs_set_stream": (C.c_int, [_P, _P, C.c_int]),
    "sdfs_synchronize": (C.c_int, [_P]),
    "sdfs_apply_T": (C.c_int, [_P, _P, _P]),
    "sdfs_apply_T_dev": (C.c_int, [_P, _P, _P, _P]),
    "sdfs_apply_jvp": (C.c_int, [_P, _P, _P, _P]),
    "sdfs_linearize_dev": (C.c_int, [_P, _P, _P]),
    "sdfs_apply_jvp_dev": (C.c_int, [_P, _P, _P, C.c_int]),
    "sdfs_apply_vjp": (C.c_int, [_P, _P, _P, _P]),
    "sdfs_apply_vjp_dev": (C.c_int, [_P, _P, _P, C.c_int]),
    "sdfs_residual": (C.c_int, [_P, _D]),
    "sdfs_solve": (C.c_int, [_P, C.c_int, C.POINTER(sdfs_opts), _P, _I64, _I64, _D]),
    "sdfs_solve_dev": (C.c_int, [_P, C.c_int, C.POINTER(sdfs_opts), _P, _I64, _I64, _D]),
    "sdfs_error_trace": (C.c_int64, [_P, _P, C.c_int64]),
    "sdfs_apply_stage_dev": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "sdfs_apply_stage_gated_dev": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_double]),
    "sdfs_pack_blocks": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "sdfs_krylov_step": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int, C.POINTER(_P), _P, C.c_double, C.c_double]),
    "sdfs_krylov_scalars": (C.c_int, [_P, _P]),
    "sdfs_krylov_gate": (C.c_int, [_P, C.POINTER(_P)]),
    "sdfs_unpack_blocks_sub": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "sdfs_anderson_begin": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, C.c_double, C.c_int64, C.c_double, C.c_double, C.c_int]),
    "sdfs_anderson_gate": (C.c_int, [_P, C.POINTER(_P)]),
    "sdfs_anderson_step": (C.c_int, [_P, C.c_int, C.c_int64, _P, _P, _P]),
    "sdfs_anderson_state": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64]),
    "sdfs_set_krylov_f32": (C.c_int, [_P, C.c_int, C.c_double]),
    "sdfs_set_t_f32": (C.c_int, [_P, C.c_int, C.c_double]),
    "sdfs_set_profiling": (C.c_int, [_P, C.c_int]),
    "sdfs_reset_counters": (C.c_int, [_P]),
    "sdfs_get_counters": (C.c_int, [_P, C.POINTER(sdfs_counters)]),
    "sdfs_describe_plan": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "sdfs_stream_copy_dev": (C.c_int, [_P, _P, _P, C.c_int64]),
    "sdfs_debug_pow": (C.c_int, [_P, C.c_double, _P, C.c_int64, C.c_int]),
    "sdfs_debug_powy": (C.c_int, [_P, C.c_double, _P, C.c_int64, C.c_int, C.c_int]),
}


def _load():
    # PyTorch-ROCm ships its own copy of the HIP runtime.  If libsdfs_hip.so pulled in /opt/rocm's
    # copy first, a later torch.cuda initialisation in the same process finds "No HIP GPUs"; loading
    # torch first makes both share one runtime (torch is what the callers use for device memory and
    # torch.distributed anyway).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise SdfsError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
            "sdfs_via_autodiff_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error(handle=None):
    msg = lib.sdfs_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, handle=None, allow=()):
    if rc != 0 and rc not in allow:
        raise SdfsError(f"libsdfs_hip error {rc}: {last_error(handle)}")
    return rc


def default_opts():
    o = sdfs_opts()
    check(lib.sdfs_default_opts(C.byref(o)))
    return o
