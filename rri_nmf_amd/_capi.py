"""ctypes binding of librri_hip.so (C ABI: include/rri_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible, every
entry point that would touch the device raises.  Importing this module is safe on a
host without a GPU (the library is only opened on first use).
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RRI_HIP_LIB', os.path.join(_PKG, 'lib', 'librri_hip.so'))

RRI_GRAM_SLICES = 8     # include/rri_hip.h
RRI_OK, RRI_PAUSED = 0, 1
RRI_ERR_INVALID, RRI_ERR_HIP, RRI_ERR_UNSUPPORTED = -1, -2, -3
RRI_ERR_UNBOUNDED, RRI_ERR_W_COL_ZERO, RRI_ERR_NOT_IMPLEMENTED, RRI_ERR_COMM = -4, -5, -6, -7
RRI_COMM_ID_BYTES = 128
RRI_F32, RRI_F64 = 0, 1
RESET_NONE, RESET_MAX_RESID_DOCUMENT, RESET_RANDOM = 0, 1, 2
EVENT_NONE, EVENT_RESET_T, EVENT_RESET_W = 0, 1, 2
ABI_VERSION = 1


class Params(C.Structure):
    """struct rri_params (include/rri_hip.h)"""
    _fields_ = [('fix_W', C.c_int32), ('fix_T', C.c_int32), ('project_T_each_iter', C.c_int32),
                ('has_t_row_sum', C.c_int32), ('has_w_row_sum', C.c_int32), ('reset_method', C.c_int32),
                ('resets_left', C.c_int32), ('reserved0', C.c_int32),
                ('t_row_sum', C.c_double), ('w_row_sum', C.c_double),
                ('reg_w_l1', C.c_double), ('reg_w_l2', C.c_double), ('reg_t_l1', C.c_double),
                ('reg_t_l2', C.c_double), ('eps_div', C.c_double)]


class Event(C.Structure):
    """struct rri_event"""
    _fields_ = [('kind', C.c_int32), ('topic', C.c_int32), ('sweep', C.c_int32), ('resume_topic', C.c_int32)]


_P = C.c_void_p
_I32, _I64, _D = C.c_int32, C.c_int64, C.c_double
# transport callbacks of rri_comm_create_host
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double))
BROADCAST_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int32)
# name -> (restype, argtypes): every symbol include/rri_hip.h declares
PROTOTYPES = {
    'rri_abi_version': (C.c_uint32, []),
    'rri_create': (_I32, [C.POINTER(_P), _I64, _I64, _I32, _I32, _I32, _I32, _P]),
    'rri_destroy': (_I32, [_P]),
    'rri_last_error': (C.c_char_p, [_P]),
    'rri_upload_X': (_I32, [_P, _P, _I64, _I32]),
    'rri_upload_mask': (_I32, [_P, _P, _I64, _I32]),
    'rri_bind_X_device': (_I32, [_P, _P, _I64]),
    'rri_bind_mask_device': (_I32, [_P, _P, _I64]),
    'rri_upload_X_csr': (_I32, [_P, C.POINTER(_I64), C.POINTER(_I32), _P, _I64, _I32]),
    'rri_upload_mask_csr_pattern': (_I32, [_P, C.POINTER(_I64), C.POINTER(_I32), _P, _I64, _I32]),
    'rri_upload_observed_csr': (_I32, [_P, C.POINTER(_I64), C.POINTER(_I32), _P, _I64, _I32]),
    'rri_set_W': (_I32, [_P, _P, _I64, _I32]),
    'rri_set_T': (_I32, [_P, _P, _I64, _I32]),
    'rri_get_W': (_I32, [_P, _P, _I64, _I32]),
    'rri_get_T': (_I32, [_P, _P, _I64, _I32]),
    'rri_set_params': (_I32, [_P, C.POINTER(Params)]),
    'rri_sweep': (_I32, [_P, _I32, C.POINTER(_I32)]),
    'rri_resume': (_I32, [_P, C.POINTER(_I32)]),
    'rri_pending_event': (_I32, [_P, C.POINTER(Event)]),
    'rri_apply_reset_max_resid': (_I32, [_P, _I32, C.POINTER(_I64)]),
    'rri_apply_reset_vectors': (_I32, [_P, _I32, C.POINTER(_D), C.POINTER(_D)]),
    'rri_skip_reset': (_I32, [_P]),
    'rri_update_T_row': (_I32, [_P, _I32]),
    'rri_update_W_col': (_I32, [_P, _I32]),
    'rri_project_W_rows': (_I32, [_P, _D, C.POINTER(_D)]),
    'rri_objective': (_I32, [_P, C.POINTER(_D)]),
    'rri_argmax_rows': (_I32, [_P, C.POINTER(_I32)]),
    'rri_masked_rmse': (_I32, [_P, C.POINTER(_I64), C.POINTER(_D), _I64, _D, _D, C.POINTER(_D)]),
    'rri_residual_rebuild': (_I32, [_P]),
    'rri_residual_update': (_I32, [_P] + [C.POINTER(_D)] * 8),
    'rri_get_residual': (_I32, [_P, _P, _I64, _I32]),
    'rri_snapshot': (_I32, [_P]),
    'rri_rollback': (_I32, [_P]),
    'rri_X_times': (_I32, [_P, C.POINTER(_D), _I32, C.POINTER(_D)]),
    'rri_Xt_times': (_I32, [_P, C.POINTER(_D), _I32, C.POINTER(_D)]),
    'rri_range_finder': (_I32, [_P, C.POINTER(_D), _I32, _I32, _I32, C.POINTER(_D), C.POINTER(_D)]),
    'rri_column_positive_counts': (_I32, [_P, C.POINTER(_D)]),
    'rri_scale_X': (_I32, [_P, C.POINTER(_D), _I32]),
    'rri_comm_unique_id': (_I32, [C.POINTER(C.c_uint8)]),
    'rri_comm_create': (_I32, [C.POINTER(_P), C.POINTER(C.c_uint8), _I32, _I32, _I32]),
    'rri_comm_create_host': (_I32, [C.POINTER(_P), _I32, _I32, ALLREDUCE_FN, ALLGATHER_FN, BROADCAST_FN, _P]),
    'rri_comm_destroy': (_I32, [_P]),
    'rri_attach_comm': (_I32, [_P, _P, _I64, _I64]),
    'rri_comm_broadcast': (_I32, [_P, C.POINTER(_D), _I64, _I32]),
    'rri_comm_allreduce_sum': (_I32, [_P, C.POINTER(_D), _I64]),
    'rri_comm_stats': (_I32, [_P, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I64)]),
    'rri_reduce_buffer': (_I32, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    'rri_bind_reduce_buffer': (_I32, [_P, _P, _I64]),
    'rri_reduce_read': (_I32, [_P, C.POINTER(C.c_double), _I64]),
    'rri_reduce_write': (_I32, [_P, C.POINTER(C.c_double), _I64]),
    'rri_topic_reduce_local': (_I32, [_P, _I32]),
    'rri_topic_finish': (_I32, [_P, _I32]),
    'rri_topic_finish_w': (_I32, [_P, _I32]),
    'rri_resid_row_argmax': (_I32, [_P, C.POINTER(_D), C.POINTER(_I64)]),
    'rri_reset_row': (_I32, [_P, _I64, C.POINTER(_D)]),
    'rri_poll': (_I32, [_P]),
    'rri_objective_parts': (_I32, [_P, C.POINTER(_D)]),
    'rri_timing_enable': (_I32, [_P, _I32]),
    'rri_timing_read': (_I32, [_P, _I32, C.POINTER(_I64), C.POINTER(_D)]),
    'rri_onchip_info': (_I32, [_P, C.POINTER(_I32), C.POINTER(_I64)]),
    'rri_onchip_fallbacks': (_I32, [_P, C.POINTER(_I64)]),
    'rri_debug_xcc': (_I32, [_P, C.POINTER(_I32), _I32]),
    'rri_sweep_until': (_I32, [_P, _I32, _D, _D, C.POINTER(_D), C.POINTER(_I32)]),
    'rri_synchronize': (_I32, [_P]),
    'rri_bench_rank1_update': (_I32, [_P, _I32, C.POINTER(_D)]),
    'rri_bench_stream_copy': (_I32, [_P, _I32, C.POINTER(_D)]),
}

_lib = None


class RRIHipUnavailable(RuntimeError):
    """librri_hip.so cannot be used here (not built, or no HIP device)."""


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels ship their own libamdhip64; if this library pulls in the
    system one first, a later torch.cuda initialisation finds "no HIP GPUs" (two runtimes in one process).  Loading
    torch's copy first (without importing torch) makes the dynamic linker bind librri_hip.so to it as well -- the
    order that has always worked, now independent of who imports what first.  RRI_HIP_OWN_RUNTIME=1 skips this."""
    if os.environ.get('RRI_HIP_OWN_RUNTIME', '0') == '1':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001  (best effort: the system runtime is the fallback)
        pass


def share_rccl_with_torch():
    """The RCCL the library resolves at run time must be the one bound to the HIP runtime in use: when torch is
    installed that is torch's bundled librccl.so (same reasoning as above).  Making it global lets the library's
    dlsym(RTLD_DEFAULT, "ncclAllReduce") find it; without torch the library opens the system librccl.so.1 itself."""
    if os.environ.get('RRI_HIP_OWN_RUNTIME', '0') == '1' or os.environ.get('RRI_RCCL_LIB'):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'librccl.so')
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001  (best effort: the library falls back to the system RCCL)
        pass


def load_library(path=None):
    """Opens librri_hip.so and types every entry point.  Raises RRIHipUnavailable
    if the file is missing; there is deliberately no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RRIHipUnavailable(
            'librri_hip.so not found at %s: build it with `python -m rri_nmf_amd.build` '
            '(needs hipcc; the RRI path has no CPU fallback)' % p)
    _share_hip_runtime_with_torch()
    try:
        lib = C.CDLL(p)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RRIHipUnavailable('cannot load %s: %s' % (p, e))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.rri_abi_version() != ABI_VERSION:
        raise RRIHipUnavailable('librri_hip.so has ABI %d, binding expects %d'
                                % (lib.rri_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib
