"""ctypes binding of librecad_hip.so (include/recad_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails this
module raises.  Pointers are taken from torch tensors with ``data_ptr()``; torch is only
the owner of device memory and streams here.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product reads no environment variable.  A/B work against a variant build of the same ABI (scripts/ only) calls
# load(path) explicitly BEFORE anything else touches the library.
LIB_PATH = os.path.join(_HERE, "lib", "librecad_hip.so")
RK_LOSS_PARTIALS = 256
RK_MAX_GRAPH_STEPS = 64
ABI_VERSION = 9
RK_LDS_SYNC_WORDS = 2560


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


class LdsInfo(C.Structure):
    """rk_lds_info (include/recad_hip.h)."""

    _fields_ = [
        ("n_wg", C.c_int32), ("lds_bytes", C.c_int32), ("lpa", C.c_int32), ("lpb", C.c_int32),
        ("n_users", C.c_int32), ("n_items", C.c_int32), ("dim", C.c_int32), ("lsu", C.c_int32), ("lsi", C.c_int32),
        ("chunk", C.c_int32), ("wgx_ofs", C.c_int32), ("dinv_ofs", C.c_int32), ("perm0_ofs", C.c_int32), ("perm1_ofs", C.c_int32),
        ("mq_ofs", C.c_int32), ("reserved", C.c_int32),
    ]


class LdsEpilogue(C.Structure):
    """rk_lds_epilogue (include/recad_hip.h)."""

    _fields_ = [
        ("add", C.c_void_p), ("y", C.c_void_p), ("sum_in", C.c_void_p), ("sum_out", C.c_void_p),
        ("sum_scale", C.c_float), ("y_row_major", C.c_int32), ("sum_out_row_major", C.c_int32), ("adam_t", C.c_int32),
        ("zero1", C.c_void_p), ("zero2", C.c_void_p),
        ("adam_p", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_shadow", C.c_void_p), ("coef_scratch", C.c_void_p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("stamps", C.c_void_p),
    ]


class LightGCNDesc(C.Structure):
    """rk_lightgcn_desc (include/recad_hip.h)."""

    _fields_ = [
        ("n_users", C.c_int32), ("n_items", C.c_int32), ("dim", C.c_int32), ("n_layers", C.c_int32),
        ("lam", C.c_float), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("reserved0", C.c_int32),
        ("rowptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p), ("wave_desc", C.c_void_p),
        ("n_blocks", C.c_int32), ("reserved1", C.c_int32),
        ("user_emb", C.c_void_p), ("item_emb", C.c_void_p),
        ("m_user", C.c_void_p), ("v_user", C.c_void_p), ("m_item", C.c_void_p), ("v_item", C.c_void_p),
        ("buf_a", C.c_void_p), ("buf_b", C.c_void_p), ("light", C.c_void_p), ("gprop", C.c_void_p), ("gego", C.c_void_p),
        ("grad", C.c_void_p), ("state", C.c_void_p), ("coef", C.c_void_p),
        ("spmm_scratch", C.c_void_p),
        ("row_bits", C.c_void_p),
        ("keep_prob", C.c_float), ("reserved3", C.c_int32), ("drop_seed", C.c_uint64), ("tpos", C.c_void_p),
        ("lds_plan", C.c_void_p), ("lds_info", LdsInfo), ("lsum", C.c_void_p), ("e0s", C.c_void_p), ("ms", C.c_void_p), ("vs", C.c_void_p), ("cnt", C.c_void_p),
        ("row_blocks", C.c_void_p), ("row_blocks_extra", C.c_int32), ("reserved4", C.c_int32),
        ("lds_sync", C.c_void_p),
    ]


class ScorePlan(C.Structure):
    """rk_score_plan (include/recad_hip.h): the scoring path and its shape knobs as an ARGUMENT of rk_score_topk"""

    _fields_ = [
        ("path", C.c_int32), ("panel_rows", C.c_int32), ("panel_ntw", C.c_int32), ("panel_safe", C.c_int32),
        ("nb", C.c_int32), ("n_items", C.c_int32), ("dim", C.c_int32), ("K", C.c_int32), ("n_targets", C.c_int32),
        ("ld_scores", C.c_int32), ("reserved", C.c_int32 * 2), ("scratch_floats", C.c_int64),
    ]


RK_SCORE_AUTO, RK_SCORE_GEMM, RK_SCORE_PANEL = 0, 1, 2


class SpmmEpilogue(C.Structure):
    """rk_spmm_epilogue (include/recad_hip.h)."""

    _fields_ = [
        ("add", C.c_void_p), ("y", C.c_void_p), ("sum_in", C.c_void_p), ("sum_out", C.c_void_p),
        ("sum_scale", C.c_float), ("adam_t", C.c_int32), ("zero1", C.c_void_p), ("zero2", C.c_void_p),
        ("adam_p", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("coef_scratch", C.c_void_p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("src_filter", C.c_void_p),
    ]


class NCFDesc(C.Structure):
    """rk_ncf_desc (include/recad_hip.h)."""

    _fields_ = [
        ("n_users", C.c_int32), ("n_items", C.c_int32), ("factor", C.c_int32), ("n_layers", C.c_int32),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("ug", C.c_void_p), ("ig", C.c_void_p), ("um", C.c_void_p), ("im", C.c_void_p),
        ("W", C.c_void_p * 8), ("b", C.c_void_p * 8), ("pw", C.c_void_p), ("pb", C.c_void_p),
        ("grad", C.c_void_p * 24), ("m", C.c_void_p * 24), ("v", C.c_void_p * 24),
        ("acts", C.c_void_p), ("dacts", C.c_void_p), ("d0", C.c_void_p),
        ("max_batch", C.c_int32), ("reserved", C.c_int32),
        ("gemm_scratch", C.c_void_p), ("gemm_scratch_floats", C.c_int64), ("wgrad_part", C.c_void_p),
        ("mode", C.c_int32), ("dropout", C.c_float), ("drop_seed", C.c_uint64), ("drop_call", C.c_int32), ("reserved2", C.c_int32),
    ]


_lib = None

# name -> argtypes (restype is always int unless listed in _RESTYPES)
_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_SIGNATURES = {
    "rk_abi_version": [],
    "rk_last_error": [],
    "rk_device_info": [C.c_char_p, _I32, C.POINTER(_I32)],
    "rk_coo_to_csr": [_I32, _I64, _P, _P, _P, _P, _P, _P, _P],
    "rk_csr_schedule_build": [_I32, _P, _I32, _I32, _P, C.POINTER(_P), C.POINTER(_I32), C.POINTER(_I64), C.POINTER(_I64)],
    "rk_csr_schedule_build_host": [_I32, _P, _I32, _I32, C.POINTER(_P), C.POINTER(_I32), C.POINTER(_I64), C.POINTER(_I64)],
    "rk_csr_schedule_words": [_P, _P],
    "rk_csr_schedule_upload": [_P, _P, _P],
    "rk_csr_schedule_destroy": [_P],
    "rk_build_norm_adj": [_I32, _I32, _P, _P, _P, _P, _P, _P, _P],
    "rk_spmm_csr": [_I32, _P, _P, _P, _P, _I32, _P, _I32, _P, _P, _P, _P],
    "rk_spmm_csr_ex": [_I32, _P, _P, _P, _P, _I32, _P, _I32, _P, _I64, C.POINTER(SpmmEpilogue), _P],
    "rk_lds_plan_build": [_I32, _I32, _P, _P, _P, _I32, _P, C.POINTER(_P), C.POINTER(_I64), C.POINTER(LdsInfo)],
    "rk_lds_plan_build_host": [_I32, _I32, _P, _P, _P, _I32, _I32, C.POINTER(_P), C.POINTER(_I64), C.POINTER(LdsInfo)],
    "rk_lds_plan_words": [_P, _P],
    "rk_lds_plan_upload": [_P, _P, _P],
    "rk_lds_plan_destroy": [_P],
    "rk_lds_pack": [C.POINTER(LdsInfo), _P, _P, _I32, _I64, _P],
    "rk_lds_unpack": [C.POINTER(LdsInfo), _P, _P, _I32, _I64, _P],
    "rk_spmm_lds": [C.POINTER(LdsInfo), _P, _P, C.POINTER(LdsEpilogue), _P],
    "rk_rows_gather_masked": [_I32, _P, _P, _P, _I64, _P, _P],
    "rk_rows_zero": [_I32, _P, _P, _P, _I64, _P],
    "rk_rows_mark_bits": [_P, _P, _I64, _I32, _P],
    "rk_adam_coef_advance": [_P, _P, _F, _F, _F, _P],
    "rk_bpr_rows": [_I32, _I32, _F, _P, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _P],
    "rk_bpr_rows_ordered": [_I32, _I32, _F, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P],
    "rk_lightgcn_create": [C.POINTER(LightGCNDesc), C.POINTER(_P)],
    "rk_lightgcn_destroy": [_P],
    "rk_lightgcn_sync_status": [_P, C.POINTER(_I32), _P],
    "rk_lightgcn_propagate": [_P, _P],
    "rk_lightgcn_propagate_dropout": [_P, C.c_uint64, _P],
    "rk_lightgcn_train_epoch": [_P, _P, _P, _P, _I64, _I32, _I32, _P, _I32, _I32, _P],
    "rk_lightgcn_prepare": [_P, _I64, _I32, _I32, _I32, _P],
    "rk_lightgcn_set_deterministic": [_P, _I32],
    "rk_pair_scores": [_I32, _P, _P, _P, _P, _F, _P, _P, _I64, _P, _F, C.c_uint64, _P],
    "rk_score_matrix": [_I32, _P, _I32, _P, _P, _I32, _P, _P, _F, _F, C.c_uint64, _P, _P],
    "rk_adam_step": [_I64, _P, _P, _P, _P, _I32, _F, _F, _F, _F, _P],
    "rk_adam_step_dev": [_I64, _P, _P, _P, _P, _P, _F, _F, _F, _P],
    "rk_score_topk_plan": [_I32, _I32, _I32, _I32, _I32, C.POINTER(ScorePlan), C.POINTER(ScorePlan)],
    "rk_score_topk": [_I32, _P, _I32, _P, _P, _I32, _P, _P, _F, _P, _P, _I32, _P, _P, _P, _I32, _P, _P, C.POINTER(ScorePlan), _P, _P],
    "rk_bpr_sample": [_I32, _I32, _P, _P, _I64, C.c_uint64, _P, _P, _P, _P, _P],
    "rk_pointwise_sample": [_I32, _I32, _P, _P, _I64, _I32, C.c_uint64, _P, _P, _P, _P],
    "rk_topk_rows": [_P, _I32, _I32, _P, _P, _P, _I32, _P, _P, _P, _I32, _P, _P, _P],
    "rk_hit_counts": [_P, _I64, _I32, _P, _I32, _P, _P],
    "rk_eligible_users": [_I32, _P, _P, _P, _I32, _P, _P, _P, _P],
    "rk_pred_shift": [_P, _P, _I64, _P, _P],
    "rk_users_rating": [_I32, _P, _I32, _P, _P, _I32, _P, _P],
    "rk_ncf_forward": [C.POINTER(NCFDesc), _P, _P, _P, _I32, _I64, _P, _P],
    "rk_ncf_train_epoch": [C.POINTER(NCFDesc), _P, _P, _P, _I64, _I32, _I32, _P, _I32, _P],
    "rk_mf_train_epoch": [_I32, _I32, _I32, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _F, _F, _F,
                          _F, _P, _I32, _F, C.c_uint64, _P],
}
_RESTYPES = {"rk_last_error": C.c_char_p}
EXPORTS = tuple(_SIGNATURES)


def load(path):
    """Tuning scripts only: bind a variant build of the same ABI (e.g. lib/librecad_hip_tuning.so, or its bare file name
    under recad_amd/lib/) instead of the product library.  Must be called before the first lib(); the product never calls it."""
    global _lib, LIB_PATH
    if _lib is not None:
        raise HipCallError(f"_lib.load({path!r}): {LIB_PATH} is already loaded")
    LIB_PATH = path if os.path.isabs(path) or os.path.exists(path) else os.path.join(_HERE, "lib", path)
    return lib()


def lib():
    """Load librecad_hip.so once; raise HipLibraryMissing when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  recad_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise HipLibraryMissing(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        if handle.rk_abi_version() != ABI_VERSION:
            raise HipLibraryMissing(f"{LIB_PATH} has ABI {handle.rk_abi_version()}, expected {ABI_VERSION}; rebuild it")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().rk_last_error()
        raise HipCallError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: no CPU fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipCallError("recad_amd kernels need tensors on a HIP device (got a CPU tensor); "
                           "move the model/dataset to 'cuda' first -- there is no CPU fallback")
    if not t.is_contiguous():
        raise HipCallError("recad_amd kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None):
    """The current HIP stream of `device` (default: the current device) as a pointer argument.  Through torch's raw accessor when it
    exists (0.3 us; torch.cuda.current_stream() builds a Stream object and resolves the device index in Python: 4 us, on every call
    of the C ABI)."""
    if _raw_stream is not None:
        if device is None:
            idx = torch.cuda.current_device()
        else:
            idx = device if isinstance(device, int) else torch.device(device).index
            if idx is None:
                idx = torch.cuda.current_device()
        return C.c_void_p(_raw_stream(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise HipCallError("no HIP device visible: recad_amd's hot path only runs on an MI355X (gfx950)")
