"""ctypes binding of libtagrec_hip.so (the C ABI declared in include/tagrec.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).  There
is deliberately no fallback: if the shared object is missing, or a call fails,
`TagrecError` is raised -- nothing here computes on the CPU.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtagrec_hip.so")
ABI_VERSION = 2

LOSS_SOFTPLUS = 0
LOSS_LOGSIGMOID = 1


class TagrecError(RuntimeError):
    pass


# name -> (argtypes); every function returns int except the two noted below
_SIGNATURES = {
    "tagrec_abi_version": [],
    "tagrec_device_info": [POINTER(c_int), POINTER(c_int), c_char_p, c_int],
    "tagrec_graph_create": [POINTER(c_void_p), c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_graph_create_like": [POINTER(c_void_p), c_void_p, c_int64, c_void_p, c_void_p],
    "tagrec_graph_workspace": [c_int64],
    "tagrec_graph_create_ws": [POINTER(c_void_p), c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                               c_void_p],
    "tagrec_graph_create_ws_deferred": [POINTER(c_void_p), c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                        c_void_p],
    "tagrec_graph_create_like_ws": [POINTER(c_void_p), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64],
    "tagrec_graph_destroy": [c_void_p],
    "tagrec_graph_info": [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                          POINTER(c_int64)],
    "tagrec_spmm_f32": [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_spmm_norm_acc_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p],
    "tagrec_spmm_normbwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p],
    "tagrec_spmm_axpy_f32": [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p],
    "tagrec_spmm_ss_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_spmm_normbwd_dot_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int,
                                    c_void_p],
    "tagrec_row_scale_acc_f32": [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_void_p],
    "tagrec_row_dot_f32": [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_void_p],
    "tagrec_rownorm_bwd_dot_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_void_p],
    "tagrec_bpr_dots_f32": [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                            c_void_p, c_void_p],
    "tagrec_rownorm_fwd_f32": [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p],
    "tagrec_rownorm_bwd_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_int, c_int64, c_int,
                               c_void_p],
    "tagrec_bpr_fwd_f32": [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                           c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_bpr_bwd_f32": [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                           c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_ngcf_wgrad_workspace": [c_int, c_int],
    "tagrec_ngcf_dense_fwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_int64, c_void_p],
    "tagrec_ngcf_dense_bwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_ngcf_dense_bwd_norm_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_ngcf_wgrad_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                              c_void_p, c_int64, c_void_p],
    "tagrec_ngcf_dense_fwd_rows_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_void_p, c_void_p],
    "tagrec_ngcf_dense_bwd_rows_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p],
    "tagrec_ngcf_wgrad_rows_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_void_p, c_void_p],
    "tagrec_tgcn_attn_workspace": [c_int, c_int],
    "tagrec_tgcn_attn_fwd_f32": [c_void_p] * 7 + [c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "tagrec_tgcn_attn_bwd_f32": [c_void_p] * 9 + [c_int64, c_int, c_int, c_int, c_int] + [c_void_p] * 7 + [c_int64, c_void_p],
    "tagrec_attn_pull_da_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_attn_invert_fill": [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_tgcn_attn_bwd_ds_f32": [c_void_p] * 8 + [c_int64, c_int, c_int, c_int, c_int] + [c_void_p] * 5 + [c_int64, c_void_p],
    "tagrec_attn_pull_dq_f32": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "tagrec_attn_seg_dq_f32": [c_void_p, c_void_p, c_int64, ctypes.c_int32, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "tagrec_nbr_gather_i32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    "tagrec_plan_flags_workspace": [c_void_p],
    "tagrec_plan_segment_result": [c_void_p, c_int],
    "tagrec_plan_scan_workspace": [c_void_p],
    "tagrec_plan_mark_u8": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p],
    "tagrec_plan_compact_i64": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "tagrec_plan_lookup_i64": [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p],
    "tagrec_inv_filter_workspace": [c_int64],
    "tagrec_inv_filter_i32": [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_int32, c_int64, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "tagrec_attn_keys_i32": [c_void_p, c_int64, ctypes.c_int32, c_void_p, c_void_p],
    "tagrec_tgcn_fuse_fwd_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 12,
    "tagrec_tgcn_fuse_fwd_drop_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 9
                                     + [c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p],
    "tagrec_tgcn_fuse_bwd_workspace": [c_int],
    "tagrec_tgcn_fuse_bwd_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 18
                                + [c_int64, c_void_p],
    "tagrec_tgcn_fuse_wf_workspace": [c_int, c_int],
    "tagrec_tgcn_fuse_wf_result": [c_int, c_int],
    "tagrec_tgcn_fuse_wf_f32": [c_void_p] * 10 + [c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p],
    "tagrec_route_softmax_f32": [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "tagrec_route_rowsum_rsqrt_f32": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "tagrec_route_permute_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "tagrec_route_spmm_f32": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p],
    "tagrec_route_score_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tagrec_route_spmm_ex_f32": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_route_score_rows_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "tagrec_slice_scale_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p],
    "tagrec_slice_norm_fwd_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p],
    "tagrec_slice_norm_bwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p],
    "tagrec_row_softmax_fwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_row_softmax_bwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_spmm_norm_acc_drop_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, ctypes.c_uint64, c_int,
                                      c_void_p],
    "tagrec_spmm_normbwd_drop_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, ctypes.c_uint64, c_void_p,
                                     c_int, c_void_p],
    "tagrec_dropout_f32": [c_void_p, c_void_p, c_int64, c_float, ctypes.c_uint64, c_void_p],
    "tagrec_dropout_rows_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, ctypes.c_uint64, c_void_p],
    "tagrec_rownorm_bwd_flags_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_int, c_int64, c_int, c_void_p,
                                     c_void_p, c_void_p],
    "tagrec_spmm_normbwd_sparse_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                       ctypes.c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_spmm_axpy_sparse_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p],
    "tagrec_spmm_normbwd_dot_sparse_f32": [c_void_p] * 8 + [c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_graph_mark_rows_u8": [c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    "tagrec_graph_mark_cols_u8": [c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    "tagrec_spmm_norm_acc_rows_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_float, ctypes.c_uint64,
                                      c_int, c_void_p],
    "tagrec_spmm_rows_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_spmm_ss_rows_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_spmm_flags_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_spmm_listed_workspace": [c_int64, c_int],
    "tagrec_spmm_listed_f32": [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p],
    "tagrec_row_flags_f32": [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    "tagrec_eval_topk_f32": [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p,
                             c_void_p, c_void_p],
    "tagrec_sample_negative_i64": [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, ctypes.c_uint64, c_void_p, c_void_p],
    "tagrec_transtag_fwd_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_float, c_void_p, c_void_p,
                                c_void_p, c_void_p],
    "tagrec_transtag_bwd_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_float, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p],
    "tagrec_spmm_axpy_adam_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_float, c_float, c_float, c_float, c_int64, c_int, c_void_p],
    "tagrec_tall_mm_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_tall_mm_adam_f32": [c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_float, c_float, c_float, c_float, c_int64, c_void_p],
    "tagrec_tall_wgrad_workspace": [c_int, c_int],
    "tagrec_tall_wgrad_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                              c_void_p, c_int64, c_void_p],
    "tagrec_small_mm_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p],
    "tagrec_masked_colsum_workspace": [c_int],
    "tagrec_masked_colsum_f32": [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p],
    "tagrec_row_add_at_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "tagrec_probe_triad_f32": [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_void_p],
    "tagrec_probe_read_f32": [c_void_p, c_int64, c_void_p, c_void_p],
    "tagrec_probe_gather_out_floats": [],
    "tagrec_probe_clock": [c_int64, c_void_p, c_void_p],
    "tagrec_probe_gather_rows_f32": [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p],
    "tagrec_adam_advance": [c_void_p, c_void_p, c_float, c_float, c_float, c_void_p],
    "tagrec_spmm_axpy_adam_graph_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_int, c_void_p],
    "tagrec_sum_n_f32": [c_void_p, c_void_p, c_int, c_int64, c_void_p],
    "tagrec_adam_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int64,
                        c_void_p],
    "tagrec_adam_multi_f32": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int64, c_void_p],
    "tagrec_adam_graph_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                              c_void_p, c_void_p, c_void_p],
}

_lib = None


def load():
    """Load the shared library once; raise TagrecError if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TagrecError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  tagrec_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = header and library disagree
        fn.argtypes = argtypes
        fn.restype = c_int64 if name.endswith(("_workspace", "_result", "_floats")) else c_int
    lib.tagrec_last_error.argtypes = []
    lib.tagrec_last_error.restype = c_char_p
    if lib.tagrec_abi_version() != ABI_VERSION:
        raise TagrecError(f"ABI mismatch: library {lib.tagrec_abi_version()}, host {ABI_VERSION}")
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGNATURES) + ["tagrec_last_error"]


def check(rc, what=""):
    if rc != 0:
        msg = load().tagrec_last_error().decode("utf-8", "replace")
        raise TagrecError(f"{what or 'tagrec call'} failed ({rc}): {msg}")


def stream_ptr():
    """torch's current HIP stream as the void* the ABI takes."""
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return c_void_p(0 if t is None else t.data_ptr())


def require_gpu_tensor(t, dtype, name):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TagrecError(f"{name}: expected a GPU tensor (tagrec_amd has no CPU path), got {type(t).__name__}"
                          f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}")
    if t.dtype != dtype:
        raise TagrecError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise TagrecError(f"{name}: must be contiguous")
    return t
