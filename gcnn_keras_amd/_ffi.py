"""ctypes binding of libmpengine.so (the C ABI declared in include/mpengine.h).

This is the only place where Python meets the engine: plain pointers and sizes cross the boundary, torch is used
for device memory and streams only.  There is NO CPU fallback: if the shared library is missing or a tensor is
not on a HIP device, the call fails loudly.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmpengine.so")

MP_OK, MP_EINVAL, MP_EINDEX, MP_EHIP, MP_ENOTSUP = 0, -1, -2, -3, -4
MP_SUM, MP_MEAN, MP_MAX, MP_MIN = 0, 1, 2, 3
MP_ADD, MP_SUB, MP_MUL = 0, 1, 2
MP_PAINN_FILTER_IMAGE_BYTES = 73728   # include/mpengine.h
MP_FLAG_OOB, MP_FLAG_UNSORTED_COL0, MP_FLAG_UNSORTED_COL1 = 1, 2, 4

ACTIVATION_CODES = {
    None: 0, "linear": 0, "relu": 1, "kgcnn>shifted_softplus": 2, "shifted_softplus": 2, "softplus": 3,
    "swish": 4, "sigmoid": 5, "tanh": 6, "kgcnn>leaky_relu": 7, "leaky_relu": 7, "kgcnn>softplus2": 8, "softplus2": 8,
    "selu": 9,
}

P = c_void_p
# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/mpengine.h one to one.
_SIGNATURES = {
    "mp_last_error": [],
    "mp_version": [],
    "mp_device_info": [c_char_p, c_int, P, P],
    "mp_graph_begin": [P],
    "mp_graph_end": [P, P],
    "mp_graph_launch": [P, P],
    "mp_graph_destroy": [P],
    "mp_event_create": [P],
    "mp_event_record": [P, P],
    "mp_event_elapsed_ms": [P, P, P],
    "mp_event_destroy": [P],
    "mp_shift_index_i64": [P, c_int64, c_int, P, P, c_int64, c_int, P, P],
    "mp_index_prepare_i64": [P, c_int64, c_int, P, P, c_int64, c_int64, P, P, P],
    "mp_csr_from_sorted_i32": [P, c_int64, c_int64, P, P],
    "mp_sort_workspace_bytes": [c_int64, P],
    "mp_sort_segments_i32": [P, c_int64, P, P, P, c_size_t, P],
    "mp_gather_rows_f32": [P, c_int64, c_int64, P, c_int64, c_int, P, P, P],
    "mp_gather_rows_i64_f32": [P, c_int64, c_int64, P, c_int64, c_int, c_int, P, P, c_int64, P, P],
    "mp_repeat_rows_f32": [P, P, c_int64, c_int64, c_int64, P, P],
    "mp_embedding_f32": [P, c_int64, c_int64, P, c_int64, P, P, P],
    "mp_segment_reduce_csr_f32": [c_int, P, c_int64, c_int64, P, P, c_int64, P, c_int, P, P],
    "mp_gather_segment_reduce_csr_f32": [c_int, P, c_int64, c_int64, P, c_int64, P, P, c_int64, P, c_int, c_int,
                                         c_float, P, P],
    "mp_pool_graph_f32": [c_int, P, P, c_int64, c_int64, P, P, P],
    "mp_segment_softmax_csr_f32": [P, c_int64, c_int64, P, P, c_int64, P, P],
    "mp_scatter_relational_f32": [c_int, P, c_int64, c_int64, P, P, c_int64, c_int64, P, P],
    "mp_dense_f32": [P, c_int64, c_int64, P, P, c_int64, c_int, c_float, P, P],
    "mp_dense_ex_f32": [P, c_int64, c_int64, P, P, c_int64, c_int, c_float, c_int, c_int, c_float, P, P, P, P, P, P],
    "mp_dense_splitk_workspace_bytes": [c_int64, c_int64, c_int, P],
    "mp_dense_splitk_f32": [P, c_int64, c_int64, P, P, c_int64, c_int, c_float, c_int, P, c_size_t, P, P],
    "mp_activation_f32": [c_int, c_float, P, c_int64, P, P],
    "mp_softmax_rows_f32": [P, c_int64, c_int64, P, P],
    "mp_binary_f32": [c_int, P, P, P, P, c_int64, c_int64, c_int64, P, P],
    "mp_copy_cols_f32": [P, c_int64, c_int64, P, c_int64, c_int64, c_int64, c_int64, P],
    "mp_euclidean_norm_f32": [P, c_int64, c_int64, c_int64, c_int, P, P],
    "mp_scalar_product_f32": [P, P, c_int64, c_int64, c_int64, P, P],
    "mp_gauss_basis_f32": [P, c_int64, c_int, c_float, c_float, c_float, P, P],
    "mp_bessel_basis_f32": [P, c_int64, P, c_int, c_float, c_int, P, P],
    "mp_cos_cutoff_f32": [P, c_int64, c_float, P, P],
    "mp_edge_geometry_f32": [P, c_int64, P, P, c_int64, P, P, P],
    "mp_ragged_to_padded_f32": [P, P, c_int64, c_int64, c_int64, P, P, P],
    "mp_edge_prepare_i64_f32": [P, c_int64, P, P, c_int64, c_int64, P, P, P, P, P, P],
    "mp_cfconv_packed_floats": [],
    "mp_cfconv_pack_f32": [P, P, c_int, P, P, P, P],
    "mp_cfconv_fused_f32": [P, c_int64, P, c_int, P, P, P, P, c_int64, c_int, P, P],
    "mp_cfconv_gauss_fused_f32": [P, c_int64, P, c_int, c_float, c_float, c_float, P, P, P, P, c_int64, c_int, P, P],
    "mp_cfconv_det_workspace_bytes": [c_int64, P],
    "mp_cfconv_fused_ws_f32": [P, c_int64, P, c_int, P, P, P, P, c_int64, c_int, P, P, c_size_t, P],
    "mp_cfconv_gauss_fused_ws_f32": [P, c_int64, P, c_int, c_float, c_float, c_float, P, P, P, P, c_int64, c_int, P, P,
                                     c_size_t, P],
    "mp_cfconv_bwd_packed_floats": [],
    "mp_cfconv_bwd_pack_f32": [P, P, c_int, P, P, P],
    "mp_cfconv_gauss_dist_grad_f32": [P, P, c_int64, P, c_int, c_float, c_float, c_float, P, P, P, c_int64, c_int, P, P],
    "mp_cfconv_gauss_diag_f32": [P, c_int64, P, c_int, c_float, c_float, c_float, P, P, P, P, c_int64, P, P, P],
    "mp_painn_message_fused_f32": [P, P, c_int64, P, c_int, P, P, P, P, P, P, P, c_int64, P, P, P],
    "mp_lstm_zero_state_f32": [P, c_int64, c_int64, c_int, c_int, P, P],
    "mp_gru_combine_f32": [P, P, P, c_int64, c_int64, c_int, c_int, P, P],
    "mp_batched_matvec_f32": [P, P, c_int64, c_int64, c_int64, P, P],
    "mp_painn_stage0_f32": [P, c_int, c_int64, P, c_int, c_float, P, P, P, c_int64, P, P, c_int64, P, P, c_int, c_float, c_int,
                            c_float, P, P, P, P, P, P, P, P, P, P],
    "mp_painn_message_f32": [P, P, c_int64, P, c_int, P, P, P, P, P, P, P, c_int64, P, P, P, P],
    "mp_painn_filter_pack_f32": [P, P, c_int, P, P],
    "mp_painn_message_tiles_lds_bytes": [c_int, c_int, c_int, c_int, P],
    "mp_painn_message_tiles_f32": [P, P, c_int64, P, c_int, P, P, P, P, P, c_int64, P, c_int, c_int, c_int, P, P, P, P],
    "mp_painn_message_bwd_f32": [P, P, c_int64, P, P, c_int, P, P, P, P, P, P, P, P, c_int64, P, P, P, P, P, P, c_int, P],
    "mp_painn_message_bwd_tiles_lds_bytes": [c_int, c_int, c_int, c_int, c_int, P],
    "mp_painn_message_bwd_tiles_f32": [P, P, c_int64, P, P, c_int, P, P, P, P, P, P, P, c_int64, P, c_int, c_int, c_int,
                                       c_int, P, P, P, P, P, P, c_int, P],
    "mp_painn_update_pre_f32": [P, P, c_int64, P, P, P],
    "mp_painn_update_fused_f32": [P, P, P, c_int64, P, P, c_int, c_float, P, P, P, P, P, P, P, P, P],
    "mp_painn_update_fused_bwd_f32": [P, P, P, P, P, P, c_int64, P, c_int, c_float, P, P, P, P, P],
    "mp_painn_update_post_f32": [P, P, P, P, P, c_int64, P, P, P],
    "mp_painn_update_post_bwd_f32": [P, P, P, P, P, c_int64, P, P, P],
    "mp_painn_update_pre_bwd_f32": [P, P, P, P, P, P, P, c_int64, P, P, P],
    "mp_edge_geometry_bwd_f32": [P, P, c_int, P, P, P, P, P, P, c_int64, c_int64, c_float, P, P],
    "mp_schnet_node_pack_f32": [P, c_int, c_int, P, P],
    "mp_schnet_node_pack_bf16_f32": [P, c_int, c_int, P, P],
    "mp_schnet_node_residual_f32": [P, c_int64, P, P, P, P, P, P, c_int, P],
    "mp_schnet_node_in_f32": [P, c_int64, P, c_int, c_int, P, P, P, P, P, c_int, P],
    "mp_schnet_stage0_f32": [P, c_int64, P, c_int, c_int, P, P, P, P, P, P, c_int64, P, P, c_int64, P, P, P, P, P, c_int,
                             P],
    "mp_schnet_node_update_f32": [P, c_int64, P, P, P, P, P, P, P, c_int, P],
    "mp_schnet_node_last_f32": [P, c_int64, P, P, P, P, P, P, P, P, P, P, c_int, P],
    "mp_schnet_readout_f32": [P, P, c_int64, P, P, P, P, P, P],
    "mp_radius_graph_workspace_bytes": [c_int64, P],
    "mp_radius_graph_count_f32": [P, P, c_int64, c_int64, c_float, c_int, P, P, P, c_size_t, P],
    "mp_radius_graph_fill_f32": [P, P, c_int64, c_int64, c_float, c_int, P, c_int64, P, P, P, P, P],
    "mp_activation_grad_f32": [c_int, c_float, P, P, c_int64, P, P],
    "mp_sum_axis_f32": [P, c_int64, c_int64, c_int64, c_int, P, P],
    "mp_euclidean_norm_grad_f32": [P, P, c_int64, c_int64, c_int64, c_int, P, P],
    "mp_bessel_basis_grad_f32": [P, c_int64, P, c_int, c_float, c_int, P, P, P],
    "mp_gauss_basis_grad_f32": [P, c_int64, c_int, c_float, c_float, c_float, P, P, P],
    "mp_cos_cutoff_grad_f32": [P, c_int64, c_float, P, P, P],
    "mp_layer_norm_f32": [P, c_int64, c_int64, P, P, c_float, P, P],
    "mp_schnet_forward_launch": [P, P],
    "mp_schnet_node_update_save_f32": [P, c_int64, P, P, P, P, P, P, P, P, c_int, P],
    "mp_schnet_node_last_save_f32": [P, c_int64, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, P],
    "mp_schnet_readout_grad_f32": [P, P, c_int64, P, P, P, P, P, P, P],
    "mp_schnet_bwd_head_f32": [P, P, P, c_int64, P, P, P, P, P, P, P, P, P],
    "mp_schnet_bwd_block_f32": [P, c_int64, P, P, P, P, P, P, P],
    "mp_schnet_force_from_gd_f32": [P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_float, P, P],
    "mp_schnet_force_launch": [P, P],
    "mp_pool_mlp2_f32": [P, P, c_int64, c_int, P, P, c_int, c_int, c_float, P, P, P, P, P],
    "mp_chain_supported": [c_int, c_int, c_int],
    "mp_chain_pack_f32": [P, c_int, c_int, P, P],
    "mp_dense_chain_f32": [P, c_int64, c_int, P, P, c_int, c_int, c_float, P, P, P, P, c_int, P, P, P],
    "mp_pack_rows_host": [P, P, c_int64, c_int64, c_int, c_int, P, P, c_int],
    "mp_pack_edge_index_host": [P, c_int, P, P, c_int64, c_int, P, P, P, P, P, P, c_int],
    "mp_host_alloc": [c_size_t, c_int, P],
    "mp_host_free": [P, c_int],
    "mp_memcpy_h2d_async": [P, P, c_size_t, P],
    "mp_gcn_tile_f32": [P, P],
    "mp_concat_batches": [P, P],
}
_RESTYPES = {"mp_last_error": c_char_p}

MP_SCHNET_MAX_DEPTH = 8


class SchnetForwardDesc(ctypes.Structure):
    """``mp_schnet_forward_desc`` of include/mpengine.h (field for field)."""
    _fields_ = ([("N", c_int64), ("M", c_int64), ("G", c_int64),
                 ("depth", ctypes.c_int32), ("vocab", ctypes.c_int32), ("flags", ctypes.c_int32),
                 ("bins", ctypes.c_int32),
                 ("g_distance", c_float), ("g_sigma", c_float), ("g_offset", c_float), ("emb_dim", ctypes.c_int32)]
                + [(name, c_void_p) for name in ("numbers", "xyz", "idx", "node_splits", "edge_splits", "embedding",
                                                 "W0", "b0")]
                + [(name, c_void_p * MP_SCHNET_MAX_DEPTH) for name in ("Wx", "packed", "W2", "b2", "W3", "b3")]
                + [(name, c_void_p) for name in ("Wl0", "bl0", "Wl1", "bl1", "Wo0", "bo0", "Wo1", "bo1", "recv", "send",
                                                 "dist", "flags_word", "n", "x", "agg", "h", "out")])


class SchnetForceDesc(ctypes.Structure):
    """``mp_schnet_force_desc`` of include/mpengine.h (field for field)."""
    _fields_ = ([("fwd", SchnetForwardDesc)]
                + [(name, c_void_p) for name in ("xs", "d2", "dl0", "dl1", "g_pool", "node_graph")]
                + [(name, c_void_p * MP_SCHNET_MAX_DEPTH) for name in ("W3T", "W2T", "WxT", "packed_bwd")]
                + [(name, c_void_p) for name in ("Wl0T", "Wl1T", "seg0", "perm0", "seg1", "perm1", "ptr0", "ptr1",
                                                 "g_n", "g_agg", "g_x", "g_d", "force")]
                + [("force_scale", c_float)])


MP_CONCAT_MAX = 8


class BatchSrc(ctypes.Structure):
    """``mp_batch_src`` of include/mpengine.h."""
    _fields_ = [(name, c_void_p) for name in ("z", "xyz", "idx", "node_splits", "edge_splits")] + \
               [("N", c_int64), ("M", c_int64), ("G", c_int64)]


class ConcatDesc(ctypes.Structure):
    """``mp_concat_desc`` of include/mpengine.h (field for field)."""
    _fields_ = [("k", ctypes.c_int32), ("z_is_i64", ctypes.c_int32), ("src", BatchSrc * MP_CONCAT_MAX)] + \
               [(name, c_void_p) for name in ("z", "xyz", "idx", "node_splits", "edge_splits")]


class GcnLayerDesc(ctypes.Structure):
    """``mp_gcn_layer`` of include/mpengine.h."""
    _fields_ = [("W", c_void_p), ("b", c_void_p), ("units", ctypes.c_int32), ("act", ctypes.c_int32), ("alpha", c_float)]


class GcnTileDesc(ctypes.Structure):
    """``mp_gcn_tile_desc`` of include/mpengine.h (field for field)."""
    _fields_ = [("N", c_int64), ("x", c_void_p), ("K", c_int64), ("W_in", c_void_p), ("b_in", c_void_p),
                ("h", c_void_p), ("ptr", c_void_p), ("perm", c_void_p), ("send", c_void_p), ("weight", c_void_p),
                ("M", c_int64), ("agg_act", ctypes.c_int32), ("agg_alpha", c_float),
                ("units_in", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("layer", GcnLayerDesc * 3),
                ("softmax_last", ctypes.c_int32), ("out", c_void_p), ("tile_start", c_void_p), ("n_tiles", c_int64)]


_lib = None


class EngineError(RuntimeError):
    pass


def declared_symbols():
    """Names of every entry point declared in include/mpengine.h (checked by tests/test_abi.py)."""
    return sorted(_SIGNATURES)


def lib():
    """Load libmpengine.so once.  Raises if it was not built (run ``python -c 'import __graft_entry__ as g; g.build()'``).
    ``MPENGINE_LIB`` names another build of the same ABI - used by ``make -C gcnn_keras_amd/csrc asan``, whose
    host-only AddressSanitizer build exports the host packer and runtime entry points only (missing kernels are then
    simply not bound; calling one raises AttributeError)."""
    global _lib
    if _lib is None:
        path = os.environ.get("MPENGINE_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise EngineError("libmpengine.so not found at %s - build it with __graft_entry__.build() "
                              "(there is no CPU fallback)" % path)
        handle = ctypes.CDLL(path)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                if path == LIB_PATH:
                    raise EngineError("libmpengine.so lacks %s - rebuild it (__graft_entry__.build())" % name)
                continue
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
        _lib = handle
    return _lib


def check(rc):
    if rc == MP_OK:
        return
    msg = lib().mp_last_error().decode("utf-8", "replace")
    if rc == MP_EINVAL:
        raise ValueError(msg)
    if rc == MP_EINDEX:
        raise IndexError(msg)
    raise EngineError("libmpengine status %d: %s" % (rc, msg))


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise EngineError("gcnn_keras_amd ops run on an MI355X only: got a %s tensor (no CPU fallback)" % t.device)


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("engine buffers must be contiguous")
    return c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_handle():
    """Integer hipStream_t of torch's current stream.  ``torch.cuda.current_stream().cuda_stream`` builds a Stream object
    and resolves the device index through several Python layers (~5 us, three times per model call before); the two C
    entry points underneath it take ~0.3 us."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def stream():
    """hipStream_t of torch's current stream (torch is plumbing: memory + streams)."""
    return c_void_p(stream_handle())


_has_gpu = None


def has_gpu():
    """``torch.cuda.is_available()`` once per process (the call reads environment variables every time)."""
    global _has_gpu
    if _has_gpu is None:
        _has_gpu = bool(torch.cuda.is_available())
    return _has_gpu


_launches = [0]


def launch_count():
    """Number of engine calls issued so far by this process (bench.py reports calls per forward)."""
    return _launches[0]


_roctx = None


def enable_roctx(on=True):
    """Bracket every engine call in a roctx range named after the entry point (``rocprofv3 --marker-trace`` then shows
    which C-ABI call a kernel belongs to).  Also switched on by ``MPENGINE_ROCTX=1``.  Off by default: two extra
    library calls per launch."""
    global _roctx
    if not on:
        _roctx = None
        return
    for cand in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so",
                 "/opt/rocm/lib/librocprofiler-sdk-roctx.so"):
        try:
            h = ctypes.CDLL(cand)
            h.roctxRangePushA.argtypes = [c_char_p]
            h.roctxRangePushA.restype = c_int
            h.roctxRangePop.restype = c_int
            _roctx = h
            return
        except (OSError, AttributeError):
            continue
    raise EngineError("no roctx library found under /opt/rocm/lib")


def call(name, *args):
    _launches[0] += 1
    if _roctx is not None:
        _roctx.roctxRangePushA(name.encode())
        try:
            check(getattr(lib(), name)(*args))
        finally:
            _roctx.roctxRangePop()
        return
    check(getattr(lib(), name)(*args))


if os.environ.get("MPENGINE_ROCTX") == "1":
    try:
        enable_roctx()
    except EngineError:
        pass


def activation_code(name):
    if isinstance(name, dict):
        name = name.get("class_name", name.get("name"))
    if name not in ACTIVATION_CODES:
        raise ValueError("Activation %r is not supported by the engine" % (name,))
    return ACTIVATION_CODES[name]


def int64_array(values):
    return (c_int64 * len(values))(*values)


def int32_array(values):
    return (ctypes.c_int32 * len(values))(*values)
