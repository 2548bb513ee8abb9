"""ctypes binding of libfpn_hip.so (the C ABI declared in include/fpn.h).

The library is the only compute back end of this package: there is no CPU or
eager-PyTorch fallback.  Importing this module without the built library, or
calling an operator with tensors that are not on a ROCm device, raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FPN_LIB_PATH selects another build of the same library (tuning experiments)
LIB_PATH = os.path.abspath(os.environ["FPN_LIB_PATH"]) if os.environ.get("FPN_LIB_PATH") \
    else os.path.join(_HERE, "libfpn_hip.so")

FPN_OK = 0
STATIC_SCRATCH_BYTES = 8 << 20   # FPN_STATIC_SCRATCH_BYTES of include/fpn.h
ABI_VERSION = 18


class FpnError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, code, message):
        super().__init__("libfpn_hip: %s (status %d)" % (message, code))
        self.code = code


_vp = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_long
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> argtypes, exactly the parameter lists of include/fpn.h
SIGNATURES = {
    "fpn_farthest_point_sample": [_i, _i, _i, _vp, _vp, _vp],
    "fpn_gather_point": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "fpn_query_ball_point": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "fpn_query_ball_point_multi": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fpn_group_point": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "fpn_group_concat": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "fpn_three_nn": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "fpn_three_interpolate": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "fpn_three_nn_interpolate": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fpn_three_nn_interpolate_bf16": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "fpn_select_top_k": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "fpn_knn_point": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "fpn_shared_mlp_f32": [_l, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp],
    "fpn_mlp_pack_bf16": [_i, _i, _vp, _vp, _vp],
    "fpn_shared_mlp_bf16": [_l, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp],
    "fpn_shared_mlp_static_f32": [_l, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp],
    "fpn_shared_mlp_flipout_f32": [_l, _i, _i, _vp, _i, _vp, _vp, _l, _vp, _vp, _i, _vp, _vp, _vp, _i, _i,
                                   _vp, _i, _vp, _sz, _vp],
    "fpn_max_pool_rows": [_l, _i, _i, _vp, _i, _vp, _i, _vp],
    "fpn_tile_weights_reg_f32": [_i, _i, _i, _vp, _vp, _vp],
    "fpn_sa_chain_supported": [_i, _i, _i, _i],
    "fpn_sa_chain_bf16_supported": [_i, _i, _i, _i],
    "fpn_sa_chain_overwrites": [_l, _i, _i, _i, _i],
    "fpn_sa_chain_bf16_overwrites": [_l, _i, _i, _i, _i],
    "fpn_sa_chain_f32": [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "fpn_sa_chain_pre_f32": [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "fpn_v1_seg_head_f32": [_l, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp],
    "fpn_v1_seg_tail_f32": [_l, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "fpn_v1_obj_chain_f32": [_l, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "fpn_flipout_signs": [_l, _vp, _vp, _vp],
    "fpn_flipout_weights_bits_f32": [_i, _i, _i, _vp, _vp, _i, _l, _vp, _l, _i, _l, _i, _i, _vp, _vp],
    "fpn_flipout_weights_f32": [_i, _i, _i, _vp, _vp, _i, _l, _vp, _i, _vp, _i, _i, _vp, _vp],
    "fpn_chain_pad_units_bf16": [],
    "fpn_chain_pack_bf16": [_i, _i, _vp, _vp, _vp, _vp],
    "fpn_flipout_stream_bf16": [_l, _i, _i, _vp, _vp, _i, _l, _vp, _i, _vp, _i, _vp, _l, _vp],
    "fpn_flipout_stream_bits_bf16": [_l, _i, _i, _vp, _vp, _i, _l, _vp, _l, _i, _l, _i, _vp, _l, _vp],
    "fpn_v1_seg_head_bf16": [_l, _i, _i, _vp, _vp, _vp, _l, _vp, _vp, _i, _vp],
    "fpn_v1_seg_tail_bf16": [_l, _i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp],
    "fpn_v1_obj_chain_bf16": [_l, _i, _i, _vp, _vp, _vp, _vp, _l, _vp, _i, _vp],
    "fpn_sa_chain_bf16": [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "fpn_sa_chain_pre_bf16": [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "fpn_sa_chain_bf16p": [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "fpn_sa_pack_features_bf16": [_l, _i, _vp, _vp, _vp],
    "fpn_fp_head_chain_f32": [_l, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "fpn_fp_head_chain_pre_f32": [_l, _i, _vp, _vp, _vp, _vp, _vp],
    "fpn_fp_head_chain_pre_bf16": [_l, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "fpn_fp_head_chain_bf16": [_l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "fpn_fp_head_chain_bf16x": [_l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "fpn_mc_statistics": [_i, _i, _vp, _vp, _vp, _vp],
    "fpn_box3d_iou_f64": [_l, _vp, _l, _vp, _l, _vp, _vp],
    "fpn_box3d_intersection_f64": [_l, _vp, _l, _vp, _l, _vp, _vp, _vp],
    "fpn_box3d_best_overlap_f64": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fpn_detection_records": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fpn_mask_gather": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
}

_lib = None


def load():
    """Load the shared library once; raise if it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfpn_hip.so is not built (%s). Build it with `make -C 3d-object-detection_amd/csrc` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`; this package has no "
            "fallback implementation." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.fpn_abi_version.restype = _i
    lib.fpn_last_error_string.restype = ctypes.c_char_p
    lib.fpn_shared_mlp_workspace_bytes.restype = _sz
    lib.fpn_shared_mlp_workspace_bytes.argtypes = [_l, _i, _i]
    lib.fpn_ball_threshold.restype = _f
    lib.fpn_ball_threshold.argtypes = [_f]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _i
        fn.argtypes = argtypes
    if lib.fpn_abi_version() != ABI_VERSION:
        raise ImportError("libfpn_hip.so ABI %d != expected %d: rebuild the library"
                          % (lib.fpn_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status):
    if status != FPN_OK:
        raise FpnError(status, load().fpn_last_error_string().decode("utf-8", "replace"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else _vp(t.data_ptr())


def stream_of(t):
    """The caller's current stream on the tensor's device, as a hipStream_t."""
    return _vp(torch.cuda.current_stream(t.device).cuda_stream)


def require_device_tensor(name, t, dtype, ndim=None, last=None):
    """The argument checks the reference's OpKernels make (rank, last dimension,
    dtype), plus the ones a raw-pointer ABI needs (device, contiguity)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
    if not t.is_cuda:
        raise ValueError("%s must live on a ROCm device (got %s); this package has no CPU path"
                         % (name, t.device))
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have rank %d, got shape %s" % (name, ndim, tuple(t.shape)))
    if last is not None and t.shape[-1] != last:
        raise ValueError("%s must have last dimension %d, got shape %s" % (name, last, tuple(t.shape)))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous (dense row-major)" % name)
    return t


def same_device(*tensors):
    dev = tensors[0].device
    for t in tensors[1:]:
        if t is not None and t.device != dev:
            raise ValueError("all tensors must be on the same device (%s vs %s)" % (dev, t.device))
    return dev
