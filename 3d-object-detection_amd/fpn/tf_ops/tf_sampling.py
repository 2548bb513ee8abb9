"""Sampling operators: counterpart of models/tf_ops/sampling/tf_sampling.py.

Same function names, argument order, shapes and dtypes as the reference
wrappers; tensors are contiguous torch tensors on a ROCm device and the work is
done by libfpn_hip.so on the caller's current stream.
"""
import torch

from .. import _lib


def prob_sample(inp, inpr):
    """Reference: tf_sampling.py:14-22.  Never called by any model of the
    reference (only by its own demo, :61-90); not part of the inference path."""
    raise NotImplementedError("prob_sample is outside the Frustum-PointNet inference path")


def gather_point(inp, idx):
    """inp (B,N,3) float32, idx (B,M) int32 -> (B,M,3) float32.
    Reference: tf_sampling.py:30-39; shape checks as GatherPointGpuOp,
    tf_sampling.cpp:128-135."""
    _lib.require_device_tensor("inp", inp, torch.float32, 3, 3)
    _lib.require_device_tensor("idx", idx, torch.int32, 2)
    if idx.shape[0] != inp.shape[0]:
        raise ValueError("GatherPoint expects (batch_size,num_result) idx shape")
    _lib.same_device(inp, idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
    with torch.cuda.device(inp.device):
        _lib.check(_lib.load().fpn_gather_point(b, n, m, _lib.ptr(inp), _lib.ptr(idx), _lib.ptr(out),
                                                _lib.stream_of(inp)))
    return out


def farthest_point_sample(npoint, inp):
    """npoint int, inp (B,N,3) float32 -> (B,npoint) int32; first index 0, ties to
    the lowest point index.  Reference: tf_sampling.py:49-57; checks as
    FarthestPointSampleGpuOp, tf_sampling.cpp:97-106."""
    if int(npoint) <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")
    _lib.require_device_tensor("inp", inp, torch.float32, 3, 3)
    b, n, _ = inp.shape
    out = torch.empty((b, int(npoint)), dtype=torch.int32, device=inp.device)
    with torch.cuda.device(inp.device):
        _lib.check(_lib.load().fpn_farthest_point_sample(b, n, int(npoint), _lib.ptr(inp), _lib.ptr(out),
                                                         _lib.stream_of(inp)))
    return out
