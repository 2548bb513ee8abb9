"""Interpolation operators: counterpart of
models/tf_ops/3d_interpolation/tf_interpolate.py."""
import torch

from .. import _lib


def three_nn(xyz1, xyz2):
    """xyz1 (B,N,3) unknown points, xyz2 (B,M,3) known points -> dist (B,N,3)
    squared distances ascending, idx (B,N,3) int32.  Reference:
    tf_interpolate.py:9-18; checks as ThreeNNOp, tf_interpolate.cpp:163-168."""
    _lib.require_device_tensor("xyz1", xyz1, torch.float32, 3, 3)
    _lib.require_device_tensor("xyz2", xyz2, torch.float32, 3, 3)
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("ThreeNN expects xyz1 and xyz2 with the same batch size")
    _lib.same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        _lib.check(_lib.load().fpn_three_nn(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist),
                                            _lib.ptr(idx), _lib.stream_of(xyz1)))
    return dist, idx


def three_interpolate(points, idx, weight):
    """points (B,M,C), idx (B,N,3) int32, weight (B,N,3) -> (B,N,C).  Reference:
    tf_interpolate.py:20-29; checks as ThreeInterpolateOp, tf_interpolate.cpp:197-206."""
    _lib.require_device_tensor("points", points, torch.float32, 3)
    _lib.require_device_tensor("idx", idx, torch.int32, 3, 3)
    _lib.require_device_tensor("weight", weight, torch.float32, 3, 3)
    b, m, c = points.shape
    if idx.shape[0] != b:
        raise ValueError("ThreeInterpolate expects (b,n,3) idx shape")
    n = idx.shape[1]
    if weight.shape[0] != b or weight.shape[1] != n:
        raise ValueError("ThreeInterpolate expects (b,n,3) weight shape")
    _lib.same_device(points, idx, weight)
    out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.load().fpn_three_interpolate(b, m, c, n, _lib.ptr(points), _lib.ptr(idx),
                                                     _lib.ptr(weight), _lib.ptr(out),
                                                     _lib.stream_of(points)))
    return out


def three_nn_interpolate(xyz1, xyz2, points2, return_neighbours=False, out_dtype=torch.float32):
    """Fused three_nn -> inverse-distance weights (models/pointnet_util.py:211-214)
    -> three_interpolate.  xyz1 (B,N,3), xyz2 (B,M,3), points2 (B,M,C) -> (B,N,C);
    with return_neighbours also dist (B,N,3) and idx (B,N,3).  Same values as the
    three separate calls; the neighbour table stays on chip unless requested.
    out_dtype=torch.bfloat16: the same sums rounded to bf16 on the way out (for a consumer
    that reads bf16 operands; no neighbour table)."""
    _lib.require_device_tensor("xyz1", xyz1, torch.float32, 3, 3)
    _lib.require_device_tensor("xyz2", xyz2, torch.float32, 3, 3)
    _lib.require_device_tensor("points2", points2, torch.float32, 3)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if xyz2.shape[0] != b or points2.shape[0] != b or points2.shape[1] != m:
        raise ValueError("three_nn_interpolate expects xyz2 (b,m,3) and points2 (b,m,c)")
    if m < 1:
        raise ValueError("three_nn_interpolate expects at least one known point")
    _lib.same_device(xyz1, xyz2, points2)
    c = points2.shape[2]
    if out_dtype == torch.bfloat16:
        if return_neighbours:
            raise ValueError("the bf16 form of three_nn_interpolate returns no neighbour table")
        out = torch.empty((b, n, c), dtype=torch.bfloat16, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            _lib.check(_lib.load().fpn_three_nn_interpolate_bf16(b, n, m, c, _lib.ptr(xyz1), _lib.ptr(xyz2),
                                                                 _lib.ptr(points2), _lib.ptr(out), _lib.stream_of(xyz1)))
        return out
    if out_dtype != torch.float32:
        raise ValueError("out_dtype must be torch.float32 or torch.bfloat16")
    out = torch.empty((b, n, c), dtype=torch.float32, device=xyz1.device)
    dist = idx = None
    if return_neighbours:
        dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
        idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        _lib.check(_lib.load().fpn_three_nn_interpolate(b, n, m, c, _lib.ptr(xyz1), _lib.ptr(xyz2),
                                                        _lib.ptr(points2), _lib.ptr(out), _lib.ptr(dist),
                                                        _lib.ptr(idx), _lib.stream_of(xyz1)))
    if return_neighbours:
        return out, dist, idx
    return out
