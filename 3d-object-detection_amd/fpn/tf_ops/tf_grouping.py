"""Grouping operators: counterpart of models/tf_ops/grouping/tf_grouping.py."""
import torch

from .. import _lib


def query_ball_point(radius, nsample, xyz1, xyz2):
    """radius float, nsample int, xyz1 (B,N,3) dataset, xyz2 (B,M,3) queries ->
    idx (B,M,nsample) int32, pts_cnt (B,M) int32.  Reference: tf_grouping.py:9-21
    (Python argument order; the op itself takes xyz1, xyz2 first, :21); checks as
    QueryBallPointGpuOp, tf_grouping.cpp:70-86."""
    if not float(radius) > 0:
        raise ValueError("QueryBallPoint expects positive radius")
    if int(nsample) <= 0:
        raise ValueError("QueryBallPoint expects positive nsample")
    _lib.require_device_tensor("xyz1", xyz1, torch.float32, 3, 3)
    _lib.require_device_tensor("xyz2", xyz2, torch.float32, 3, 3)
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("QueryBallPoint expects xyz1 and xyz2 with the same batch size")
    _lib.same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        _lib.check(_lib.load().fpn_query_ball_point(b, n, m, float(radius), int(nsample), _lib.ptr(xyz1),
                                                    _lib.ptr(xyz2), _lib.ptr(idx), _lib.ptr(cnt),
                                                    _lib.stream_of(xyz1)))
    return idx, cnt


def query_ball_point_multi(radius_list, nsample_list, xyz1, xyz2, counts=True):
    """query_ball_point for several (radius, nsample) pairs over the same xyz1 / xyz2 in one
    launch (fpn_query_ball_point_multi; the multi-scale set abstraction of
    models/pointnet_util.py:176-178 asks once per radius): a list of (idx, pts_cnt) pairs,
    each equal to the single call's.  At most 4 pairs per launch (more are split);
    counts=False skips the pts_cnt outputs (None in their place)."""
    import ctypes
    pairs = [(float(r), int(ns)) for r, ns in zip(radius_list, nsample_list)]
    if len(pairs) != len(list(radius_list)) or len(pairs) != len(list(nsample_list)) or not pairs:
        raise ValueError("radius_list and nsample_list must be non-empty and of equal length")
    for r, ns in pairs:
        if not r > 0:
            raise ValueError("QueryBallPoint expects positive radius")
        if ns <= 0:
            raise ValueError("QueryBallPoint expects positive nsample")
    _lib.require_device_tensor("xyz1", xyz1, torch.float32, 3, 3)
    _lib.require_device_tensor("xyz2", xyz2, torch.float32, 3, 3)
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("QueryBallPoint expects xyz1 and xyz2 with the same batch size")
    _lib.same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    out = []
    for lo in range(0, len(pairs), 4):
        part = pairs[lo:lo + 4]
        nr = len(part)
        idx = [torch.empty((b, m, ns), dtype=torch.int32, device=xyz1.device) for _, ns in part]
        cnt = [torch.empty((b, m), dtype=torch.int32, device=xyz1.device) if counts else None for _ in part]
        radii = (ctypes.c_float * nr)(*[r for r, _ in part])
        nsamples = (ctypes.c_int * nr)(*[ns for _, ns in part])
        idx_p = (ctypes.c_void_p * nr)(*[t.data_ptr() for t in idx])
        cnt_p = (ctypes.c_void_p * nr)(*[None if t is None else t.data_ptr() for t in cnt])
        with torch.cuda.device(xyz1.device):
            _lib.check(_lib.load().fpn_query_ball_point_multi(b, n, m, nr, radii, nsamples, _lib.ptr(xyz1), _lib.ptr(xyz2),
                                                              idx_p, cnt_p, _lib.stream_of(xyz1)))
        out += list(zip(idx, cnt))
    return out


def select_top_k(k, dist):
    """k int, dist (B,M,N) float32 -> idx (B,M,N) int32, dist_out (B,M,N) float32;
    the first k columns hold the k smallest of each row.  Reference:
    tf_grouping.py:23-32; checks as SelectionSortGpuOp, tf_grouping.cpp:110-118."""
    if int(k) <= 0:
        raise ValueError("SelectionSort expects positive k")
    _lib.require_device_tensor("dist", dist, torch.float32, 3)
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    with torch.cuda.device(dist.device):
        _lib.check(_lib.load().fpn_select_top_k(b, n, m, int(k), _lib.ptr(dist), _lib.ptr(outi),
                                                _lib.ptr(out), _lib.stream_of(dist)))
    return outi, out


def group_point(points, idx):
    """points (B,N,C) float32, idx (B,M,nsample) int32 -> (B,M,nsample,C) float32.
    Reference: tf_grouping.py:34-42; checks as GroupPointGpuOp,
    tf_grouping.cpp:146-156."""
    _lib.require_device_tensor("points", points, torch.float32, 3)
    _lib.require_device_tensor("idx", idx, torch.int32, 3)
    if idx.shape[0] != points.shape[0]:
        raise ValueError("GroupPoint expects (batch_size, npoints, nsample) idx shape")
    _lib.same_device(points, idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.load().fpn_group_point(b, n, c, m, ns, _lib.ptr(points), _lib.ptr(idx),
                                               _lib.ptr(out), _lib.stream_of(points)))
    return out


# knn_point as one kernel (fpn_knn_point); False: the reference's own composition - the
# (B,M,N) distance matrix formed with tensor operations and handed to select_top_k - which
# the tests keep as the fused kernel's twin
KNN_FUSED = True


def knn_point(k, xyz1, xyz2):
    """k int, xyz1 (B,N,C) dataset, xyz2 (B,M,C) queries -> val (B,M,k) squared
    distances, idx (B,M,k) int32.  Reference: tf_grouping.py:49-74: the (B,M,N)
    distance matrix sum((xyz1-xyz2)**2,-1) goes through select_top_k and the first k
    columns are sliced out; here one kernel forms each query's row on chip, runs the
    same selection steps and writes the k columns (no (B,M,N) array)."""
    if int(k) <= 0:
        raise ValueError("SelectionSort expects positive k")
    _lib.require_device_tensor("xyz1", xyz1, torch.float32, 3)
    _lib.require_device_tensor("xyz2", xyz2, torch.float32, 3)
    if xyz1.shape[0] != xyz2.shape[0] or xyz1.shape[2] != xyz2.shape[2]:
        raise ValueError("knn_point expects xyz1 (B,N,C) and xyz2 (B,M,C) with the same batch size and channels")
    _lib.same_device(xyz1, xyz2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    if int(k) > n:
        raise ValueError("knn_point: k = %d exceeds the %d dataset points" % (int(k), n))
    if not KNN_FUSED:
        diff = xyz1[:, None, :, :] - xyz2[:, :, None, :]
        sq = diff * diff
        dist = sq[..., 0]
        for ch in range(1, c):          # left to right, as the kernel sums
            dist = dist + sq[..., ch]
        outi, out = select_top_k(k, dist.contiguous())
        return out[:, :, :k].contiguous(), outi[:, :, :k].contiguous()
    val = torch.empty((b, m, int(k)), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, m, int(k)), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        _lib.check(_lib.load().fpn_knn_point(b, n, c, m, int(k), _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(val),
                                             _lib.ptr(idx), _lib.stream_of(xyz1)))
    return val, idx
