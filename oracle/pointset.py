"""numpy front-end to the CPU oracle for the point-set operators.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

Each function takes/returns C-contiguous numpy arrays with the reference's dtypes
(float32 / int32) and shapes; the arithmetic lives in oracle/fpn_oracle.c, whose
functions cite the reference lines they restate.  `ref_*` functions call the
reference's own CPU loops compiled into oracle/_ref/libfpn_ref.so (present only
where oracle/Makefile's `ref` target was built).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "_build", "libfpn_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libfpn_ref.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(with_ref=None):
    """Compile the oracle (and the reference shim when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if with_ref is None:
        with_ref = os.path.isdir("/root/reference/models/tf_ops")
    if with_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _load(path):
    return ctypes.CDLL(path)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_ORACLE_SO):
            build(with_ref=False)
        _lib = _load(_ORACLE_SO)
    return _lib


def have_ref():
    return os.path.exists(_REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = _load(_REF_SO)
    return _ref


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def farthest_point_sample(npoint, inp, lane_emulation=False):
    inp, pin = _f(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    tmp = np.empty(max(n, 1), np.float32)
    lib().oracle_farthest_point_sample(b, n, npoint, pin, tmp.ctypes.data_as(_f32p),
                                       out.ctypes.data_as(_i32p), int(bool(lane_emulation)))
    return out


def gather_point(inp, idx):
    inp, pin = _f(inp)
    idx, pidx = _i(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.empty((b, m, 3), np.float32)
    lib().oracle_gather_point(b, n, m, pin, pidx, out.ctypes.data_as(_f32p))
    return out


def query_ball_point(radius, nsample, xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    cnt = np.empty((b, m), np.int32)
    lib().oracle_query_ball_point(b, n, m, ctypes.c_float(radius), nsample, p1, p2,
                                  idx.ctypes.data_as(_i32p), cnt.ctypes.data_as(_i32p))
    return idx, cnt


def group_point(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    lib().oracle_group_point(b, n, c, m, ns, pp, pi, out.ctypes.data_as(_f32p))
    return out


def three_nn(xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    with np.errstate(over="ignore"):
        lib().oracle_three_nn(b, n, m, p1, p2, dist.ctypes.data_as(_f32p),
                              idx.ctypes.data_as(_i32p))
    return dist, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().oracle_three_interpolate(b, m, c, n, pp, pi, pw, out.ctypes.data_as(_f32p))
    return out


def fp_weights(dist):
    dist, pd = _f(dist)
    w = np.empty_like(dist)
    lib().oracle_fp_weights(ctypes.c_size_t(dist.size // 3), pd, w.ctypes.data_as(_f32p))
    return w


def select_top_k(k, dist):
    dist, pd = _f(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), np.int32)
    out = np.empty((b, m, n), np.float32)
    lib().oracle_selection_sort(b, n, m, k, pd, outi.ctypes.data_as(_i32p),
                                out.ctypes.data_as(_f32p))
    return outi, out


def knn_distances(xyz1, xyz2):
    """The (b,m,n) float32 matrix of knn_point, models/tf_ops/grouping/tf_grouping.py:65-67:
    sum over the channels of (xyz1 - xyz2)**2, every product and sum rounded to float32,
    channels added left to right."""
    xyz1 = np.asarray(xyz1, np.float32)
    xyz2 = np.asarray(xyz2, np.float32)
    sq = (xyz1[:, None, :, :] - xyz2[:, :, None, :]) ** 2
    dist = sq[..., 0]
    for ch in range(1, sq.shape[-1]):
        dist = (dist + sq[..., ch]).astype(np.float32)
    return np.ascontiguousarray(dist)


def knn_point(k, xyz1, xyz2):
    """knn_point, tf_grouping.py:49-74: the matrix above through the selection sort, first k
    columns (:69-71) -> val (b,m,k), idx (b,m,k)."""
    outi, out = select_top_k(k, knn_distances(xyz1, xyz2))
    return np.ascontiguousarray(out[:, :, :k]), np.ascontiguousarray(outi[:, :, :k])


# ---- reference-compiled loops (oracle/_ref) -------------------------------

def ref_query_ball_point(radius, nsample, xyz1, xyz2):
    """query_ball_point_cpu of the reference; rows of empty balls keep the -1 fill."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.full((b, m, nsample), -1, np.int32)
    ref().ref_query_ball_point_cpu(b, n, m, ctypes.c_float(radius), nsample, p1, p2,
                                   idx.ctypes.data_as(_i32p))
    return idx


def ref_group_point(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    ref().ref_group_point_cpu(b, n, c, m, ns, pp, pi, out.ctypes.data_as(_f32p))
    return out


def ref_three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    ref().ref_interpolate_cpu(b, m, c, n, pp, pi, pw, out.ctypes.data_as(_f32p))
    return out


def ref_three_nn_origin(xyz2, n=1):
    """threenn_cpu of the reference's interpolate.cpp (:21-64) as it is: the three known
    points of xyz2 (b,m,3) nearest to the ORIGIN (its distance ignores the queries), for n
    queries per batch."""
    xyz2, p2 = _f(xyz2)
    b, m, _ = xyz2.shape
    xyz1 = np.zeros((b, n, 3), np.float32)
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    with np.errstate(over="ignore"):
        ref().ref_threenn_origin_cpu(b, n, m, xyz1.ctypes.data_as(_f32p), p2, dist.ctypes.data_as(_f32p),
                                     idx.ctypes.data_as(_i32p))
    return dist, idx


def ref_three_nn(xyz1, xyz2):
    """three_nn(xyz1, xyz2) computed by the reference's compiled threenn_cpu: for every
    query the known set is translated by the query in fp32 (x2' = x2 - x1, one IEEE
    subtraction per coordinate, done here by numpy), after which the reference's
    x2'*x2'+y2'*y2'+z2'*z2' IS tf_interpolate.cpp:73's (x2-x1)*(x2-x1)+... and the rest of
    the loop (double compare, 1e40 start, strict-< cascade) is the reference's own."""
    xyz1 = np.ascontiguousarray(xyz1, np.float32)
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    for i in range(b):
        shifted = (xyz2[i][None, :, :] - xyz1[i][:, None, :]).astype(np.float32)   # (n,m,3): one "batch" per query
        d, k = ref_three_nn_origin(shifted, 1)
        dist[i], idx[i] = d[:, 0], k[:, 0]
    return dist, idx


def ref_select_top_k(k, dist):
    dist, pd = _f(dist)
    b, m, n = dist.shape
    outi = np.zeros((b, m, n), np.int32)
    out = np.zeros((b, m, n), np.float32)
    ref().ref_selection_sort_cpu(b, n, m, k, pd, outi.ctypes.data_as(_i32p),
                                 out.ctypes.data_as(_f32p))
    return outi, out
