"""The HIP operators against the REFERENCE's outputs with no intermediary: every *_ref
array of tests/golden/pointset_golden.npz was produced by the reference's own CPU loops
compiled from its sources (query_ball_point.cpp:19-66, interpolate.cpp:21-64,84-104,
selection_sort.cpp:20-63; tests/golden/make_golden.py) and is compared bit for bit with
what libfpn_hip.so computes from the same inputs.  The oracle is not involved."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pointset_golden.npz"))
TN_TAGS = sorted(k[3:-8] for k in GOLD.files if k.startswith("tn_") and k.endswith("_idx_ref"))


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_every_reference_array_is_used():
    """Each *_ref array of the golden file belongs to one of the families tested below."""
    fams = ("qb_", "gp_", "ti_", "tn_", "ss_")
    refs = [k for k in GOLD.files if k.endswith("_ref")]
    assert len(refs) >= 40 and all(k.startswith(fams) for k in refs), refs
    assert len(TN_TAGS) == 11


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "empty"])
def test_ball_query_and_group_on_reference_outputs(device, tag):
    from fpn.tf_ops.tf_grouping import group_point, query_ball_point
    r, ns = GOLD["qb_%s_args" % tag]
    idx, cnt = query_ball_point(float(r), int(ns), dev(GOLD["qb_%s_xyz1" % tag], device),
                                dev(GOLD["qb_%s_xyz2" % tag], device))
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    ref = GOLD["qb_%s_idx_ref" % tag]
    empty = (ref < 0).all(-1)                      # rows the reference never wrote (-1 fill of the generator)
    np.testing.assert_array_equal(idx[~empty], ref[~empty])
    assert (idx[empty] == 0).all() and (cnt[empty] == 0).all() and ((cnt > 0) == ~empty).all()
    if tag == "empty":
        assert empty.any() and (~empty).any()
        return
    # (the generator grouped with the -1 rows of empty balls set to 0, which is what the HIP
    # ball query writes there)
    got = group_point(dev(GOLD["gp_%s_points" % tag], device), dev(np.where(ref < 0, 0, ref), device)).cpu().numpy()
    np.testing.assert_array_equal(got, GOLD["gp_%s_out_ref" % tag])


@pytest.mark.parametrize("tag", TN_TAGS)
def test_three_nn_on_reference_outputs(device, tag):
    from fpn.tf_ops.tf_interpolate import three_nn
    d, i = three_nn(dev(GOLD["tn_%s_xyz1" % tag], device), dev(GOLD["tn_%s_xyz2" % tag], device))
    np.testing.assert_array_equal(i.cpu().numpy(), GOLD["tn_%s_idx_ref" % tag])
    np.testing.assert_array_equal(d.cpu().numpy(), GOLD["tn_%s_dist_ref" % tag])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_three_interpolate_and_fused_on_reference_outputs(device, tag):
    from fpn.tf_ops.tf_interpolate import three_interpolate, three_nn, three_nn_interpolate
    x1, x2, p2 = (dev(GOLD["ti_%s_%s" % (tag, k)], device) for k in ("xyz1", "xyz2", "points"))
    d, i = three_nn(x1, x2)
    np.testing.assert_array_equal(i.cpu().numpy(), GOLD["ti_%s_idx_ref" % tag])
    np.testing.assert_array_equal(d.cpu().numpy(), GOLD["ti_%s_dist_ref" % tag])
    got = three_interpolate(p2, dev(GOLD["ti_%s_idx_ref" % tag], device), dev(GOLD["ti_%s_weight" % tag], device))
    np.testing.assert_array_equal(got.cpu().numpy(), GOLD["ti_%s_out_ref" % tag])
    # the fused kernel computes neighbours, weights and the weighted gather itself
    out, fd, fi = three_nn_interpolate(x1, x2, p2, return_neighbours=True)
    np.testing.assert_array_equal(fi.cpu().numpy(), GOLD["ti_%s_idx_ref" % tag])
    np.testing.assert_array_equal(fd.cpu().numpy(), GOLD["ti_%s_dist_ref" % tag])
    np.testing.assert_array_equal(out.cpu().numpy(), GOLD["ti_%s_out_ref" % tag])


@pytest.mark.parametrize("tag,k", [("kat", 3), ("ties", 7)])
def test_select_top_k_on_reference_outputs(device, tag, k):
    from fpn.tf_ops.tf_grouping import select_top_k
    oi, ov = select_top_k(k, dev(GOLD["ss_%s_dist" % tag], device))
    np.testing.assert_array_equal(oi.cpu().numpy(), GOLD["ss_%s_idx_ref" % tag])
    np.testing.assert_array_equal(ov.cpu().numpy(), GOLD["ss_%s_val_ref" % tag])


KNN = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_golden.npz"))


def test_knn_point_on_the_reference_inputs(device):
    """knn_point (one kernel, no distance matrix) on the inputs of the reference's own
    models/tf_ops/grouping/test_knn.py:5-18 (seed 0, shapes (2,5,3) / (2,2,3)): the indices
    of the three nearest are those of the script's float64 brute-force distances, the values
    are those distances to float32 accuracy, and both equal the float32 matrix pushed through
    the reference's selection_sort_cpu."""
    from fpn.tf_ops.tf_grouping import knn_point
    a = dev(KNN["ref_a_val"].astype(np.float32), device)
    b = dev(KNN["ref_b_val"].astype(np.float32), device)
    val, idx = knn_point(3, a, b)
    np.testing.assert_array_equal(idx.cpu().numpy(), KNN["ref_top3_float64"])
    want = np.take_along_axis(KNN["ref_dist_float64"], KNN["ref_top3_float64"].astype(np.int64), -1)
    np.testing.assert_allclose(val.cpu().numpy(), want, rtol=2e-6, atol=1e-7)
    np.testing.assert_array_equal(idx.cpu().numpy(), KNN["ref_idx_ref"])
    np.testing.assert_array_equal(val.cpu().numpy(), KNN["ref_val_ref"])


@pytest.mark.parametrize("tag", ["ties", "wide", "all"])
def test_knn_point_ties_on_reference_outputs(device, tag):
    """Lattice coordinates (many exactly equal distances; five channels; k = n): the order
    among equal distances is the one the reference's swap sequence leaves."""
    from fpn.tf_ops import tf_grouping
    x1, x2, k = dev(KNN[tag + "_xyz1"], device), dev(KNN[tag + "_xyz2"], device), int(KNN[tag + "_k"])
    for fused in (True, False):
        tf_grouping.KNN_FUSED = fused
        try:
            val, idx = tf_grouping.knn_point(k, x1, x2)
        finally:
            tf_grouping.KNN_FUSED = True
        np.testing.assert_array_equal(idx.cpu().numpy(), KNN[tag + "_idx_ref"], err_msg=str(fused))
        np.testing.assert_array_equal(val.cpu().numpy(), KNN[tag + "_val_ref"], err_msg=str(fused))
