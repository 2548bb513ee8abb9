"""Pins the CPU oracle (oracle/fpn_oracle.c) before anything is compared with it:
  * against the reference's one known-answer case (selection_sort.cpp:65-94);
  * against the golden vectors in tests/golden/pointset_golden.npz, whose *_ref
    arrays were produced by the reference's own CPU loops compiled from
    /root/reference (tests/golden/make_golden.py);
  * directly against those compiled loops on fresh random inputs wherever
    oracle/_ref/libfpn_ref.so is present.
three_nn is pinned by the reference's stand-alone threenn_cpu (interpolate.cpp:21-64,
whose distance ignores the queries): run as it is on origin queries, and on known sets
translated by each query in fp32, where its distance is tf_interpolate.cpp:73's bit for
bit (oracle/pointset.ref_three_nn).  FPS and gather_point have no CPU code in the
reference (parity unpinned by reference tests); for them the golden file pins the
restatement against regressions and the structural properties below are checked."""
import os

import numpy as np
import pytest

from oracle import pointset as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pointset_golden.npz"))


def test_selection_sort_known_answer():
    # dist[i] = 10 - i, b=2, n=4, m=2, k=3  (selection_sort.cpp:68,75-77)
    dist = (10 - np.arange(16, dtype=np.float32)).reshape(2, 2, 4)
    idx, val = O.select_top_k(3, dist)
    assert (idx == np.array([3, 2, 1, 0])).all()
    np.testing.assert_array_equal(val.reshape(4, 4), [[7, 8, 9, 10], [3, 4, 5, 6], [-1, 0, 1, 2], [-5, -4, -3, -2]])
    np.testing.assert_array_equal(idx, GOLD["ss_kat_idx_ref"])
    np.testing.assert_array_equal(val, GOLD["ss_kat_val_ref"])


def test_selection_sort_golden_ties():
    idx, val = O.select_top_k(7, GOLD["ss_ties_dist"])
    np.testing.assert_array_equal(idx, GOLD["ss_ties_idx_ref"])
    np.testing.assert_array_equal(val, GOLD["ss_ties_val_ref"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "empty"])
def test_ball_query_and_group_golden(tag):
    r, ns = GOLD["qb_%s_args" % tag]
    idx, cnt = O.query_ball_point(float(r), int(ns), GOLD["qb_%s_xyz1" % tag], GOLD["qb_%s_xyz2" % tag])
    ref = GOLD["qb_%s_idx_ref" % tag]
    empty = (ref < 0).all(-1)                      # rows the reference never wrote
    np.testing.assert_array_equal(idx[~empty], ref[~empty])
    assert (idx[empty] == 0).all() and (cnt[empty] == 0).all()
    assert ((cnt > 0) == ~empty).all()
    if tag == "empty":
        assert empty.any() and (~empty).any()
        return
    np.testing.assert_array_equal(O.group_point(GOLD["gp_%s_points" % tag], idx), GOLD["gp_%s_out_ref" % tag])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_three_nn_interpolate_golden(tag):
    d, i = O.three_nn(GOLD["ti_%s_xyz1" % tag], GOLD["ti_%s_xyz2" % tag])
    np.testing.assert_array_equal(i, GOLD["ti_%s_idx_ref" % tag])
    np.testing.assert_array_equal(d, GOLD["ti_%s_dist_ref" % tag])
    w = O.fp_weights(d)
    np.testing.assert_array_equal(w, GOLD["ti_%s_weight" % tag])
    np.testing.assert_array_equal(O.three_interpolate(GOLD["ti_%s_points" % tag], i, w), GOLD["ti_%s_out_ref" % tag])


TN_TAGS = ["origin_a", "origin_b", "origin_c", "origin_ties", "m1", "m2", "m5", "m128", "ties", "subset", "metres"]


@pytest.mark.parametrize("tag", TN_TAGS)
def test_three_nn_golden_reference_outputs(tag):
    d, i = O.three_nn(GOLD["tn_%s_xyz1" % tag], GOLD["tn_%s_xyz2" % tag])
    np.testing.assert_array_equal(i, GOLD["tn_%s_idx_ref" % tag])
    np.testing.assert_array_equal(d, GOLD["tn_%s_dist_ref" % tag])
    m = GOLD["tn_%s_xyz2" % tag].shape[1]
    if m < 3:                                      # (float)1e40 = +inf, index 0 (interpolate.cpp:27-28)
        assert np.isinf(d[..., m:]).all() and (i[..., m:] == 0).all()


def test_three_nn_is_true_nearest():
    rng = np.random.RandomState(3)
    x1 = rng.random_sample((2, 50, 3)).astype(np.float32)
    x2 = rng.random_sample((2, 20, 3)).astype(np.float32)
    d, i = O.three_nn(x1, x2)
    full = ((x1[:, :, None, :].astype(np.float64) - x2[:, None, :, :]) ** 2).sum(-1)
    np.testing.assert_array_equal(i, np.argsort(full, -1, kind="stable")[..., :3])
    np.testing.assert_allclose(d, np.sort(full, -1)[..., :3], rtol=1e-5)
    # fewer than three known points: +inf distance, index 0 (tf_interpolate.cpp:66-67)
    d, i = O.three_nn(x1, x2[:, :2])
    assert np.isinf(d[..., 2]).all() and (i[..., 2] == 0).all()
    w = O.fp_weights(d)
    assert (w[..., 2] == 0).all() and np.allclose(w.sum(-1), 1)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_fps_golden_and_properties(tag):
    x = GOLD["fps_%s_xyz" % tag]
    m = GOLD["fps_%s_idx_restated" % tag].shape[1]
    idx = O.farthest_point_sample(m, x)
    np.testing.assert_array_equal(idx, GOLD["fps_%s_idx_restated" % tag])
    np.testing.assert_array_equal(O.farthest_point_sample(m, x, lane_emulation=True),
                                  GOLD["fps_%s_idx_lane512_restated" % tag])
    np.testing.assert_array_equal(O.gather_point(x, idx), GOLD["fps_%s_gather_restated" % tag])
    # greedy property, recomputed independently in float64-free numpy fp32
    for b in range(x.shape[0]):
        md = np.full(x.shape[1], np.float32(1e38))
        assert idx[b, 0] == 0
        for j in range(1, m):
            p = x[b, idx[b, j - 1]]
            dx, dy, dz = (x[b, :, 0] - p[0]), (x[b, :, 1] - p[1]), (x[b, :, 2] - p[2])
            d = (dx * dx + dy * dy) + dz * dz
            md = np.minimum(md, d)
            assert idx[b, j] == int(np.argmax(md))          # argmax = lowest index on ties


def test_fps_tie_rules_differ_only_on_exact_ties():
    # n <= 512: the reference's 512-lane order IS lowest-index
    rng = np.random.RandomState(1)
    x = rng.randint(0, 4, (2, 300, 3)).astype(np.float32)       # lattice: many exact ties
    np.testing.assert_array_equal(O.farthest_point_sample(40, x), O.farthest_point_sample(40, x, True))
    # n > 512 with an exact tie between k=100 (lane 100) and k=520 (lane 520%512 = 8):
    # this build takes the lowest index, the reference's block the lowest lane
    x = np.zeros((1, 1024, 3), np.float32)
    x[0, 100] = [0, 1, 0]
    x[0, 520] = [0, 0, 1]
    assert O.farthest_point_sample(2, x)[0, 1] == 100
    assert O.farthest_point_sample(2, x, lane_emulation=True)[0, 1] == 520


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_against_reference_loops_random():
    rng = np.random.RandomState(77)
    for (b, n, m, r, ns, c) in [(3, 200, 40, 0.25, 16, 5), (2, 777, 64, 0.12, 48, 3), (1, 64, 64, 0.5, 64, 1)]:
        x1 = rng.random_sample((b, n, 3)).astype(np.float32)
        x2 = x1[:, rng.permutation(n)[:m]].copy()             # queries are dataset points: no empty ball
        idx, cnt = O.query_ball_point(r, ns, x1, x2)
        np.testing.assert_array_equal(idx, O.ref_query_ball_point(r, ns, x1, x2))
        assert (cnt >= 1).all()
        pts = rng.standard_normal((b, n, c)).astype(np.float32)
        np.testing.assert_array_equal(O.group_point(pts, idx), O.ref_group_point(pts, idx))
        d, i = O.three_nn(x1, x2)
        rd, ri = O.ref_three_nn(x1, x2)
        np.testing.assert_array_equal(i, ri)
        np.testing.assert_array_equal(d, rd)
        rd, ri = O.ref_three_nn_origin(x2, 2)
        d0, i0 = O.three_nn(np.zeros((b, 2, 3), np.float32), x2)
        np.testing.assert_array_equal(i0, ri)
        np.testing.assert_array_equal(d0, rd)
        w = O.fp_weights(d)
        p2 = rng.standard_normal((b, m, c)).astype(np.float32)
        np.testing.assert_array_equal(O.three_interpolate(p2, i, w), O.ref_three_interpolate(p2, i, w))
        dist = rng.randint(0, 30, (b, 4, n)).astype(np.float32)
        for k in (1, 5, 32):
            ri, rv = O.ref_select_top_k(k, dist)
            oi, ov = O.select_top_k(k, dist)
            np.testing.assert_array_equal(oi, ri)
            np.testing.assert_array_equal(ov, rv)


def test_knn_point_oracle_on_the_reference_inputs():
    """The oracle's knn_point on the inputs of models/tf_ops/grouping/test_knn.py:5-18 and on
    the lattice sets of tests/golden/knn_golden.npz (whose *_ref outputs come from the
    reference's selection_sort_cpu): same indices as the script's float64 distances give,
    same bits as the reference's sort."""
    knn = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_golden.npz"))
    # the fixture's inputs ARE the script's: numpy's legacy generator, seed 0, two draws
    rs = np.random.RandomState(0)
    np.testing.assert_array_equal(knn["ref_a_val"], rs.random_sample((2, 5, 3)))
    np.testing.assert_array_equal(knn["ref_b_val"], rs.random_sample((2, 2, 3)))
    val, idx = O.knn_point(3, knn["ref_a_val"].astype(np.float32), knn["ref_b_val"].astype(np.float32))
    np.testing.assert_array_equal(idx, knn["ref_top3_float64"])
    np.testing.assert_array_equal(idx, knn["ref_idx_ref"])
    np.testing.assert_array_equal(val, knn["ref_val_ref"])
    for tag in ("ties", "wide", "all"):
        val, idx = O.knn_point(int(knn[tag + "_k"]), knn[tag + "_xyz1"], knn[tag + "_xyz2"])
        np.testing.assert_array_equal(idx, knn[tag + "_idx_ref"], err_msg=tag)
        np.testing.assert_array_equal(val, knn[tag + "_val_ref"], err_msg=tag)
    if O.have_ref():
        oi, ov = O.ref_select_top_k(16, O.knn_distances(knn["ties_xyz1"], knn["ties_xyz2"]))
        np.testing.assert_array_equal(oi[..., :16], knn["ties_idx_ref"])
