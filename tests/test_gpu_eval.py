"""GPU parity of the box-overlap kernels and of the mAP module built on them: the
float64 IoU must equal the scalar CPU restatement bit for bit (same operations in the
same order, no fused multiply-add), and the mAP code must reproduce the restated
reference loop (helpers/mAP_evaluation.py:263-373) exactly."""
import math

import numpy as np
import pytest
import torch

from oracle import map_numpy as om

pytestmark = pytest.mark.gpu


def random_boxes(rng, n, spread=3.0, planar=True):
    t = rng.uniform(-spread, spread, (n, 3)) * [1, 1, 0.3]
    s = rng.uniform(0.5, 4.0, (n, 3))
    yaw = rng.uniform(-math.pi, math.pi, n)
    q = np.stack([np.cos(yaw / 2), np.zeros(n), np.zeros(n), np.sin(yaw / 2)], 1)
    if not planar:
        q = q + rng.normal(0, 0.05, (n, 4))          # slight roll / pitch, not normalised
        q *= rng.uniform(0.5, 2.0, (n, 1))
    return np.concatenate([t, s, q], 1)


def test_iou_pairs_bit_exact(device):
    from fpn.helpers import mAP_evaluation as me
    rng = np.random.default_rng(1)
    for planar in (True, False):
        a, b = random_boxes(rng, 600, planar=planar), random_boxes(rng, 600, planar=planar)
        a[:20] = b[:20]                                # identical boxes
        a[20:40, :2] = b[20:40, :2] + 100.0            # far apart
        got = me._iou_pairs(a, b, device)
        want = np.array([om.box_iou(x, y) for x, y in zip(a, b)])
        assert (got > 0).sum() > 100
        np.testing.assert_array_equal(got, want)
        assert np.all(got[20:40] == 0.0)
        if planar:      # with roll / pitch the reference's footprint shrinks, identical boxes then score < 1
            assert np.all(np.abs(got[:20] - 1.0) < 1e-12)
    # one box against many (get_ious)
    one = random_boxes(rng, 1)
    got = me._iou_pairs(one, b, device)
    np.testing.assert_array_equal(got, [om.box_iou(one[0], y) for y in b])


def make_scene(rng, nsamples=40, classes=("car", "pedestrian", "bus")):
    gt, pred = [], []
    for s in range(nsamples):
        token = "sample%03d" % s
        for k in range(rng.integers(0, 7)):
            box = random_boxes(rng, 1, spread=20.0)[0]
            name = classes[rng.integers(0, len(classes))]
            gt.append(dict(sample_token=token, translation=box[:3].tolist(), size=box[3:6].tolist(),
                           rotation=box[6:].tolist(), name=name))
            # 0-2 detections near each ground-truth box, some duplicates with equal scores
            for d in range(rng.integers(0, 3)):
                jit = box.copy()
                jit[:3] += rng.normal(0, 0.3, 3)
                jit[3:6] *= rng.uniform(0.8, 1.25, 3)
                pred.append(dict(sample_token=token, translation=jit[:3].tolist(), size=jit[3:6].tolist(),
                                 rotation=jit[6:].tolist(), name=name,
                                 score=float(np.round(rng.uniform(0, 1), 1))))
        for k in range(rng.integers(0, 3)):            # clutter, also in samples without ground truth
            box = random_boxes(rng, 1, spread=20.0)[0]
            pred.append(dict(sample_token=token if k else "unlabelled%03d" % s, translation=box[:3].tolist(),
                             size=box[3:6].tolist(), rotation=box[6:].tolist(),
                             name=classes[rng.integers(0, len(classes))], score=float(rng.uniform(0, 1))))
    return gt, pred


def test_map_matches_restated_reference_loop(device):
    from fpn.helpers import mAP_evaluation as me
    rng = np.random.default_rng(2)
    gt, pred = make_scene(rng)
    names = me.get_class_names(gt)
    assert names == sorted(set(g["name"] for g in gt))
    for thr in (0.3, 0.5, 0.7):
        got = me.get_average_precisions(gt, pred, names, thr, device)
        want = om.get_average_precisions(gt, pred, names, thr)
        np.testing.assert_array_equal(got, want)
        assert got.max() > 0
    name = names[0]
    g, p = [x for x in gt if x["name"] == name], [x for x in pred if x["name"] == name]
    r1, p1, ap1 = me.recall_precision(g, p, 0.5, device)
    r2, p2, ap2 = om.recall_precision(g, p, 0.5)
    np.testing.assert_array_equal(r1, r2)
    np.testing.assert_array_equal(p1, p2)
    assert ap1 == ap2


def test_box3d_interface(device):
    from fpn.helpers import mAP_evaluation as me
    a = me.Box3D(sample_token="s", translation=[0, 0, 0], size=[2, 4, 2], rotation=[1, 0, 0, 0], name="car")
    b = me.Box3D(sample_token="s", translation=[1, 0, 0], size=[2, 4, 2], rotation=[1, 0, 0, 0], name="car", score=0.5)
    assert a.score == -1 and b.score == 0.5 and a.volume == 16
    np.testing.assert_array_equal(a.ground_bbox_coords, [(2, -1), (2, 1), (-2, 1), (-2, -1)])
    assert a.get_iou(b) == 0.6 and b.get_iou(a) == 0.6
    assert a.get_height_intersection(b) == 2
    assert a.get_intersection(b) == 12.0 and a.get_area_intersection(b) == 6.0      # computed, not derived from the IoU
    # rotated pair, disjoint in height: zero volume but the ground rectangles still overlap;
    # against the numpy restatement of the evaluation
    from oracle import map_numpy as om
    q = [np.cos(0.35), 0, 0, np.sin(0.35)]
    c = me.Box3D(sample_token="s", translation=[0.5, 0.25, 10], size=[1.5, 3, 2], rotation=q, name="car")
    assert a.get_height_intersection(c) == 0 and a.get_intersection(c) == 0 and a.get_iou(c) == 0
    want = om.quad_intersection_area(om.ground_corners(om.pack(a.serialize())), om.ground_corners(om.pack(c.serialize())))
    assert want > 0 and a.get_area_intersection(c) == want
    # a box against itself, and a box wholly inside another, at awkward yaws and sizes: the overlap is
    # the smaller box's own area up to rounding, and the sanity check inside must not trip on that rounding
    rng = np.random.default_rng(5)
    for _ in range(40):
        yaw = rng.uniform(-3.2, 3.2)
        qq = [float(np.cos(yaw / 2)), 0, 0, float(np.sin(yaw / 2))]
        size = list(rng.uniform(0.3, 7.0, 3))
        t = list(rng.uniform(-50, 50, 3))
        inner = me.Box3D(sample_token="s", translation=t, size=size, rotation=qq, name="car")
        outer = me.Box3D(sample_token="s", translation=t, size=[size[0] * 1.5, size[1] * 1.5, size[2]], rotation=qq, name="car")
        own = size[0] * size[1]
        assert abs(inner.get_area_intersection(inner) - own) <= 1e-9 * own
        assert abs(inner.get_area_intersection(outer) - own) <= 1e-9 * own
        assert abs(outer.get_area_intersection(inner) - own) <= 1e-9 * own
        assert abs(inner.get_iou(inner) - 1.0) <= 1e-9
    assert me.get_ious([a, b], b) == [0.6, 1.0]
    assert me.get_ious([], b) == []
    with pytest.raises(AssertionError):
        me.Box3D(sample_token="s", translation=[0, 0, 0], size=[2, 0, 2], rotation=[1, 0, 0, 0], name="car")
    with pytest.raises(AssertionError):
        me.Box3D(sample_token=3, translation=[0, 0, 0], size=[2, 1, 2], rotation=[1, 0, 0, 0], name="car")
    env = me.get_envelope(np.array([0.0, 1.0, 0.5, 2 / 3, 0.5, 0.0]))
    assert env.tolist() == [1.0, 1.0, 2 / 3, 2 / 3, 0.5, 0.0]
    # a prediction in a sample without ground truth is a false positive; no predictions -> AP 0
    gt = [a.serialize()]
    for rec in gt:
        rec.pop("volume"); rec.pop("score")
    pr = [dict(b.serialize(), sample_token="other")]
    for rec in pr:
        rec.pop("volume")
    assert me.get_average_precisions(gt, pr, ["car", "bus"], 0.5, device).tolist() == [0.0, 0.0]
