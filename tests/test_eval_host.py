"""CPU checks of the evaluation side: the overlap / AP oracle against hand-computed
cases (the reference holds no fixtures for helpers/mAP_evaluation.py and its shapely /
pyquaternion dependencies are not installed, so exact agreement with it is parity
unpinned; these cases have answers known in closed form), and the box-decode helpers
of scripts/provider.py and scripts/test.py:write_detection_results."""
import math
import os

import numpy as np
import pytest

from oracle import map_numpy as om

Q0 = [1.0, 0.0, 0.0, 0.0]


def yaw_quat(t):
    return [math.cos(t / 2), 0.0, 0.0, math.sin(t / 2)]


def test_iou_closed_form_cases():
    a = [0, 0, 0, 2, 4, 2] + Q0                 # width 2 along y, length 4 along x, height 2
    assert om.ground_corners(a) == [(2.0, -1.0), (2.0, 1.0), (-2.0, 1.0), (-2.0, -1.0)]
    assert om.box_iou(a, a) == 1.0
    assert om.box_iou(a, [1, 0, 0, 2, 4, 2] + Q0) == 0.6              # 3x2x2 / (32 - 12)
    assert om.box_iou(a, [10, 0, 0, 2, 4, 2] + Q0) == 0.0             # disjoint
    assert om.box_iou(a, [0, 0, 2.5, 2, 4, 2] + Q0) == 0.0            # above each other
    assert om.box_iou(a, [4, 0, 0, 2, 4, 2] + Q0) == 0.0              # touching faces
    # a unit cube half a unit up inside a: 1*1*0.5 / (16 + 1 - 0.5)
    assert om.box_iou(a, [0, 0, 1.0, 1, 1, 1] + Q0) == 0.5 / 16.5
    # rotated by 90 degrees about z: the footprints share a 2x2 square
    assert om.box_iou(a, [0, 0, 0, 2, 4, 2] + yaw_quat(math.pi / 2)) == pytest.approx(8 / 24, abs=1e-15)
    # a square rotated by 45 degrees inside a larger square: area ratio 2 / 4... as volumes
    big = [0, 0, 0, 2, 2, 1] + Q0
    diamond = [0, 0, 0, math.sqrt(2), math.sqrt(2), 1] + yaw_quat(math.pi / 4)
    assert om.box_iou(big, diamond) == pytest.approx(2 / 4, abs=1e-15)
    # symmetric, and a non-unit quaternion is normalised first
    b = [0.3, -0.2, 0.1, 1.5, 3.0, 1.2] + yaw_quat(0.7)
    assert om.box_iou(a, b) == pytest.approx(om.box_iou(b, a), abs=1e-15)
    b3 = b[:6] + [3 * v for v in b[6:]]
    assert om.box_iou(a, b3) == pytest.approx(om.box_iou(a, b), abs=1e-15)


def test_intersection_against_monte_carlo():
    rng = np.random.default_rng(0)
    a = [0, 0, 0, 2.0, 3.0, 1.0] + yaw_quat(0.4)
    b = [0.5, 0.3, 0, 1.5, 2.5, 1.0] + yaw_quat(-0.9)
    pa, pb = om.ground_corners(a), om.ground_corners(b)

    def inside(poly, x, y):
        ok = np.ones_like(x, dtype=bool)
        for i in range(4):
            (ax, ay), (bx, by) = poly[i], poly[(i + 1) % 4]
            ok &= (bx - ax) * (y - ay) - (by - ay) * (x - ax) >= 0
        return ok
    x, y = rng.uniform(-3, 3, 2_000_000), rng.uniform(-3, 3, 2_000_000)
    mc = 36.0 * np.mean(inside(pa, x, y) & inside(pb, x, y))
    assert om.quad_intersection_area(pa, pb) == pytest.approx(mc, rel=5e-3)


def test_ap_hand_worked():
    # precision envelope: [1, .5, .67, .5] -> [1, .67, .67, .5]; recall steps .25 each... 2 of 4 gts found
    rec = np.array([0.25, 0.25, 0.5, 0.5])
    prec = np.array([1.0, 0.5, 2 / 3, 0.5])
    assert om.get_ap(rec, prec) == pytest.approx(0.25 * 1.0 + 0.25 * (2 / 3))
    assert om.get_ap(np.array([1.0]), np.array([1.0])) == 1.0
    assert om.get_ap(np.array([0.0]), np.array([0.0])) == 0.0


def _records():
    gt = [dict(sample_token="s0", translation=[0, 0, 0], size=[2, 4, 2], rotation=Q0, name="car"),
          dict(sample_token="s0", translation=[10, 0, 0], size=[2, 4, 2], rotation=Q0, name="car"),
          dict(sample_token="s1", translation=[0, 0, 0], size=[1, 1, 2], rotation=Q0, name="pedestrian")]
    pred = [dict(sample_token="s0", translation=[0.2, 0, 0], size=[2, 4, 2], rotation=Q0, name="car", score=0.9),
            dict(sample_token="s0", translation=[0.4, 0, 0], size=[2, 4, 2], rotation=Q0, name="car", score=0.8),
            dict(sample_token="s0", translation=[10, 0.1, 0], size=[2, 4, 2], rotation=Q0, name="car", score=0.7),
            dict(sample_token="s2", translation=[0, 0, 0], size=[2, 4, 2], rotation=Q0, name="car", score=0.95),
            dict(sample_token="s1", translation=[5, 5, 0], size=[1, 1, 2], rotation=Q0, name="pedestrian", score=0.5)]
    return gt, pred


def test_recall_precision_hand_worked():
    gt, pred = _records()
    cars_gt = [g for g in gt if g["name"] == "car"]
    cars = [p for p in pred if p["name"] == "car"]
    rec, prec, ap = om.recall_precision(cars_gt, cars, 0.5)
    # score order: s2 (no gt: fp), 0.9 (tp), 0.8 (duplicate of a matched gt: fp), 0.7 (tp)
    assert rec.tolist() == [0.0, 0.5, 0.5, 1.0]
    assert prec.tolist() == [0.0, 0.5, 1 / 3, 0.5]
    assert ap == pytest.approx(0.5 * 0.5 + 0.5 * 0.5)
    aps = om.get_average_precisions(gt, pred, ["car", "pedestrian", "truck"], 0.5)
    assert aps.tolist() == [pytest.approx(0.5), 0.0, 0.0]


def test_provider_box_decode():
    from fpn import provider
    assert provider.class2angle(0, 0.1, 12) == 0.1
    assert provider.class2angle(3, 0.0, 12) == 3 * (2 * np.pi / 12)
    assert provider.class2angle(11, 0.3, 12) == 11 * (2 * np.pi / 12) + 0.3 - 2 * np.pi     # above pi: wraps
    assert provider.class2angle(11, 0.3, 12, to_label_format=False) == 11 * (2 * np.pi / 12) + 0.3
    np.testing.assert_array_equal(provider.class2angle(np.array([0, 11]), np.array([0.1, 0.3]), 12),
                                  [0.1, 11 * (2 * np.pi / 12) + 0.3 - 2 * np.pi])
    for ang in (0.0, 0.2, 3.0, 6.2, -1.0):
        c, r = provider.angle2class(ang, 12)
        assert 0 <= c < 12 and abs(r) <= np.pi / 12 + 1e-12
        assert (provider.class2angle(c, r, 12, False) - ang) % (2 * np.pi) == pytest.approx(0, abs=1e-9) or \
               (provider.class2angle(c, r, 12, False) - ang) % (2 * np.pi) == pytest.approx(2 * np.pi, abs=1e-9)
    cls, res = provider.size2class(np.array([4.0, 1.6, 1.5]), "Car")
    assert cls == 0
    np.testing.assert_allclose(provider.class2size(cls, res), [4.0, 1.6, 1.5], rtol=0, atol=1e-15)
    # axis-aligned box: l along x, h along y, w along z
    box = provider.get_3d_box((4.0, 2.0, 1.0), 0.0, (10.0, 20.0, 30.0))
    assert box.shape == (8, 3)
    np.testing.assert_array_equal(box[0], [12.0, 20.5, 31.0])
    np.testing.assert_array_equal(box[6], [8.0, 19.5, 29.0])
    # heading pi/2: x' = z, z' = -x
    box = provider.get_3d_box((4.0, 2.0, 1.0), np.pi / 2, (0.0, 0.0, 0.0))
    np.testing.assert_allclose(box[0], [1.0, 0.5, -2.0], atol=1e-15)
    pc = np.array([[1.0, 5.0, 0.0, 9.0]])
    out = provider.rotate_pc_along_y(pc, np.pi / 2)
    assert out is pc
    np.testing.assert_allclose(pc, [[0.0, 5.0, 1.0, 9.0]], atol=1e-15)
    h, w, l, tx, ty, tz, ry = provider.from_prediction_to_label_format(
        np.array([1.0, 2.0, 3.0]), 1, 0.05, 0, np.array([0.1, 0.0, -0.1]), 0.0)
    mean = provider.g_type_mean_size["Car"]
    assert (l, w, h) == (mean[0] + 0.1, mean[1], mean[2] - 0.1)
    assert (tx, tz) == (1.0, 3.0) and ty == 2.0 + h / 2.0
    assert ry == 2 * np.pi / 12 + 0.05


def test_detection_lines_and_files(tmp_path):
    from fpn import inference, provider
    args = dict(id_list=[7, 7, 9], type_list=[0, 1, 2], box2d_list=[[1, 2, 3, 4]] * 3,
                center_list=[np.array([1.0, 2.0, 3.0])] * 3, heading_cls_list=[1, 2, 3],
                heading_res_list=[0.05, 0.0, -0.05], size_cls_list=[0, 3, 5],
                size_res_list=[np.zeros(3)] * 3, rot_angle_list=[0.0, 0.1, -0.1], score_list=[0.5, 0.25, 0.125])
    lines = inference.detection_lines(**args)
    assert list(lines) == [7, 9] and len(lines[7]) == 2
    first = lines[7][0].split(" ")
    assert first[:4] == ["Car", "-1", "-1", "-10"] and len(first) == 16
    mean = provider.g_type_mean_size["Car"]
    assert first[4:8] == ["1.000000", "2.000000", "3.000000", "4.000000"]
    assert first[8:11] == ["%f" % mean[2], "%f" % mean[1], "%f" % mean[0]]
    assert first[-1] == "0.500000"
    assert lines[7][1].startswith("Pedestrian -1 -1 -10 ") and lines[9][0].startswith("Cyclist ")
    inference.write_detection_results(str(tmp_path / "res"), csv_path=str(tmp_path / "out.csv"), **args)
    assert sorted(os.listdir(tmp_path / "res" / "data")) == ["7.txt", "9.txt"]
    assert open(tmp_path / "res" / "data" / "7.txt").read().splitlines() == lines[7]
    assert len(open(tmp_path / "out.csv").read().splitlines()) == 2
    assert inference.write_detection_results(None, **args) is None
