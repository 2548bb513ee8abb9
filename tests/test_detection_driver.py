"""Host side after the network: the detection driver (scripts/test.py:100-247) and the
result-to-box arithmetic (helpers/parse_pointnet_output.py).  The Lyft SDK and pyquaternion
are not installed, so the expected values are hand-derived from the reference's formulas
(parity otherwise unpinned: the reference holds no fixtures for these files)."""
import os

import numpy as np

from fpn import frustum_records as R
from fpn import inference, provider
from fpn.helpers import parse_pointnet_output as PPO
from fpn.model_util import NUM_HEADING_BIN, NUM_SIZE_CLUSTER


class FakeModel:
    """Deterministic stand-in for the detection model: outputs are fixed functions of the
    inputs, so the driver's bookkeeping can be checked without a GPU."""

    def __init__(self):
        self.calls = []

    def predict(self, x):
        pc, oh = np.asarray(x["frustum_point_cloud"]), np.asarray(x["one_hot_vec"])
        b, n, _ = pc.shape
        self.calls.append(b)
        logits = np.stack([pc[:, :, 0], pc[:, :, 2] - 20.0], -1).astype(np.float32)
        center = pc.mean(1)[:, :3].astype(np.float32)
        k = np.arange(NUM_HEADING_BIN, dtype=np.float32)
        hs = np.cos(k[None] * (1 + oh.argmax(1)[:, None]) + pc[:, 0, 0:1]).astype(np.float32)
        hr = (0.01 * k[None] + pc[:, 1, 0:1]).astype(np.float32)
        j = np.arange(NUM_SIZE_CLUSTER, dtype=np.float32)
        ss = np.sin(j[None] * 2 + pc[:, 2, 0:1]).astype(np.float32)
        sr = (j[None, :, None] * 0.1 + pc[:, 3, None, 0:3]).astype(np.float32)
        return [logits, center, hs, hr, ss, sr]


def write_test_records(path, n, seed=0):
    rng = np.random.RandomState(seed)
    recs, rows = [], []
    for i in range(n):
        pc = (rng.standard_normal((R.NUM_POINT, 3)) * [2, 1, 2] + [0, 0, 20]).astype(np.float32)
        rot, prob = float(rng.uniform(-3, 3)), float(rng.uniform(0, 1))
        cls = i % 3
        one_hot = np.eye(3, dtype=np.int64)[cls]
        box2d = rng.uniform(0, 100, 4).astype(np.float32)
        token = "frame%d" % (i // 2)             # two detections per frame
        recs.append(R.serialize_test_example(pc, rot, prob, one_hot, cls, token, box2d))
        rows.append((pc, one_hot, rot, prob, cls, token, box2d))
    R.write_records(path, recs)
    return rows


def test_driver_runs_batches_and_writes_results(tmp_path):
    path = str(tmp_path / "test.tfrec")
    rows = write_test_records(path, 7)
    batches = list(inference.make_test_iterator(path, 3))
    assert [b["frustum_point_cloud"].shape[0] for b in batches] == [3, 3]       # remainder dropped (:269)
    assert batches[1]["token"][0] == b"frame1" and batches[0]["rot_angle"].dtype == np.float32
    np.testing.assert_array_equal(batches[1]["frustum_point_cloud"][2], rows[5][0])

    model = FakeModel()
    out_pickle = str(tmp_path / "det.pkl")
    result_dir = str(tmp_path / "result")
    idx_path = str(tmp_path / "val.txt")
    with open(idx_path, "w") as f:
        f.write("frame0\nframe1\nframe9\n")
    lists = inference.test(model, iter(batches), out_pickle, result_dir=result_dir, idx_path=idx_path,
                           csv_path=str(tmp_path / "out.csv"))
    assert model.calls == [3, 3] and len(lists["ps"]) == 6

    # the lists equal a direct post-processing of the model's outputs
    pc_all = np.stack([r[0] for r in rows[:6]])
    oh_all = np.stack([r[1] for r in rows[:6]]).astype(np.float32)
    seg, c, hc, hr, sc, sr, s = inference.postprocess(*FakeModel().predict({"frustum_point_cloud": pc_all,
                                                                            "one_hot_vec": oh_all}))
    np.testing.assert_array_equal(np.stack(lists["segp"]), seg)
    np.testing.assert_array_equal(np.stack(lists["center"]), c)
    np.testing.assert_array_equal(np.asarray(lists["heading_cls"]), hc)
    np.testing.assert_array_equal(np.asarray(lists["heading_res"]), hr)
    np.testing.assert_array_equal(np.asarray(lists["size_cls"]), sc)
    np.testing.assert_array_equal(np.stack(lists["size_res"]), sr)
    np.testing.assert_array_equal(np.asarray(lists["score"]), s)

    # label files: one per frame seen, empty ones for the rest of the index file
    data = os.path.join(result_dir, "data")
    assert sorted(os.listdir(data)) == ["frame0.txt", "frame1.txt", "frame2.txt", "frame9.txt"]
    assert os.path.getsize(os.path.join(data, "frame9.txt")) == 0
    lines = open(os.path.join(data, "frame1.txt")).read().splitlines()
    assert len(lines) == 2
    f = lines[0].split()
    assert f[0] == "Cyclist" and f[1:4] == ["-1", "-1", "-10"] and len(f) == 16       # row 2: class 2
    h, w, l, tx, ty, tz, ry = provider.from_prediction_to_label_format(c[2], hc[2], hr[2], sc[2], sr[2], rows[2][2])
    np.testing.assert_allclose([float(v) for v in f[8:15]], [h, w, l, tx, ty, tz, ry], atol=1e-6)
    np.testing.assert_allclose(float(f[15]), s[2], atol=1e-6)

    # the pickle reads back (restricted unpickler) into boxes
    boxes = PPO.read_frustum_pointnet_output(out_pickle, from_rgb_detection=True)
    assert len(boxes) == 6
    want = PPO.get_box_from_inference(hc[4], hr[4], np.float32(rows[4][2]), sc[4], sr[4], c[4], s[4])
    np.testing.assert_array_equal(boxes[4]["center"], want["center"])
    np.testing.assert_array_equal(boxes[4]["orientation"], want["orientation"])


def test_inference_leaves_rows_beyond_the_last_batch_zero():
    rng = np.random.RandomState(1)
    pc = (rng.standard_normal((5, 64, 3)) + [0, 0, 20]).astype(np.float32)
    oh = np.eye(3, dtype=np.float32)[[0, 1, 2, 0, 1]]
    m = FakeModel()
    seg, c, hc, hr, sc, sr, s = inference.inference({"frustum_point_cloud": pc, "one_hot_vec": oh}, m, 2)
    assert m.calls == [2, 2]
    assert np.all(c[4] == 0) and s[4] == 0 and hc[4] == 0 and np.all(seg[4] == 0)
    ref = inference.postprocess(*FakeModel().predict({"frustum_point_cloud": pc[2:4], "one_hot_vec": oh[2:4]}))
    np.testing.assert_array_equal(c[2:4], ref[1])
    np.testing.assert_array_equal(s[2:4], ref[6])


def rotation_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_box_from_inference_geometry():
    """Hand-derived: after the two rotations the box's length axis (+x) points along
    (cos a, 0, -sin a) in the camera frame (a = heading angle: a turn about +y), its 'up'
    axis (+z) along -y (the camera's y points down), and the centre is the frustum-frame
    centre turned by -(rot_angle + pi/2) about y."""
    hcls, hres, rot = 3, 0.1, 0.4
    scls, sres = 1, np.array([0.1, -0.2, 0.3])
    center = np.array([1.0, 2.0, 10.0])
    box = PPO.get_box_from_inference(hcls, hres, rot, scls, sres, center, 0.9, "Car")
    a = hcls * 2 * np.pi / NUM_HEADING_BIN + hres + rot
    assert np.isclose(PPO.get_heading_angle(hcls, hres, rot), a)
    rm = rotation_matrix(box["orientation"])
    np.testing.assert_allclose(rm @ [1, 0, 0], [np.cos(a), 0, -np.sin(a)], atol=1e-12)
    np.testing.assert_allclose(rm @ [0, 0, 1], [0, -1, 0], atol=1e-12)
    assert np.isclose(np.linalg.norm(box["orientation"]), 1)
    l, w, h = provider.class2size(scls, sres)
    assert box["size"] == [w, l, h] and box["name"] == "Car" and box["score"] == 0.9
    t = rot + np.pi / 2                       # rotate_pc_along_y(pc, -t): x' = c x - s z, z' = s x + c z, c/s of -t
    c_, s_ = np.cos(-t), np.sin(-t)
    np.testing.assert_allclose(box["center"], [c_ * 1.0 - s_ * 10.0, 2.0, s_ * 1.0 + c_ * 10.0], atol=1e-12)

    # carried into a world frame: rotation by 90 degrees about z plus a translation
    m = np.eye(4)
    m[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    m[:3, 3] = [5, 6, 7]
    wbox = PPO.get_box_from_inference(hcls, hres, rot, scls, sres, center, 0.9, "Car", sensor_to_world=m)
    np.testing.assert_allclose(wbox["center"], m[:3, :3] @ box["center"] + [5, 6, 7], atol=1e-12)
    np.testing.assert_allclose(rotation_matrix(wbox["orientation"]), m[:3, :3] @ rm, atol=1e-12)
    # the quaternion extraction on matrices with negative trace
    for axis, ang in (([1, 0, 0], 3.0), ([0, 1, 0], 2.9), ([0, 0, 1], -3.1), ([1, 2, 3], 2.5)):
        q = PPO.quaternion_from_axis_angle(axis, ang)
        q2 = PPO.quaternion_from_rotation_matrix(rotation_matrix(q))
        np.testing.assert_allclose(rotation_matrix(q2), rotation_matrix(q), atol=1e-12)
