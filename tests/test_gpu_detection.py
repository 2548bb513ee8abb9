"""GPU parity of the detection-record kernel (fpn_detection_records) with the CPU oracle:
oracle.v1_numpy.inference_postprocess (the arithmetic of scripts/test.py:145-160) and
oracle.v1_numpy.label_fields (scripts/provider.py:366-373 written out independently of
the product's fpn/provider.py).

Bars.  Integer outputs (mask bits, class picks, positive count), the copied residuals and
centres, and the seven label fields: bit-exact (the label conversion is float64 arithmetic
compiled without contraction, with correctly rounded float32 cos/sin - the oracle's
trig="exact").  Probabilities and the score pass through a float32 exp whose last bit
differs between numpy and the device library: |got - want| <= 1e-6 (observed ~1e-7).
Against numpy's own float32 cos/sin (what the reference's host would compute) tx and tz
agree to 4e-7 relative to the centre's distance."""
import numpy as np
import pytest
import torch

from oracle import v1_numpy as V

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-6


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def oracle_records(logits, box, c1, rot):
    """Detection records from the oracle's two functions on host arrays (fp32 model graph
    products re-done in numpy float32: models/model_util.py:229-241)."""
    box = np.asarray(box, np.float32)
    centers = box[:, 0:3] + np.asarray(c1, np.float32)
    hs, hrn = box[:, 3:15], box[:, 15:27]
    ss, srn = box[:, 27:35], box[:, 35:59].reshape(-1, 8, 3)
    hr = hrn * np.float32(np.pi / 12)
    sr = srn * V._MEAN_SIZE.astype(np.float32)[None]
    seg, cen, hc, hres, sc, sres, score = V.inference_postprocess(logits, centers, hs, hr, ss, sr)
    rot = np.zeros(len(box), np.float32) if rot is None else rot
    lab = V.label_fields(cen, hc, hres, sc, sres, rot, trig="exact")
    lab_np = V.label_fields(cen, hc, hres, sc, sres, rot, trig="numpy")
    hp = np.max(V.softmax(hs), 1)
    sp = np.max(V.softmax(ss), 1)
    return dict(seg=seg, center=cen, hc=hc, hres=hres, sc=sc, sres=sres, score=score, label=lab, label_numpy=lab_np,
                hp=hp, sp=sp)


def check(device, logits, box, c1, rot, want_seg=True):
    from fpn import inference as I
    rec, bits = I.detection_records(dev(logits, device), dev(box, device), dev(c1, device),
                                    None if rot is None else dev(rot, device), want_seg=want_seg)
    rec = rec.cpu().numpy()
    want = oracle_records(logits, box, c1, rot)
    b, n = logits.shape[:2]
    assert rec.shape == (b, I.RECORD) and rec.dtype == np.float64
    if want_seg:
        seg = I.unpack_seg_bits(bits, n)
        np.testing.assert_array_equal(seg, want["seg"])
        np.testing.assert_array_equal(rec[:, I.POSITIVES], want["seg"].sum(1))
        words = bits.cpu().numpy().view(np.uint32)
        if n % 32:      # bits past the last point stay clear
            assert (words[:, -1] >> np.uint32(n % 32)).max() == 0
    np.testing.assert_array_equal(rec[:, I.HEADING_CLS], want["hc"])
    np.testing.assert_array_equal(rec[:, I.SIZE_CLS], want["sc"])
    np.testing.assert_array_equal(rec[:, I.HEADING_RES], want["hres"])
    np.testing.assert_array_equal(rec[:, I.SIZE_RES], want["sres"])
    np.testing.assert_array_equal(rec[:, I.CENTER], want["center"])
    np.testing.assert_array_equal(rec[:, I.LABEL], want["label"])
    dist = np.sqrt((want["center"] ** 2).sum(1, keepdims=True)) + 1.0
    assert (np.abs(rec[:, I.LABEL] - want["label_numpy"]) / dist).max() <= 4e-7
    np.testing.assert_allclose(rec[:, I.HEADING_PROB], want["hp"], rtol=0, atol=SCORE_TOL)
    np.testing.assert_allclose(rec[:, I.SIZE_PROB], want["sp"], rtol=0, atol=SCORE_TOL)
    nan = np.isnan(want["score"])
    np.testing.assert_array_equal(np.isnan(rec[:, I.SCORE]), nan)
    if (~nan).any():
        err = np.abs(rec[~nan, I.SCORE] - want["score"][~nan]).max()
        assert err <= SCORE_TOL, "score off by %.3e" % err
    return rec, want


def random_outputs(rng, b, n, positive_rate=0.3):
    logits = rng.standard_normal((b, n, 2)).astype(np.float32) * 3
    logits[..., 1] += np.float32(np.log(positive_rate / (1 - positive_rate)))
    box = rng.standard_normal((b, 59)).astype(np.float32)
    box[:, 0:3] *= 2
    c1 = (rng.standard_normal((b, 3)) * np.array([3, 1, 20])).astype(np.float32)
    rot = rng.uniform(-np.pi, np.pi, b).astype(np.float32)
    return logits, box, c1, rot


@pytest.mark.parametrize("b,n", [(32, 1024), (1, 1), (3, 31), (5, 33), (7, 64), (9, 100), (130, 2048), (4, 4097)])
def test_records_match_oracle_random(device, b, n):
    rng = np.random.RandomState(b * 10007 + n)
    check(device, *random_outputs(rng, b, n))


def test_records_without_rotation_and_without_mask_words(device):
    rng = np.random.RandomState(5)
    logits, box, c1, _ = random_outputs(rng, 17, 512)
    rec, want = check(device, logits, box, c1, None, want_seg=False)
    # unrotated: tx, tz are the centre's own x, z
    np.testing.assert_array_equal(rec[:, 3], want["center"][:, 0])
    np.testing.assert_array_equal(rec[:, 5], want["center"][:, 2])


def test_frustums_without_positive_points_score_nan(device):
    """scripts/test.py:148 divides by the positive count: 0/0 -> NaN score, everything else
    defined."""
    rng = np.random.RandomState(6)
    logits, box, c1, rot = random_outputs(rng, 12, 256)
    logits[::3, :, 1] = logits[::3, :, 0] - 1.0         # no positive point in every third frustum
    logits[1, :, 1] = logits[1, :, 0]                   # exact ties are not positive either (argmax -> 0)
    rec, want = check(device, logits, box, c1, rot)
    assert np.isnan(rec[::3, 7]).all() and np.isnan(rec[1, 7])
    assert not np.isnan(rec[2, 7])
    assert np.isfinite(rec[:, :7]).all()


def test_heading_wrap_and_argmax_ties(device):
    """class2angle moves an angle above pi down by 2 pi BEFORE the frustum angle is added
    (provider.py:80-81, :370); np.argmax keeps the first of equal maxima."""
    rng = np.random.RandomState(7)
    b = 24
    logits, box, c1, rot = random_outputs(rng, b, 128)
    for i in range(b):                       # heading bin i mod 12 wins, residual pushes across pi for bins 6..
        box[i, 3:15] = -1.0
        box[i, 3 + i % 12] = 2.0
    box[:, 15:27] = rng.uniform(-1, 1, (b, 12)).astype(np.float32)
    # ties: two and three equal maxima, first index wins; all equal -> 0
    box[0, 3:15] = 0.5
    box[1, 3:15] = -1.0
    box[1, [4, 9]] = 3.0
    box[2, 27:35] = 1.25
    box[3, 27:35] = 0.0
    box[3, [30, 33, 34]] = 7.0
    rec, want = check(device, logits, box, c1, rot)
    assert rec[0, 8] == 0 and rec[1, 8] == 1 and rec[2, 9] == 0 and rec[3, 9] == 3
    angle = want["hc"] * (2 * np.pi / 12) + want["hres"]
    assert (angle > np.pi).any() and (angle <= np.pi).any()
    wrapped = angle > np.pi
    np.testing.assert_array_equal(rec[wrapped, 6], (angle[wrapped] - 2 * np.pi) + rot[wrapped].astype(np.float64))
    # exactly pi is not wrapped (strict >): bin 6 with zero residual
    box[4, 3:15] = -1.0
    box[4, 3 + 6] = 2.0
    box[4, 15:27] = 0.0
    rec, _ = check(device, logits, box, c1, rot)
    assert rec[4, 6] == 6 * (2 * np.pi / 12) + np.float64(rot[4])


def test_v1_network_to_records_config0(device):
    """BASELINE config 0 (32 frustums x 1024 points, v1): network on the device, then the
    record kernel; the oracle post-processes the same network outputs.  The model's own
    derived end points (heading_residuals, size_residuals, center: fp32 torch products)
    must be the numbers the kernel derives from the raw box vector."""
    from fpn import inference as I
    from fpn import plan, synthetic, variables
    st = variables.VariableStore(device, seed=1)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(2)
    pc, one_hot = synthetic.make_frustums(32, 1024, 4, seed=0)
    ep = plan.V1Plan(st, 4, True).forward(dev(pc, device), dev(one_hot, device))
    rng = np.random.RandomState(8)
    rot = rng.uniform(-np.pi, np.pi, 32).astype(np.float32)
    logits = ep["mask_logits"].cpu().numpy()
    rec, want = check(device, logits, ep["box_output"].cpu().numpy(), ep["stage1_center"].cpu().numpy(), rot)
    # the torch-side end points agree with the kernel's derivation bit for bit
    hc, sc = rec[:, I.HEADING_CLS].astype(int), rec[:, I.SIZE_CLS].astype(int)
    rows = np.arange(32)
    np.testing.assert_array_equal(rec[:, I.HEADING_RES], ep["heading_residuals"].cpu().numpy()[rows, hc])
    np.testing.assert_array_equal(rec[:, I.SIZE_RES], ep["size_residuals"].cpu().numpy()[rows, sc])
    np.testing.assert_array_equal(rec[:, I.CENTER], ep["center"].cpu().numpy())
    np.testing.assert_array_equal(I.unpack_seg_bits(
        I.detection_records(ep["mask_logits"], ep["box_output"], ep["stage1_center"], dev(rot, device), True)[1], 1024),
        ep["mask"].cpu().numpy().astype(np.int64))
    # and the host form of the same arithmetic (fpn.inference.postprocess) gives the oracle's arrays
    host = I.postprocess(logits, ep["center"], ep["heading_scores"], ep["heading_residuals"], ep["size_scores"],
                         ep["size_residuals"])
    np.testing.assert_array_equal(host[2], want["hc"])
    np.testing.assert_allclose(host[6], want["score"], rtol=0, atol=1e-12, equal_nan=True)


def test_detect_method_and_label_lines(device, tmp_path):
    """DetectionModel.detect -> records; fpn.inference.inference / test use them: the seven
    arrays of scripts/test.py:162 and the KITTI label lines equal the host path's
    (per-box provider.from_prediction_to_label_format) to the printed precision."""
    from fpn import inference as I
    from fpn import synthetic, variables
    from fpn.models import FPointNet_Keras as K
    from fpn import plan
    st = variables.VariableStore(device, seed=3)
    plan.init_variables(st, 3, False)
    with variables.use_store(st):
        _, det = K.Frustum_Pointnet_Model(point_cloud_shape=(512, 3), batch_size=8)
    pc, one_hot = synthetic.make_frustums(8, 512, 3, seed=4)
    rot = np.random.RandomState(9).uniform(-3, 3, 8).astype(np.float32)
    feed = {"frustum_point_cloud": pc, "one_hot_vec": one_hot, "rot_angle": rot}
    seg, centers, hcls, hres, scls, sres, scores = I.inference(feed, det, 8)

    class HostOnly:            # the same model without `detect`: the numpy path
        predict = staticmethod(det.predict)
    seg2, centers2, hcls2, hres2, scls2, sres2, scores2 = I.inference(feed, HostOnly, 8)
    np.testing.assert_array_equal(seg, seg2)
    np.testing.assert_array_equal(centers, centers2)
    np.testing.assert_array_equal(hcls, hcls2)
    np.testing.assert_array_equal(scls, scls2)
    np.testing.assert_array_equal(hres, hres2)
    np.testing.assert_array_equal(sres, sres2)
    np.testing.assert_allclose(scores, scores2, rtol=0, atol=SCORE_TOL, equal_nan=True)
    records, _ = I.inference_records(feed, det, 8)
    ids = ["a", "b", "a", "c", "b", "a", "d", "d"]
    types = [0, 1, 2, 0, 1, 2, 0, 1]
    boxes = np.arange(32, dtype=np.float32).reshape(8, 4)
    from_rec = I.detection_lines_from_records(ids, types, boxes, records)
    from_host = I.detection_lines(ids, types, boxes, [c.copy() for c in centers2], hcls2, hres2, scls2, sres2, rot,
                                  records[:, 7])
    assert list(from_rec) == list(from_host)
    for k in from_rec:
        for a, b in zip(from_rec[k], from_host[k]):
            fa, fb = a.split(), b.split()
            assert fa[:8] == fb[:8]
            np.testing.assert_allclose([float(x) for x in fa[8:]], [float(x) for x in fb[8:]], rtol=0, atol=2e-6)
