"""Post-processing of the detector outputs: counterpart of the arithmetic of
scripts/test.py:inference (:122-162) and of the label conversion its writer calls
(scripts/provider.py:366-373).

Device path: `detection_records` runs fpn_detection_records on the network's outputs
(one wave per frustum: soft-max scores, class picks, residual picks, label fields
h,w,l,tx,ty,tz,ry in float64) and the (B, 20) float64 record is all that has to leave the
GPU - or travel between GPUs.  `inference` and `test` use it whenever the model offers
`detect` (the models of this package do).  `postprocess` is the same arithmetic on host
arrays for callers that only hold the six `det_model` outputs in numpy."""
import numpy as np

# columns of a detection record (FPN_DETECTION_RECORD doubles, include/fpn.h)
RECORD = 20
LABEL = slice(0, 7)          # h, w, l, tx, ty, tz, ry
SCORE, HEADING_CLS, SIZE_CLS, HEADING_RES = 7, 8, 9, 10
SIZE_RES = slice(11, 14)
CENTER = slice(14, 17)
POSITIVES, HEADING_PROB, SIZE_PROB = 17, 18, 19


def detection_records(mask_logits, box_output, stage1_center, rot_angle=None, want_seg=False):
    """Network outputs on the device -> (records (B,20) float64, seg_bits (B,ceil(N/32))
    int32 or None), both on the device.  mask_logits (B,N,2), box_output (B,59),
    stage1_center (B,3), rot_angle (B,) float32 or None (frustums not rotated)."""
    import torch
    from . import _lib
    _lib.require_device_tensor("mask_logits", mask_logits, torch.float32, 3, 2)
    _lib.require_device_tensor("box_output", box_output, torch.float32, 2, 59)
    _lib.require_device_tensor("stage1_center", stage1_center, torch.float32, 2, 3)
    b, n, _ = mask_logits.shape
    if box_output.shape[0] != b or stage1_center.shape[0] != b:
        raise ValueError("mask_logits, box_output and stage1_center must hold the same number of frustums")
    if n < 1:
        raise ValueError("detection_records needs at least one point per frustum")
    if rot_angle is not None:
        _lib.require_device_tensor("rot_angle", rot_angle, torch.float32, 1)
        if rot_angle.shape[0] != b:
            raise ValueError("rot_angle must hold one angle per frustum")
    dev = _lib.same_device(mask_logits, box_output, stage1_center, rot_angle)
    records = torch.empty((b, RECORD), dtype=torch.float64, device=dev)
    seg = torch.empty((b, (n + 31) // 32), dtype=torch.int32, device=dev) if want_seg else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fpn_detection_records(
            b, n, _lib.ptr(mask_logits), _lib.ptr(box_output), _lib.ptr(stage1_center), _lib.ptr(rot_angle),
            _lib.ptr(records), _lib.ptr(seg), _lib.stream_of(mask_logits)))
    return records, seg


def unpack_seg_bits(seg_bits, n):
    """(B, ceil(n/32)) mask words (device tensor or array) -> (B,n) int64 0/1 host array,
    the `np.argmax(logits, 2)` of scripts/test.py:162."""
    words = seg_bits.cpu().numpy() if hasattr(seg_bits, "cpu") else np.asarray(seg_bits)
    bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), axis=1, bitorder="little")
    return bits[:, :n].astype(np.int64)


def outputs_from_records(records, seg_bits=None, num_point=None):
    """Detection records -> the seven arrays scripts/test.py:inference returns (:162):
    (seg (B,N) or None, centers (B,3), heading_cls, heading_res, size_cls, size_res (B,3),
    scores), float64 / int64 host arrays."""
    rec = records.cpu().numpy() if hasattr(records, "cpu") else np.asarray(records)
    seg = None if seg_bits is None else unpack_seg_bits(seg_bits, num_point)
    return (seg, rec[:, CENTER].copy(), rec[:, HEADING_CLS].astype(np.int64), rec[:, HEADING_RES].copy(),
            rec[:, SIZE_CLS].astype(np.int64), rec[:, SIZE_RES].copy(), rec[:, SCORE].copy())


def softmax(x):
    """scripts/test.py:38-43."""
    probs = np.exp(x - np.max(x, axis=len(x.shape) - 1, keepdims=True))
    probs /= np.sum(probs, axis=len(x.shape) - 1, keepdims=True)
    return probs


def postprocess(logits, centers, heading_scores, heading_residuals, size_scores, size_residuals):
    """-> (seg (B,N) int, centers (B,3), heading_cls (B,), heading_res (B,), size_cls (B,),
    size_res (B,3), scores (B,)).  scores = mean of (mean object-point probability,
    max heading probability, max size probability), test.py:145-152; a frustum with no
    object point gets NaN there exactly as in the reference (:148 divides by zero)."""
    to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    bl, bc, bhs, bhr, bss, bsr = [to_np(t) for t in
                                  (logits, centers, heading_scores, heading_residuals, size_scores, size_residuals)]
    n = bl.shape[0]
    logits64 = np.zeros(bl.shape); logits64[...] = bl
    centers64 = np.zeros((n, 3)); centers64[...] = bc
    hl = np.zeros(bhs.shape); hl[...] = bhs
    hr = np.zeros(bhr.shape); hr[...] = bhr
    sl = np.zeros(bss.shape); sl[...] = bss
    sr = np.zeros(bsr.shape); sr[...] = bsr
    seg_prob = softmax(bl)[:, :, 1]
    seg_mask = np.argmax(bl, 2)
    with np.errstate(invalid="ignore", divide="ignore"):
        mask_mean_prob = np.sum(seg_prob * seg_mask, 1) / np.sum(seg_mask, 1)
    heading_prob = np.max(softmax(bhs), 1)
    size_prob = np.max(softmax(bss), 1)
    scores = np.zeros((n,)); scores[...] = (mask_mean_prob + heading_prob + size_prob) / 3
    heading_cls = np.argmax(hl, 1)
    size_cls = np.argmax(sl, 1)
    heading_res = np.array([hr[i, heading_cls[i]] for i in range(n)])
    size_res = np.vstack([sr[i, size_cls[i], :] for i in range(n)])
    return np.argmax(logits64, 2), centers64, heading_cls, heading_res, size_cls, size_res, scores


def detection_lines(id_list, type_list, box2d_list, center_list, heading_cls_list, heading_res_list,
                    size_cls_list, size_res_list, rot_angle_list, score_list):
    """KITTI label lines of scripts/test.py:write_detection_results (:58-72), grouped by
    sample id in first-seen order: 'Type -1 -1 -10 x1 y1 x2 y2 h w l tx ty tz ry score'
    with %f formatting."""
    from . import provider
    onehot2type = {v: k for k, v in provider.g_type2onehotclass.items()}
    results = {}
    for i in range(len(center_list)):
        box2d = box2d_list[i]
        fields = provider.from_prediction_to_label_format(center_list[i], heading_cls_list[i], heading_res_list[i],
                                                          size_cls_list[i], size_res_list[i], rot_angle_list[i])
        line = onehot2type[type_list[i]] + " -1 -1 -10 "
        line += "%f %f %f %f " % (box2d[0], box2d[1], box2d[2], box2d[3])
        line += "%f %f %f %f %f %f %f %f" % (fields + (score_list[i],))
        results.setdefault(id_list[i], []).append(line)
    return results


def detection_lines_from_records(id_list, type_list, box2d_list, records):
    """The same label lines from detection records (columns 0..7 = h w l tx ty tz ry score):
    the label conversion has already run on the device, only the formatting is left."""
    from . import provider
    onehot2type = {v: k for k, v in provider.g_type2onehotclass.items()}
    rec = records.cpu().numpy() if hasattr(records, "cpu") else np.asarray(records)
    results = {}
    for i in range(rec.shape[0]):
        box2d = box2d_list[i]
        line = onehot2type[type_list[i]] + " -1 -1 -10 "
        line += "%f %f %f %f " % (box2d[0], box2d[1], box2d[2], box2d[3])
        line += "%f %f %f %f %f %f %f %f" % tuple(rec[i, :8])
        results.setdefault(id_list[i], []).append(line)
    return results


def write_detection_results(result_dir, id_list, type_list, box2d_list, center_list, heading_cls_list,
                            heading_res_list, size_cls_list, size_res_list, rot_angle_list, score_list,
                            csv_path="results_output.csv", records=None):
    """Write <result_dir>/data/<id>.txt label files and the id -> lines CSV
    (scripts/test.py:51-90; the reference always writes the CSV into the working
    directory, pass csv_path=None to skip it).  records: detection records of the same
    frustums - the lines are then formatted from them and the per-box lists are not used."""
    import csv
    import os
    if result_dir is None:
        return
    if records is not None:
        results = detection_lines_from_records(id_list, type_list, box2d_list, records)
    else:
        results = detection_lines(id_list, type_list, box2d_list, center_list, heading_cls_list, heading_res_list,
                                  size_cls_list, size_res_list, rot_angle_list, score_list)
    output_dir = os.path.join(result_dir, "data")
    os.makedirs(output_dir, exist_ok=True)
    for idx, lines in results.items():
        with open(os.path.join(output_dir, "%s.txt" % (idx,)), "w") as fout:
            for line in lines:
                fout.write(line + "\n")
    if csv_path:
        with open(csv_path, "w") as csv_file:
            writer = csv.writer(csv_file)
            for key, value in results.items():
                writer.writerow([key, value])


# ---------------------------------------------------------------------------
# the detection driver of scripts/test.py (:100-247): record file -> model -> result lists
# -> pickle + KITTI label files
# ---------------------------------------------------------------------------
NUM_CLASSES = 2


def fill_files(output_dir, to_fill_filename_list):
    """Create empty files for the names that do not exist yet (scripts/test.py:99-105)."""
    import os
    for filename in to_fill_filename_list:
        filepath = os.path.join(output_dir, filename)
        if not os.path.exists(filepath):
            open(filepath, "w").close()


def make_test_iterator(path, batch_size, lenient=True):
    """Batches of a detection-time record file as the dicts scripts/test.py:108-119 yields
    (keys frustum_point_cloud, one_hot_vec, rot_angle, rgb_prob, cls_type, token, box_2D),
    numpy arrays, incomplete last batch dropped (`batch(..., drop_remainder=True)`, :269).
    lenient: accept the list types the reference's writer produces where its parser
    declares others (sample_token is written as bytes, declared int64)."""
    from . import frustum_records as fr
    rows = []

    def flush():
        cols = list(zip(*rows))
        rows.clear()
        return {"frustum_point_cloud": np.stack(cols[0]), "one_hot_vec": np.stack(cols[1]),
                "rot_angle": np.asarray(cols[2], np.float32), "rgb_prob": np.asarray(cols[3], np.float32),
                "cls_type": np.asarray(cols[4]), "token": np.asarray(cols[5]), "box_2D": np.stack(cols[6])}
    for rec in fr.read_records(path):
        rows.append(fr.parse_test_data(rec, lenient))
        if len(rows) == batch_size:
            yield flush()


def inference_records(feed_dict, model, batch_size):
    """The device form of `inference`: batch by batch through `model.detect` (network +
    fpn_detection_records) -> (records (total,20) float64, seg (total,N) int64) host arrays.
    feed_dict may carry "rot_angle" (total,) for the label fields; without it they are
    those of unrotated frustums.  Rows beyond the last complete batch stay zero, like the
    reference's buffers."""
    pc = np.asarray(feed_dict["frustum_point_cloud"])
    oh = np.asarray(feed_dict["one_hot_vec"])
    total = pc.shape[0]
    records = np.zeros((total, RECORD))
    seg = np.zeros((total, pc.shape[1]), np.int64)
    rot = feed_dict.get("rot_angle")
    for i in range(int(total / batch_size)):
        sl = slice(i * batch_size, (i + 1) * batch_size)
        rec, bits = model.detect(pc[sl], oh[sl], None if rot is None else np.asarray(rot)[sl], want_seg=True)
        records[sl] = rec.cpu().numpy()
        seg[sl] = unpack_seg_bits(bits, pc.shape[1])
    return records, seg


def inference(feed_dict, model, batch_size):
    """scripts/test.py:122-162: run `model.predict` batch by batch and post-process.
    The reference hands the WHOLE feed_dict to predict in every iteration (:135), which
    only fits its buffers when the dict holds exactly one batch - the way its driver calls
    it; here iteration i predicts rows [i*batch_size, (i+1)*batch_size), identical in that
    case.  Rows beyond the last complete batch keep the zero initialisation, as there."""
    pc = np.asarray(feed_dict["frustum_point_cloud"])
    oh = np.asarray(feed_dict["one_hot_vec"])
    total = pc.shape[0]
    num_batches = int(total / batch_size)
    if hasattr(model, "detect"):
        records, seg = inference_records(feed_dict, model, batch_size)
        return (seg,) + outputs_from_records(records)[1:]
    logits = np.zeros((total, pc.shape[1], NUM_CLASSES))
    centers = np.zeros((total, 3))
    from .model_util import NUM_HEADING_BIN, NUM_SIZE_CLUSTER
    heading_logits = np.zeros((total, NUM_HEADING_BIN))
    heading_residuals = np.zeros((total, NUM_HEADING_BIN))
    size_logits = np.zeros((total, NUM_SIZE_CLUSTER))
    size_residuals = np.zeros((total, NUM_SIZE_CLUSTER, 3))
    scores = np.zeros((total,))
    for i in range(num_batches):
        sl = slice(i * batch_size, (i + 1) * batch_size)
        out = model.predict({"frustum_point_cloud": pc[sl], "one_hot_vec": oh[sl]})
        seg, c, hc, hr, sc, sr, s = postprocess(*out)
        logits[sl], centers[sl], heading_logits[sl] = out[0], c, out[2]
        heading_residuals[sl], size_logits[sl], size_residuals[sl] = out[3], out[4], out[5]
        scores[sl] = s
    heading_cls = np.argmax(heading_logits, 1)
    size_cls = np.argmax(size_logits, 1)
    heading_res = np.array([heading_residuals[i, heading_cls[i]] for i in range(total)])
    size_res = np.vstack([size_residuals[i, size_cls[i], :] for i in range(total)]) if total else np.zeros((0, 3))
    return np.argmax(logits, 2), centers, heading_cls, heading_res, size_cls, size_res, scores


def test(model, test_loader, output_filename, result_dir=None, idx_path=None, batch_size=None, csv_path="results_output.csv"):
    """scripts/test.py:165-247: run the detector over `test_loader` (an iterable of the
    dicts of make_test_iterator), pickle the thirteen result lists into `output_filename`
    (ps, segp, center, heading_cls, heading_res, size_cls, size_res, rot_angle, score,
    onehot, class, token, box2d - the order of :223-236) and write the KITTI label files;
    `idx_path` lists frame names that get an empty file when nothing was detected.
    Returns the lists as a dict."""
    import os
    import pickle
    names = ("ps", "segp", "center", "heading_cls", "heading_res", "size_cls", "size_res", "rot_angle", "score",
             "onehot", "class", "token", "box2d")
    lists = {k: [] for k in names}
    all_records = []
    for data_dict in test_loader:
        pc = data_dict["frustum_point_cloud"]
        bs = batch_size or pc.shape[0]
        if hasattr(model, "detect"):
            rec, seg = inference_records({"frustum_point_cloud": pc, "one_hot_vec": data_dict["one_hot_vec"],
                                          "rot_angle": data_dict["rot_angle"]}, model, bs)
            all_records.append(rec if pc.shape[0] % bs == 0 else None)
            _, centers, hcls, hres, scls, sres, scores = outputs_from_records(rec)
        else:
            seg, centers, hcls, hres, scls, sres, scores = inference(
                {"frustum_point_cloud": pc, "one_hot_vec": data_dict["one_hot_vec"]}, model, bs)
        for i in range(pc.shape[0]):
            lists["ps"].append(pc[i, ...])
            lists["segp"].append(seg[i, ...])
            lists["center"].append(centers[i, :])
            lists["heading_cls"].append(hcls[i])
            lists["heading_res"].append(hres[i])
            lists["size_cls"].append(scls[i])
            lists["size_res"].append(sres[i, :])
            lists["rot_angle"].append(data_dict["rot_angle"][i])
            lists["score"].append(scores[i])
            lists["onehot"].append(data_dict["one_hot_vec"][i])
            lists["class"].append(data_dict["cls_type"][i])
            lists["token"].append(data_dict["token"][i])
            lists["box2d"].append(data_dict["box_2D"][i])
    with open(output_filename, "wb") as fp:
        for k in names:
            pickle.dump(lists[k], fp)
    if result_dir is not None:
        token_ids = [t.decode("utf-8") if isinstance(t, bytes) else t for t in lists["token"]]
        # (copies: the label conversion rotates the centre it is given in place, provider.py:366-373)
        write_detection_results(result_dir, token_ids, [int(c) for c in lists["class"]], lists["box2d"],
                                [np.array(c) for c in lists["center"]],
                                lists["heading_cls"], lists["heading_res"], lists["size_cls"], lists["size_res"],
                                lists["rot_angle"], lists["score"], csv_path=csv_path,
                                records=np.concatenate(all_records)
                                if all_records and all(r is not None for r in all_records) else None)
        if idx_path is not None:
            with open(idx_path) as f:
                fill_files(os.path.join(result_dir, "data"), [line.rstrip() + ".txt" for line in f])
    return lists
