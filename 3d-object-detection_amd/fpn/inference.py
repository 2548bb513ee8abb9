"""Post-processing of the detector outputs: counterpart of the arithmetic of
scripts/test.py:inference (:122-162).  Takes the six `det_model` outputs and returns
the reference's seven arrays.  Like the reference this runs on the host in numpy
float64 buffers (np.zeros default dtype, test.py:125-131); it is per-frustum scalar
work, not part of the device path."""
import numpy as np


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


def write_detection_results(result_dir, id_list, type_list, box2d_list, center_list, heading_cls_list,
                            heading_res_list, size_cls_list, size_res_list, rot_angle_list, score_list,
                            csv_path="results_output.csv"):
    """Write <result_dir>/data/<id>.txt label files and the id -> lines CSV
    (scripts/test.py:51-90; the reference always writes the CSV into the working
    directory, pass csv_path=None to skip it)."""
    import csv
    import os
    if result_dir is None:
        return
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
    for data_dict in test_loader:
        pc = data_dict["frustum_point_cloud"]
        bs = batch_size or pc.shape[0]
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
                                lists["rot_angle"], lists["score"], csv_path=csv_path)
        if idx_path is not None:
            with open(idx_path) as f:
                fill_files(os.path.join(result_dir, "data"), [line.rstrip() + ".txt" for line in f])
    return lists
