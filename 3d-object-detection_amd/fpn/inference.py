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
