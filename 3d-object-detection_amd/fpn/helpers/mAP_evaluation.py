"""3-D detection mAP in the nuScenes/Lyft record format: the interface of the
reference's helpers/mAP_evaluation.py, with the box overlaps computed on the GPU.

Records are dictionaries with `sample_token`, `translation` (x,y,z), `size`
(width,length,height), `rotation` (quaternion w,x,y,z), `name` and, for predictions,
`score` (mAP_evaluation.py:10-25).  The public names and their results are the
reference's: Box3D (:50), group_by_key (:205), wrap_in_box (:212), get_envelope
(:220), get_ap (:234), get_ious (:259), recall_precision (:263),
get_average_precisions (:326), get_class_names (:376).

What differs is where the geometry runs.  The reference intersects ground rectangles
one pair at a time with shapely (:162-174, not installed here); this module packs
all boxes of a class into one float64 array and asks libfpn_hip.so for every
prediction's best-overlapping ground-truth box in a single launch
(fpn_box3d_best_overlap_f64).  The greedy true/false-positive marking that follows
is inherently sequential in score order and stays on the host.
"""
import argparse
import json
from collections import defaultdict

import numpy as np
import torch

from .. import _lib

_FIELDS = 10   # translation(3) size(3) rotation(4)


def _device(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("mAP_evaluation computes box overlaps on a ROCm device and none is visible; "
                           "this package has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _iou_pairs(a, b, device=None):
    """a, b: (n,10) or (10,) float64 arrays (a single box is broadcast) -> (n,) IoU."""
    dev = _device(device)
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, _FIELDS)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, _FIELDS)
    n = max(len(a), len(b))
    if len(a) not in (1, n) or len(b) not in (1, n):
        raise ValueError("box arrays of %d and %d rows do not pair up" % (len(a), len(b)))
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fpn_box3d_iou_f64(n, _lib.ptr(ta), 0 if len(a) == 1 and n > 1 else _FIELDS,
                                                 _lib.ptr(tb), 0 if len(b) == 1 and n > 1 else _FIELDS,
                                                 _lib.ptr(out), _lib.stream_of(out)))
    return out.cpu().numpy()


def _intersection_pair(a, b, device=None):
    """(ground-rectangle overlap area, height overlap) of two packed boxes, float64, through
    fpn_box3d_intersection_f64."""
    dev = _device(device)
    ta = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(1, _FIELDS)).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64).reshape(1, _FIELDS)).to(dev)
    out = torch.empty(2, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fpn_box3d_intersection_f64(1, _lib.ptr(ta), _FIELDS, _lib.ptr(tb), _FIELDS,
                                                          _lib.ptr(out[0:1]), _lib.ptr(out[1:2]), _lib.stream_of(out)))
    area, height = out.cpu().tolist()
    return area, height


class Box3D:
    """A prediction or a ground-truth box (mAP_evaluation.py:50-202).  Keeps the
    reference's attributes; `ground_bbox_coords` is the (4,2) corner array of the
    base rectangle instead of a shapely Polygon."""

    def __init__(self, **kwargs):
        sample_token = kwargs["sample_token"]
        translation = kwargs["translation"]
        size = kwargs["size"]
        rotation = kwargs["rotation"]
        name = kwargs["name"]
        self.score = kwargs.get("score", -1)
        # the reference's input checks, same messages (:62-73, :82)
        assert type(sample_token) == str, "Error: sample_token must be a string!"
        for what, value, count in (("Translation", translation, 3), ("Size", size, 3), ("Rotation", rotation, 4)):
            assert len(value) == count, "Error: %s must have %d elements!" % (what, count)
            assert not np.any(np.isnan(value)), "Error: %s may not be NaN!" % what
        assert name is not None, "Error: detection_name cannot be empty!"
        assert np.all([x > 0 for x in size])
        self.sample_token, self.translation, self.size = sample_token, translation, size
        self.rotation, self.name = rotation, name
        self.volume = np.prod(self.size)
        self.width, self.length, self.height = size
        self.center_x, self.center_y, self.center_z = translation
        self.min_z = self.center_z - self.height / 2
        self.max_z = self.center_z + self.height / 2
        self.ground_bbox_coords = None
        self.ground_bbox_coords = self.get_ground_bbox_coords()

    def packed(self):
        """The box as the 10 doubles the library reads."""
        return np.array(list(self.translation) + list(self.size) + list(self.rotation), dtype=np.float64)

    @staticmethod
    def check_orthogonal(a, b, c):
        """(b - a) is orthogonal to (c - a) (:98-101)."""
        return np.isclose((b[0] - a[0]) * (c[0] - a[0]) + (b[1] - a[1]) * (c[1] - a[1]), 0)

    def _heading(self):
        """rotation_matrix[0,0], rotation_matrix[0,1] of the (normalised) quaternion,
        the two numbers the reference takes from pyquaternion (:117-120)."""
        w, x, y, z = (float(v) for v in self.rotation)
        ss = ((w * w + x * x) + y * y) + z * z
        if not abs(1.0 - ss) < 1e-14:
            n = np.sqrt(ss)
            if n > 0:
                w, x, y, z = w / n, x / n, y / n, z / n
        return ((x * x + w * w) + (-z) * z) + y * (-y), ((x * y + w * (-z)) + (-z) * w) + y * x

    def get_ground_bbox_coords(self):
        if self.ground_bbox_coords is not None:
            return self.ground_bbox_coords
        return self.calculate_ground_bbox_coords()

    def calculate_ground_bbox_coords(self):
        """Corners of the base rectangle, which the reference assumes parallel to the
        ground, in its order (:122-132), with its orthogonality checks (:139-142)."""
        if self.ground_bbox_coords is not None:
            return self.ground_bbox_coords
        c, s = self._heading()
        hl, hw = self.length / 2, self.width / 2
        pts = [(self.center_x + sl * hl * c + sw * hw * s, self.center_y + sl * hl * s - sw * hw * c)
               for sl, sw in ((1, 1), (1, -1), (-1, -1), (-1, 1))]
        for i in range(4):
            assert self.check_orthogonal(pts[i], pts[(i + 1) % 4], pts[(i + 3) % 4])
        self.ground_bbox_coords = np.array(pts, dtype=np.float64)
        return self.ground_bbox_coords

    def get_height_intersection(self, other):
        return max(0, min(other.max_z, self.max_z) - max(other.min_z, self.min_z))

    def get_area_intersection(self, other, device=None):
        """Area of the overlap of the two ground rectangles (:162-167), with the
        reference's sanity assertion."""
        result, _ = _intersection_pair(self.packed(), other.packed(), device)
        # the reference's check is `result <= self.width * self.length` (:166); the overlap is the
        # shoelace area of a clipped polygon and, for a box against itself or inside another,
        # IS the rectangle's area up to rounding.  A shoelace sum adds products of coordinates,
        # so its rounding error scales with sum |x_i y_j| (the box's distance from the origin
        # squared), not with the area: the check allows exactly that much, nothing more.
        x, y = self.ground_bbox_coords[:, 0], self.ground_bbox_coords[:, 1]
        own = 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))
        slack = 16 * np.finfo(np.float64).eps * float(np.dot(np.abs(x), np.abs(np.roll(y, -1))) + np.dot(np.abs(y), np.abs(np.roll(x, -1))))
        assert result <= max(own, self.width * self.length) + slack
        return result

    def get_intersection(self, other, device=None):
        """Volume of the intersection: height overlap x ground overlap (:169-174)."""
        area, _ = _intersection_pair(self.packed(), other.packed(), device)
        return self.get_height_intersection(other) * area

    def get_iou(self, other, device=None):
        return _iou_pairs(self.packed(), other.packed(), device)[0]

    def serialize(self):
        return {"sample_token": self.sample_token, "translation": self.translation, "size": self.size,
                "rotation": self.rotation, "name": self.name, "volume": self.volume, "score": self.score}

    def __repr__(self):
        return str(self.serialize())


def group_by_key(detections, key):
    groups = defaultdict(list)
    for detection in detections:
        groups[detection[key]].append(detection)
    return groups


def wrap_in_box(input):
    return {key: [Box3D(**x) for x in value] for key, value in input.items()}


def get_envelope(precisions):
    """Running maximum from the right, in place (:220-231)."""
    precisions[:] = np.maximum.accumulate(precisions[::-1])[::-1]
    return precisions


def get_ap(recalls, precisions):
    """Area under the precision envelope at the points where recall changes (:234-256)."""
    recalls = np.concatenate(([0.0], recalls, [1.0]))
    precisions = get_envelope(np.concatenate(([0.0], precisions, [0.0])))
    step = np.where(recalls[1:] != recalls[:-1])[0]
    return np.sum((recalls[step + 1] - recalls[step]) * precisions[step + 1])


def get_ious(gt_boxes, predicted_box, device=None):
    """IoU of one predicted box with each ground-truth box (:259-260), one launch."""
    if len(gt_boxes) == 0:
        return []
    return list(_iou_pairs(predicted_box.packed(), np.stack([g.packed() for g in gt_boxes]), device))


def _pack_records(records):
    for r in records:
        Box3D(**r)          # the reference validates every record it touches (:215, :278)
    if not records:
        return np.zeros((0, _FIELDS))
    return np.array([list(r["translation"]) + list(r["size"]) + list(r["rotation"]) for r in records],
                    dtype=np.float64)


def recall_precision(gt, predictions, iou_threshold, device=None):
    """-> (recalls, precisions, ap) of one class (:263-323)."""
    dev = _device(device)
    num_gts = len(gt)
    # ground truth grouped by sample, flattened so that a sample's boxes are contiguous
    image_gts = group_by_key(gt, "sample_token")
    begin, flat = {}, []
    for token, boxes in image_gts.items():
        begin[token] = (len(flat), len(boxes))
        flat.extend(boxes)
    gt_packed = _pack_records(flat)
    gt_checked = np.zeros(len(flat))

    predictions = sorted(predictions, key=lambda x: x["score"], reverse=True)      # :270
    num_predictions = len(predictions)
    pred_packed = _pack_records(predictions)
    spans = np.array([begin.get(p["sample_token"], (0, 0)) for p in predictions], dtype=np.int32).reshape(-1, 2)

    best = np.full(num_predictions, -np.inf)
    jmax = np.full(num_predictions, -1, dtype=np.int32)
    if num_predictions:
        t_pred = torch.from_numpy(pred_packed).to(dev)
        t_gt = torch.from_numpy(gt_packed).to(dev) if len(flat) else None
        t_begin = torch.from_numpy(np.ascontiguousarray(spans[:, 0])).to(dev)
        t_count = torch.from_numpy(np.ascontiguousarray(spans[:, 1])).to(dev)
        t_best = torch.empty(num_predictions, dtype=torch.float64, device=dev)
        t_j = torch.empty(num_predictions, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().fpn_box3d_best_overlap_f64(
                num_predictions, _lib.ptr(t_pred), _lib.ptr(t_gt), _lib.ptr(t_begin), _lib.ptr(t_count),
                _lib.ptr(t_best), _lib.ptr(t_j), _lib.stream_of(t_pred)))
        best, jmax = t_best.cpu().numpy(), t_j.cpu().numpy()

    # go down the detections in score order and mark true / false positives (:277-306)
    tp = np.zeros(num_predictions)
    fp = np.zeros(num_predictions)
    for i in range(num_predictions):
        if best[i] > iou_threshold:
            slot = spans[i, 0] + jmax[i]
            if gt_checked[slot] == 0:
                tp[i] = 1.0
                gt_checked[slot] = 1
            else:
                fp[i] = 1.0
        else:
            fp[i] = 1.0

    fp = np.cumsum(fp, axis=0)
    tp = np.cumsum(tp, axis=0)
    recalls = tp / float(num_gts)
    assert np.all(0 <= recalls) & np.all(recalls <= 1)
    precisions = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    assert np.all(0 <= precisions) & np.all(precisions <= 1)
    return recalls, precisions, get_ap(recalls, precisions)


def get_average_precisions(gt, predictions, class_names, iou_threshold, device=None):
    """Average precision per class name, 0 for a class without predictions (:326-373)."""
    assert 0 <= iou_threshold <= 1
    gt_by_class_name = group_by_key(gt, "name")
    pred_by_class_name = group_by_key(predictions, "name")
    average_precisions = np.zeros(len(class_names))
    for class_id, class_name in enumerate(class_names):
        if class_name in pred_by_class_name:
            average_precisions[class_id] = recall_precision(
                gt_by_class_name[class_name], pred_by_class_name[class_name], iou_threshold, device)[2]
    return average_precisions


def get_class_names(gt):
    return sorted(list(set([x["name"] for x in gt])))


def main(argv=None):
    """The reference's command line (:388-415)."""
    parser = argparse.ArgumentParser()
    parser.add_argument("-p", "--pred_file", type=str, required=True, help="Path to the predictions file.")
    parser.add_argument("-g", "--gt_file", type=str, required=True, help="Path to the ground truth file.")
    parser.add_argument("-t", "--iou_threshold", type=float, default=0.5, help="iou threshold")
    args = parser.parse_args(argv)
    with open(args.pred_file) as f:
        predictions = json.load(f)
    with open(args.gt_file) as f:
        gt = json.load(f)
    class_names = get_class_names(gt)
    print("Class_names = ", class_names)
    average_precisions = get_average_precisions(gt, predictions, class_names, args.iou_threshold)
    print("Average per class mean average precision = ", np.mean(average_precisions))
    for class_id in sorted(list(zip(class_names, average_precisions.flatten().tolist()))):
        print(class_id)


if __name__ == "__main__":
    main()
