"""TEST INFRASTRUCTURE ONLY - CPU restatement of the detection evaluation of the
reference's helpers/mAP_evaluation.py, used by tests/ as the checker of the GPU
overlap kernels and of the host-side mAP code.  Nothing in the product imports it.

Parity status.  The reference gets its geometry from two third-party packages that
are not installed here and not vendored in /root/reference: shapely (Polygon
intersection area, mAP_evaluation.py:47,163,172; version unpinned, the repository has
no requirements file) and pyquaternion (rotation_matrix, :46,85,117; unpinned).  Their
published algorithms are restated: the intersection of two convex quadrilaterals by
Sutherland-Hodgman clipping plus the shoelace area, and the rotation matrix as the
product Q(q) Qbar(q)^T of the normalised quaternion.  The reference holds no golden
vectors for this file, so agreement with it to the last bit is PARITY UNPINNED; what
is pinned (tests/test_oracle_pinning.py) is a set of hand-computed cases whose exact
answers are known (axis-aligned overlaps, identical boxes, rotations by 90 degrees,
disjoint boxes, containment) and the AP arithmetic on hand-worked PR curves.

Everything is scalar Python float (IEEE double, no fused multiply-add), in the same
operation order as the GPU kernel, so the two agree bit for bit.
"""
import math
from collections import defaultdict

import numpy as np


def heading(rotation):
    """(rotation_matrix[0,0], rotation_matrix[0,1]) of pyquaternion.Quaternion(rotation)
    (mAP_evaluation.py:117-120): normalise unless |1 - |q|^2| < 1e-14 or |q| == 0."""
    w, x, y, z = (float(v) for v in rotation)
    ss = ((w * w + x * x) + y * y) + z * z
    if not abs(1.0 - ss) < 1e-14:
        n = math.sqrt(ss)
        if n > 0:
            w, x, y, z = w / n, x / n, y / n, z / n
    c = ((x * x + w * w) + (-z) * z) + y * (-y)
    s = ((x * y + w * (-z)) + (-z) * w) + y * x
    return c, s


def ground_corners(box):
    """box = 10 floats (translation, size w/l/h, rotation) -> 4 (x,y) corners, :122-132."""
    cx, cy = float(box[0]), float(box[1])
    width, length = float(box[3]), float(box[4])
    c, s = heading(box[6:10])
    hl, hw = length / 2, width / 2
    return [(cx + hl * c + hw * s, cy + hl * s - hw * c),
            (cx + hl * c - hw * s, cy + hl * s + hw * c),
            (cx - hl * c - hw * s, cy - hl * s + hw * c),
            (cx - hl * c + hw * s, cy - hl * s - hw * c)]


def _shoelace(poly):
    a = 0.0
    n = len(poly)
    for i in range(n):
        j = 0 if i + 1 == n else i + 1
        a = a + (poly[i][0] * poly[j][1] - poly[j][0] * poly[i][1])
    return a


def quad_intersection_area(p, q):
    """Area of the intersection of two convex quadrilaterals (lists of 4 (x,y))."""
    orient = 1.0 if _shoelace(q) >= 0 else -1.0
    poly = list(p)
    for e in range(4):
        if not poly:
            break
        ax, ay = q[e]
        bx, by = q[(e + 1) % 4]
        ex, ey = bx - ax, by - ay
        out = []
        n = len(poly)
        for i in range(n):
            cur, nxt = poly[i], poly[0 if i + 1 == n else i + 1]
            dc = orient * (ex * (cur[1] - ay) - ey * (cur[0] - ax))
            dn = orient * (ex * (nxt[1] - ay) - ey * (nxt[0] - ax))
            in_c, in_n = dc >= 0, dn >= 0
            if in_c and len(out) < 8:
                out.append(cur)
            if in_c != in_n and len(out) < 8:
                t = dc / (dc - dn)
                out.append((cur[0] + t * (nxt[0] - cur[0]), cur[1] + t * (nxt[1] - cur[1])))
        poly = out
    if len(poly) < 3:
        return 0.0
    return abs(_shoelace(poly)) / 2


def box_iou(a, b):
    """Box3D.get_iou, mAP_evaluation.py:156-182, for two packed boxes."""
    a = [float(v) for v in a]
    b = [float(v) for v in b]
    ga, gb = ground_corners(a), ground_corners(b)
    a_min, a_max = a[2] - a[5] / 2, a[2] + a[5] / 2
    b_min, b_max = b[2] - b[5] / 2, b[2] + b[5] / 2
    dz = min(b_max, a_max) - max(b_min, a_min)
    height = dz if dz > 0 else 0.0
    inter = height * quad_intersection_area(ga, gb)
    va, vb = (a[3] * a[4]) * a[5], (b[3] * b[4]) * b[5]
    union = (va + vb) - inter
    iou = inter / union
    return 0.0 if iou < 0 else (1.0 if iou > 1 else iou)


def pack(record):
    return [float(v) for v in list(record["translation"]) + list(record["size"]) + list(record["rotation"])]


def get_ap(recalls, precisions):
    """mAP_evaluation.py:220-256 with explicit loops."""
    r = [0.0] + [float(v) for v in recalls] + [1.0]
    p = [0.0] + [float(v) for v in precisions] + [0.0]
    for i in range(len(p) - 1, 0, -1):
        p[i - 1] = max(p[i - 1], p[i])
    ap = 0.0
    terms = [(r[i + 1] - r[i]) * p[i + 1] for i in range(len(r) - 1) if r[i + 1] != r[i]]
    return float(np.sum(np.array(terms))) if terms else 0.0


def recall_precision(gt, predictions, iou_threshold):
    """mAP_evaluation.py:263-323, one prediction at a time as the reference does it."""
    num_gts = len(gt)
    image_gts = defaultdict(list)
    for g in gt:
        image_gts[g["sample_token"]].append(g)
    checked = {k: [0] * len(v) for k, v in image_gts.items()}
    predictions = sorted(predictions, key=lambda x: x["score"], reverse=True)
    tp, fp = [], []
    for pred in predictions:
        boxes = image_gts.get(pred["sample_token"], [])
        max_overlap, jmax = -math.inf, -1
        if boxes:
            overlaps = [box_iou(pack(pred), pack(g)) for g in boxes]
            max_overlap = max(overlaps)
            jmax = overlaps.index(max_overlap)
        hit = False
        if max_overlap > iou_threshold and checked[pred["sample_token"]][jmax] == 0:
            checked[pred["sample_token"]][jmax] = 1
            hit = True
        tp.append(1.0 if hit else 0.0)
        fp.append(0.0 if hit else 1.0)
    tp = np.cumsum(np.array(tp, dtype=np.float64))
    fp = np.cumsum(np.array(fp, dtype=np.float64))
    recalls = tp / float(num_gts)
    precisions = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return recalls, precisions, get_ap(recalls, precisions)


def get_average_precisions(gt, predictions, class_names, iou_threshold):
    """mAP_evaluation.py:326-373."""
    out = np.zeros(len(class_names))
    for k, name in enumerate(class_names):
        preds = [p for p in predictions if p["name"] == name]
        if preds:
            out[k] = recall_precision([g for g in gt if g["name"] == name], preds, iou_threshold)[2]
    return out
