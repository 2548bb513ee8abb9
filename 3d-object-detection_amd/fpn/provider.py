"""Box decoding helpers: the step right after the network, which turns the
discretised heading / size outputs back into metric boxes.  Counterparts of the
evaluation helpers of scripts/provider.py (rotate_pc_along_y :35, angle2class :51,
class2angle :73, size2class :85, class2size :101, get_3d_box :282,
from_prediction_to_label_format :366).  Host-side numpy, per-box scalar work exactly
as in the reference; every function also accepts arrays and then works element-wise.

Not provided: FrustumDataset (:107, a Keras Sequence over pickled KITTI frustums) and
compute_box3d_iou (:313), which calls a `box3d_iou` whose import the reference has
commented out (:20) and therefore cannot run there either; box overlaps are in
helpers.mAP_evaluation.
"""
import numpy as np

from .model_util import (NUM_HEADING_BIN, NUM_SIZE_CLUSTER, g_class2type, g_type2class,  # noqa: F401
                         g_type2onehotclass, g_type_mean_size)

TWO_PI = 2 * np.pi


def rotate_pc_along_y(pc, rot_angle):
    """Rotate the x,z channels of pc (N,C) in place by rot_angle about the y axis
    (z forward, x left, y down) and return pc (:35-48)."""
    c, s = np.cos(rot_angle), np.sin(rot_angle)
    # row vector times transpose([[c,-s],[s,c]])
    pc[:, [0, 2]] = np.dot(pc[:, [0, 2]], np.array([[c, s], [-s, c]]))
    return pc


def angle2class(angle, num_class):
    """Angle (rad) -> (bin index, residual) with bin centres at k * 2pi/num_class
    (:51-70): bin*(2pi/N) + residual == angle (mod 2pi)."""
    angle = angle % TWO_PI
    assert 0 <= angle <= TWO_PI
    per_class = TWO_PI / float(num_class)
    shifted = (angle + per_class / 2) % TWO_PI
    class_id = int(shifted / per_class)
    return class_id, shifted - (class_id * per_class + per_class / 2)


def class2angle(pred_cls, residual, num_class, to_label_format=True):
    """Inverse of angle2class (:73-82); with to_label_format an angle above pi is
    moved down by 2pi into the labels' (-pi, pi] range."""
    angle = pred_cls * (TWO_PI / float(num_class)) + residual
    if np.ndim(angle) == 0:
        if to_label_format and angle > np.pi:
            angle = angle - TWO_PI
        return angle
    return np.where(to_label_format & (angle > np.pi), angle - TWO_PI, angle)


def size2class(size, type_name):
    """(l,w,h) and a type name -> (size cluster index, residual to the cluster mean) (:85-98)."""
    return g_type2class[type_name], size - g_type_mean_size[type_name]


def class2size(pred_cls, residual):
    """Inverse of size2class (:101-104)."""
    return g_type_mean_size[g_class2type[pred_cls]] + residual


def get_3d_box(box_size, heading_angle, center):
    """(l,w,h), heading about y (clockwise from +x), centre -> corners (8,3) in the
    reference's order (:282-310): top face (y = +h/2) first."""
    l, w, h = box_size
    c, s = np.cos(heading_angle), np.sin(heading_angle)
    rot_y = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1]) * (l / 2)
    sy = np.array([1, 1, 1, 1, -1, -1, -1, -1]) * (h / 2)
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1]) * (w / 2)
    corners = np.dot(rot_y, np.vstack([sx, sy, sz]))
    corners[0, :] = corners[0, :] + center[0]
    corners[1, :] = corners[1, :] + center[1]
    corners[2, :] = corners[2, :] + center[2]
    return np.transpose(corners)


def from_prediction_to_label_format(center, angle_class, angle_res, size_class, size_res, rot_angle):
    """Network outputs of one frustum -> KITTI label fields (h,w,l,tx,ty,tz,ry)
    (:366-373): the centre is rotated back out of the frustum frame and moved from
    the box centre to the bottom face (ty += h/2)."""
    l, w, h = class2size(size_class, size_res)
    ry = class2angle(angle_class, angle_res, NUM_HEADING_BIN) + rot_angle
    tx, ty, tz = rotate_pc_along_y(np.expand_dims(center, 0), -rot_angle).squeeze()
    ty += h / 2.0
    return h, w, l, tx, ty, tz, ry
