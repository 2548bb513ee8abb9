"""Box decoding helpers: the step right after the network, which turns the
discretised heading / size outputs back into metric boxes.  Counterparts of the
evaluation helpers of scripts/provider.py (rotate_pc_along_y :35, angle2class :51,
class2angle :73, size2class :85, class2size :101, get_3d_box :282,
from_prediction_to_label_format :366).  Host-side numpy, per-box scalar work exactly
as in the reference; every function also accepts arrays and then works element-wise.

FrustumDataset (:107-275) reads the pickled frustum lists of the extraction step and
returns resampled, centre-rotated frustums with their labels; it draws from numpy's
global generator in the reference's order (resampling choice, flip, shift), so the same
np.random.seed gives the same samples.

Not provided: compute_box3d_iou (:313), which calls a `box3d_iou` whose import the
reference has commented out (:20) and therefore cannot run there either; box overlaps
are in helpers.mAP_evaluation.
"""
import io
import pickle

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


class _ListUnpickler(pickle.Unpickler):
    """The frustum files hold lists of strings, numbers and numpy arrays; nothing else is
    allowed to be constructed while reading them."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.numeric", "_frombuffer"),
                ("numpy._core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("frustum file refers to %s.%s: only lists of strings, numbers and numpy "
                                     "arrays are accepted" % (module, name))


def load_frustum_lists(path, count):
    """The `count` consecutive pickled lists of a frustum file (latin1, as the reference
    opens the Python-2 era files, :143-162)."""
    with open(path, "rb") as fp:
        data = fp.read()
    buf = io.BytesIO(data)
    return [_ListUnpickler(buf, encoding="latin1").load() for _ in range(count)]


class FrustumDataset(object):
    """scripts/provider.py:107-275.  Items: with from_rgb_detection (point_set, rot_angle,
    prob[, one_hot_vec]); otherwise (point_set, seg, box3d_center, angle_class,
    angle_residual, size_class, size_residual, rot_angle[, one_hot_vec]).
    `lists`, if given, supplies the frustum lists directly instead of a pickle file."""

    def __init__(self, npoints, split, batch_size=32, random_flip=False, random_shift=False, rotate_to_center=False,
                 overwritten_data_path=None, from_rgb_detection=False, one_hot=False, lists=None):
        self.npoints = npoints
        self.batch_size = batch_size
        self.random_flip = random_flip
        self.random_shift = random_shift
        self.rotate_to_center = rotate_to_center
        self.one_hot = one_hot
        self.from_rgb_detection = from_rgb_detection
        if lists is None:
            if overwritten_data_path is None:
                raise ValueError("no frustum file: pass overwritten_data_path (the reference's default, "
                                 "kitti/frustum_carpedcyc_%s.pickle next to its sources, does not exist here)" % split)
            lists = load_frustum_lists(overwritten_data_path, 6 if from_rgb_detection else 9)
        if from_rgb_detection:
            (self.id_list, self.box2d_list, self.input_list, self.type_list, self.frustum_angle_list,
             self.prob_list) = lists
        else:
            (self.id_list, self.box2d_list, self.box3d_list, self.input_list, self.label_list, self.type_list,
             self.heading_list, self.size_list, self.frustum_angle_list) = lists

    def __len__(self):
        return len(self.input_list)

    def get_center_view_rot_angle(self, index):
        """Frustum angle shifted by pi/2 so that it also corrects the heading (:242-245)."""
        return np.pi / 2.0 + self.frustum_angle_list[index]

    def get_box3d_center(self, index):
        return (self.box3d_list[index][0, :] + self.box3d_list[index][6, :]) / 2.0

    def get_center_view_box3d_center(self, index):
        c = (self.box3d_list[index][0, :] + self.box3d_list[index][6, :]) / 2.0
        return rotate_pc_along_y(np.expand_dims(c, 0), self.get_center_view_rot_angle(index)).squeeze()

    def get_center_view_box3d(self, index):
        return rotate_pc_along_y(np.copy(self.box3d_list[index]), self.get_center_view_rot_angle(index))

    def get_center_view_point_set(self, index):
        return rotate_pc_along_y(np.copy(self.input_list[index]), self.get_center_view_rot_angle(index))

    def __getitem__(self, index):
        rot_angle = self.get_center_view_rot_angle(index)
        if self.one_hot:
            cls_type = self.type_list[index]
            assert cls_type in g_type2onehotclass, cls_type
            one_hot_vec = np.zeros(3)
            one_hot_vec[g_type2onehotclass[cls_type]] = 1
        point_set = self.get_center_view_point_set(index) if self.rotate_to_center else self.input_list[index]
        # resample to npoints: with replacement only when there are too few (:186-191)
        n = point_set.shape[0]
        choice = np.random.choice(n, self.npoints, replace=n < self.npoints)
        point_set = point_set[choice, :]
        if self.from_rgb_detection:
            if self.one_hot:
                return point_set, rot_angle, self.prob_list[index], one_hot_vec
            return point_set, rot_angle, self.prob_list[index]
        seg = self.label_list[index][choice]
        box3d_center = self.get_center_view_box3d_center(index) if self.rotate_to_center else self.get_box3d_center(index)
        heading_angle = self.heading_list[index] - rot_angle if self.rotate_to_center else self.heading_list[index]
        size_class, size_residual = size2class(self.size_list[index], self.type_list[index])
        if self.random_flip and np.random.random() > 0.5:       # :219-225
            point_set[:, 0] *= -1
            box3d_center[0] *= -1
            heading_angle = np.pi - heading_angle
        if self.random_shift:                                     # :226-230, the reference's formula as written
            dist = np.sqrt(np.sum(box3d_center[0] ** 2 + box3d_center[1] ** 2))
            shift = np.clip(np.random.randn() * dist * 0.05, dist * 0.8, dist * 1.2)
            point_set[:, 2] += shift
            box3d_center[2] += shift
        angle_class, angle_residual = angle2class(heading_angle, NUM_HEADING_BIN)
        if self.one_hot:
            return (point_set, seg, box3d_center, angle_class, angle_residual, size_class, size_residual, rot_angle,
                    one_hot_vec)
        return point_set, seg, box3d_center, angle_class, angle_residual, size_class, size_residual, rot_angle

    def batch(self, indices, device, channels=None):
        """Items `indices` as get_model inputs: point_cloud (B,npoints,C) float32 and, with
        one_hot, one_hot (B,3) float32 on `device`, plus the list of the items' other
        fields."""
        import torch
        items = [self[i] for i in indices]
        pc = np.stack([it[0][:, :channels] for it in items]).astype(np.float32)
        out = [torch.from_numpy(pc).to(device)]
        if self.one_hot:
            out.append(torch.from_numpy(np.stack([it[-1] for it in items]).astype(np.float32)).to(device))
        return tuple(out) + ([it[1:] for it in items],)


def from_prediction_to_label_format(center, angle_class, angle_res, size_class, size_res, rot_angle):
    """Network outputs of one frustum -> KITTI label fields (h,w,l,tx,ty,tz,ry)
    (:366-373): the centre is rotated back out of the frustum frame and moved from
    the box centre to the bottom face (ty += h/2)."""
    l, w, h = class2size(size_class, size_res)
    ry = class2angle(angle_class, angle_res, NUM_HEADING_BIN) + rot_angle
    tx, ty, tz = rotate_pc_along_y(np.expand_dims(center, 0), -rot_angle).squeeze()
    ty += h / 2.0
    return h, w, l, tx, ty, tz, ry
