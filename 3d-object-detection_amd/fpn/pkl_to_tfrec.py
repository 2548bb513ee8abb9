"""Pickled frustum lists -> record files: counterpart of dataset/pkl_to_tfrec.py
(tfrecGen_test :130-220, tfrec_Gen_Train_Val :223-367) without TensorFlow; the records
are written by fpn.frustum_records in the same container and message format.

This is the Lyft variant of the tables (lower-case type names, its own mean sizes,
:9-24), which differs from models/model_util.py's KITTI tables.  Random draws come from
numpy's global generator in the reference's order.
"""
import numpy as np

from . import frustum_records as R
from .provider import angle2class, load_frustum_lists, rotate_pc_along_y

NUM_HEADING_BIN = 12
NUM_OBJECT_POINT = 512
NUM_SIZE_CLUSTER = 8

g_type2class = {'car': 0, 'Van': 1, 'Truck': 2, 'pedestrian': 3,
                'Person_sitting': 4, 'bicycle': 5, 'Tram': 6, 'Misc': 7}
g_class2type = {g_type2class[t]: t for t in g_type2class}
g_type2onehotclass = {'car': 0, 'pedestrian': 1, 'bicycle': 2}
g_type_mean_size = {'car': np.array([4.76, 1.93, 1.72]),
                    'Van': np.array([5.06763659, 1.9007158, 2.20532825]),
                    'Truck': np.array([10.13586957, 2.58549199, 3.2520595]),
                    'pedestrian': np.array([0.81, 0.77, 1.78]),
                    'Person_sitting': np.array([0.80057803, 0.5983815, 1.27450867]),
                    'bicycle': np.array([1.76, 0.63, 1.44]),
                    'Tram': np.array([16.17150617, 2.53246914, 3.53079012]),
                    'Misc': np.array([3.64300781, 1.54298177, 1.92320313])}
g_mean_size_arr = np.stack([g_type_mean_size[g_class2type[i]] for i in range(NUM_SIZE_CLUSTER)])


def size2class(size, type_name):
    """:108-121"""
    return g_type2class[type_name], size - g_type_mean_size[type_name]


def class2size(pred_cls, residual):
    """:124-127"""
    return g_type_mean_size[g_class2type[pred_cls]] + residual


class _Base(object):
    npoints = 1024
    random_flip = True
    random_shift = True
    rotate_to_center = True
    one_hot = True
    from_rgb_detection = False

    def get_center_view_rot_angle(self, index):
        return np.pi / 2.0 + self.frustum_angle_list[index]

    def get_center_view_point_set(self, index):
        return rotate_pc_along_y(np.copy(self.input_list[index]), self.get_center_view_rot_angle(index))

    def _resampled(self, index):
        point_set = self.get_center_view_point_set(index) if self.rotate_to_center else self.input_list[index]
        n = point_set.shape[0]
        choice = np.random.choice(n, self.npoints, replace=n < self.npoints)
        return point_set[choice, :3], choice

    def write_tfrec(self, tfrec_name):
        return R.write_records(tfrec_name, (self.serialize_example(*self.feature_extraction(i))
                                            for i in range(len(self.id_list))))


class tfrecGen_test(_Base):
    """Detection-time records (:130-220): point set, rotation angle, 2-D detector score,
    one-hot class, class index, sample token, 2-D box."""

    def __init__(self, pickle_path=None, lists=None):
        self.g_type2onehotclass = g_type2onehotclass
        if lists is None:
            lists = load_frustum_lists(pickle_path, 6)
        (self.id_list, self.box2d_list, self.input_list, self.type_list, self.frustum_angle_list,
         self.prob_list) = lists

    def feature_extraction(self, index):
        rot_angle = self.get_center_view_rot_angle(index)
        cls_type = self.type_list[index]
        assert cls_type in g_type2onehotclass, cls_type
        cls_index = g_type2onehotclass[cls_type]
        one_hot_vec = np.zeros(3, dtype=int)
        one_hot_vec[cls_index] = 1
        point_set, _ = self._resampled(index)
        return (point_set, rot_angle, self.prob_list[index], one_hot_vec, cls_index, self.id_list[index],
                self.box2d_list[index])

    def serialize_example(self, point_set, rot_angle, prob_value, one_hot_vec, cls_index, scene_id, box_2D):
        return R.serialize_test_example(point_set, rot_angle, prob_value, one_hot_vec, cls_index, scene_id, box_2D)


class tfrec_Gen_Train_Val(_Base):
    """Training / validation records (:223-367) with the reference's augmentation."""

    def __init__(self, pickle_path=None, lists=None):
        if lists is None:
            lists = load_frustum_lists(pickle_path, 9)
        (self.id_list, self.box2d_list, self.box3d_list, self.input_list, self.label_list, self.type_list,
         self.heading_list, self.size_list, self.frustum_angle_list) = lists

    def get_box3d_center(self, index):
        return (self.box3d_list[index][0, :] + self.box3d_list[index][6, :]) / 2.0

    def get_center_view_box3d_center(self, index):
        return rotate_pc_along_y(np.expand_dims(self.get_box3d_center(index), 0),
                                 self.get_center_view_rot_angle(index)).squeeze()

    def feature_extraction(self, index):
        rot_angle = self.get_center_view_rot_angle(index)
        cls_type = self.type_list[index]
        assert cls_type in g_type2onehotclass, cls_type
        one_hot_vec = np.zeros(3, dtype=int)
        one_hot_vec[g_type2onehotclass[cls_type]] = 1
        point_set, choice = self._resampled(index)
        seg = self.label_list[index][choice]
        box3d_center = self.get_center_view_box3d_center(index) if self.rotate_to_center else self.get_box3d_center(index)
        heading_angle = self.heading_list[index] - rot_angle if self.rotate_to_center else self.heading_list[index]
        size_class, size_residual = size2class(self.size_list[index], cls_type)
        if self.random_flip and np.random.random() > 0.5:
            point_set[:, 0] *= -1
            box3d_center[0] *= -1
            heading_angle = np.pi - heading_angle
        if self.random_shift:
            dist = np.sqrt(np.sum(box3d_center[0] ** 2 + box3d_center[1] ** 2))
            shift = np.clip(np.random.randn() * dist * 0.05, dist * 0.8, dist * 1.2)
            point_set[:, 2] += shift
            box3d_center[2] += shift
        angle_class, angle_residual = angle2class(heading_angle, NUM_HEADING_BIN)
        return (point_set, seg, box3d_center, angle_class, angle_residual, size_class, size_residual, rot_angle,
                one_hot_vec)

    def serialize_example(self, point_set, seg, box3d_center, angle_class, angle_residual, size_class, size_residual,
                          rot_angle, one_hot_vec):
        return R.serialize_train_example(point_set, seg, box3d_center, angle_class, angle_residual, size_class,
                                         size_residual, rot_angle, one_hot_vec)
