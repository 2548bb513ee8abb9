"""Detector results -> oriented boxes: the arithmetic of helpers/parse_pointnet_output.py
without the Lyft SDK (its `Box`, `Quaternion` and `LyftDataset` classes are not part of the
reference tree).  A box is a plain dict {center (3,), size [w,l,h], orientation
(w,x,y,z), score, name}; the sensor -> world step of :160 (an SDK lookup by
sample_data_token) takes the 4x4 matrix from the caller instead.
"""
import io

import numpy as np

from ..model_util import NUM_HEADING_BIN
from ..provider import _ListUnpickler, class2angle, class2size


def rotate_pc_along_y(pc, rot_angle):
    """helpers/parse_pointnet_output.py:16-30: rotated COPY of pc (N,C) (x,z channels)."""
    npc = np.copy(pc)
    cosval, sinval = np.cos(rot_angle), np.sin(rot_angle)
    rotmat = np.array([[cosval, -sinval], [sinval, cosval]])
    npc[:, [0, 2]] = np.dot(npc[:, [0, 2]], np.transpose(rotmat))
    return npc


def get_heading_angle(heading_cls, heading_res, rot_angle):
    """:102-106."""
    return class2angle(heading_cls, heading_res, NUM_HEADING_BIN) + rot_angle


def get_size(size_cls, size_res):
    """:109-118 -> np.ndarray([l, w, h])."""
    return class2size(size_cls, size_res)


def get_center_in_sensor_coord(center_coord, rot_angle):
    """:121-124: the centre rotated back out of the frustum frame."""
    return rotate_pc_along_y(np.expand_dims(center_coord, 0), rot_angle=-rot_angle).squeeze()


def get_center_in_world_coord(center_in_sensor_coord, sensor_to_world):
    """:127-138 with the 4x4 sensor -> world matrix given by the caller."""
    h = np.concatenate((center_in_sensor_coord, np.ones(1)))
    return np.dot(sensor_to_world, h).ravel()[0:3]


def quaternion_from_axis_angle(axis, angle):
    """Unit quaternion (w,x,y,z) of a rotation by `angle` about `axis` (normalised here)."""
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate(([np.cos(angle / 2.0)], axis * np.sin(angle / 2.0)))


def quaternion_multiply(a, b):
    """Hamilton product a * b (apply b first, then a)."""
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def quaternion_from_rotation_matrix(m):
    """(w,x,y,z) of a proper rotation matrix (largest-component branch, w >= 0 where it can)."""
    m = np.asarray(m, np.float64)
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
        q = np.zeros(4)
        q[0] = (m[k, j] - m[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (m[j, i] + m[i, j]) / s
        q[1 + k] = (m[k, i] + m[i, k]) / s
    return q / np.linalg.norm(q)


def get_box_from_inference(heading_cls, heading_res, rot_angle, size_cls, size_res, center_coord, score,
                           type_name=None, sensor_to_world=None):
    """:141-166.  The box starts heading along +x with y left and z up; it is turned by
    90 degrees about x and by the heading angle about -y to stand in the camera frame
    (z forward, y down), sizes are stored as [w, l, h], and the centre leaves the frustum
    frame through a rotation by -(rot_angle + pi/2).  With `sensor_to_world` (4x4) the
    centre and the orientation are carried on into the world frame."""
    heading_angle = get_heading_angle(heading_cls, heading_res, rot_angle)
    l, w, h = get_size(size_cls, size_res)
    center = get_center_in_sensor_coord(center_coord=np.asarray(center_coord, np.float64),
                                        rot_angle=rot_angle + np.pi / 2)
    first_rot = quaternion_from_axis_angle([1, 0, 0], np.pi / 2)
    second_rot = quaternion_from_axis_angle([0, -1, 0], -heading_angle)
    orientation = quaternion_multiply(second_rot, first_rot)
    if sensor_to_world is not None:
        m = np.asarray(sensor_to_world, np.float64)
        center = get_center_in_world_coord(center, m)
        orientation = quaternion_multiply(quaternion_from_rotation_matrix(m[:3, :3]), orientation)
    return {"center": center, "size": [w, l, h], "orientation": orientation, "score": score, "name": type_name}


def read_frustum_pointnet_output(inference_pickle_file, from_rgb_detection=True, type_names=None,
                                 sensor_to_world=None):
    """:55-99 without the token file and the SDK look-ups: the result lists of the
    detection driver (ps, [seg,] segp, center, heading_cls, heading_res, size_cls,
    size_res, rot_angle, score [, onehot]) -> list of boxes.  The file is read with an
    unpickler that constructs nothing but lists, numbers, strings and numpy arrays."""
    with open(inference_pickle_file, "rb") as fp:
        buf = io.BytesIO(fp.read())

    def load():
        return _ListUnpickler(buf, encoding="latin1").load()
    ps_list = load()
    if not from_rgb_detection:
        load()                      # seg_list (ground truth labels)
    load()                          # segp_list
    center_list, heading_cls_list, heading_res_list = load(), load(), load()
    size_cls_list, size_res_list, rot_angle_list, score_list = load(), load(), load(), load()
    boxes = []
    for i in range(len(ps_list)):
        m = None if sensor_to_world is None else sensor_to_world[i]
        boxes.append(get_box_from_inference(heading_cls_list[i], heading_res_list[i], rot_angle_list[i],
                                            size_cls_list[i], size_res_list[i], center_list[i], score_list[i],
                                            None if type_names is None else type_names[i], m))
    return boxes


__all__ = ["rotate_pc_along_y", "get_heading_angle", "get_size", "get_center_in_sensor_coord",
           "get_center_in_world_coord", "get_box_from_inference", "read_frustum_pointnet_output",
           "quaternion_from_axis_angle", "quaternion_multiply", "quaternion_from_rotation_matrix"]
