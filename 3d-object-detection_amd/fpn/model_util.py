"""Shared sub-graphs of the Frustum-PointNet models: counterpart of the inference
parts of models/model_util.py (constants :14-48, tf_gather_object_pc :55-86,
parse_output_to_tensors :211-243, point_cloud_masking :275-319,
get_center_regression_net :322-357).  The TFRecord parsers (:89-144) live in
fpn.frustum_records and are re-exported at the end of this module.  Losses and
box-corner helpers are training code and are not part of this path.
"""
import math

import numpy as np
import torch

from . import ops, tf_util

NUM_HEADING_BIN = 12     # model_util.py:14
NUM_SIZE_CLUSTER = 8     # model_util.py:15
NUM_OBJECT_POINT = 512   # model_util.py:16

g_type2class = {'Car': 0, 'Van': 1, 'Truck': 2, 'Pedestrian': 3,
                'Person_sitting': 4, 'Cyclist': 5, 'Tram': 6, 'Misc': 7}       # model_util.py:34-35
g_class2type = {g_type2class[t]: t for t in g_type2class}
g_type2onehotclass = {'Car': 0, 'Pedestrian': 1, 'Cyclist': 2}                # model_util.py:37
g_type_mean_size = {'Car': np.array([3.88311640418, 1.62856739989, 1.52563191462]),
                    'Van': np.array([5.06763659, 1.9007158, 2.20532825]),
                    'Truck': np.array([10.13586957, 2.58549199, 3.2520595]),
                    'Pedestrian': np.array([0.84422524, 0.66068622, 1.76255119]),
                    'Person_sitting': np.array([0.80057803, 0.5983815, 1.27450867]),
                    'Cyclist': np.array([1.76282397, 0.59706367, 1.73698127]),
                    'Tram': np.array([16.17150617, 2.53246914, 3.53079012]),
                    'Misc': np.array([3.64300781, 1.54298177, 1.92320313])}   # model_util.py:38-45
g_mean_size_arr = np.zeros((NUM_SIZE_CLUSTER, 3))
for _i in range(NUM_SIZE_CLUSTER):
    g_mean_size_arr[_i, :] = g_type_mean_size[g_class2type[_i]]               # model_util.py:46-48

_mean_size_cache = {}


def mean_size_tensor(device):
    t = _mean_size_cache.get(device)
    if t is None:
        t = torch.tensor(g_mean_size_arr, dtype=torch.float32, device=device)
        _mean_size_cache[device] = t
    return t


def tf_gather_object_pc(point_cloud, mask, npoints=512):
    """Reference: model_util.py:55-86.  Returns (object_pc (B,npoints,C), indices
    (B,npoints,2)) with indices[...,0] the batch index and indices[...,1] the
    point index, as tf.gather_nd consumes them.  The reference picks the points
    with numpy's global RNG inside a tf.py_func; here the pick is the
    deterministic one of fpn_mask_gather (first npoints positives in index order,
    cyclic repeats when there are fewer) and runs on the device.  `mask` is a
    (B,N) 0/1 tensor."""
    b, n, c = point_cloud.shape
    logits = torch.stack([torch.zeros_like(mask), mask], dim=2).contiguous()   # 0 < mask <=> picked
    zero_mean_pc = point_cloud.contiguous()
    _, _, sel, _ = ops.mask_gather(zero_mean_pc, logits, npoints, want_mask=False, want_sel=True)
    gather_idx = sel.long()[:, :, None].expand(b, npoints, c)
    object_pc = torch.gather(point_cloud, 1, gather_idx)
    batch_idx = torch.arange(b, device=sel.device, dtype=torch.int32)[:, None].expand(b, npoints)
    return object_pc, torch.stack([batch_idx, sel], dim=2)


def parse_output_to_tensors(output, end_points):
    """Reference: model_util.py:211-243: slice the (B,59) box vector."""
    nh, ns = NUM_HEADING_BIN, NUM_SIZE_CLUSTER
    b = output.shape[0]
    end_points['center_boxnet'] = output[:, 0:3]
    end_points['heading_scores'] = output[:, 3:3 + nh]
    hrn = output[:, 3 + nh:3 + 2 * nh]
    end_points['heading_residuals_normalized'] = hrn
    end_points['heading_residuals'] = hrn * (math.pi / nh)
    end_points['size_scores'] = output[:, 3 + 2 * nh:3 + 2 * nh + ns]
    srn = output[:, 3 + 2 * nh + ns:3 + 2 * nh + 4 * ns].reshape(b, ns, 3)
    end_points['size_residuals_normalized'] = srn
    end_points['size_residuals'] = srn * mean_size_tensor(output.device)[None]
    return end_points


def point_cloud_masking(point_cloud, logits, end_points, xyz_only=True):
    """Reference: model_util.py:275-319.  One device kernel computes the mask, the
    masked centroid, the object-point pick and the centred gather (the reference
    crosses to the host for the pick, :84).  Returns (object_point_cloud
    (B,512,3), mask_xyz_mean (B,3), end_points); additionally records the picked
    indices as end_points['object_point_indices']."""
    if not xyz_only:
        raise NotImplementedError("point_cloud_masking: only xyz_only=True is used by the models")
    mask, mean, sel, obj = ops.mask_gather(point_cloud.contiguous(), logits.contiguous(), NUM_OBJECT_POINT)
    end_points['mask'] = mask
    end_points['object_point_indices'] = sel
    return obj, mean, end_points


def _tnet_stack_fused(object_point_cloud, one_hot_vec):
    """The T-Net's three convolutions + max-pool + one-hot concat as one chain kernel
    (fpn_v1_obj_chain_f32 / _bf16) on the variables of the active store: (B,M,3) ->
    (B, 256 + 3).  Same variables and scopes as the layer-by-layer form."""
    from . import pointnet_util as U
    from . import variables
    store = variables.get_default_store()
    b, m, _ = object_point_cloud.shape
    scopes = ('conv-reg1-stage1', 'conv-reg2-stage1', 'conv-reg3-stage1')
    names = [variables.scoped(s) for s in scopes]
    cache = store.__dict__.setdefault("_tnet_chain_cache", {})
    key = (names[0], U.SA_PRECISION)
    plan = cache.get(key)
    if plan is None or plan["version"] != store.version:
        dims = ((3, 128), (128, 128), (128, 256))
        params = [U._layer_params(store, names[i], dims[i][0], dims[i][1], True) for i in range(3)]
        version = store.version
        (w1, s1, h1), (w2, s2, h2), (w3, s3, h3) = [(w.contiguous(), torch.ones_like(h) if s is None else s, h)
                                                    for w, s, h in params]
        if U.SA_PRECISION == "bf16":
            tensors = [w1, s1, h1, h2, h3]
            plan = {"version": version, "tensors": tensors, "table": ops.pointer_table(tensors),
                    "stream": ops.chain_stream_bf16([(w2, s2), (w3, s3)])}
        else:
            tensors = [w1, s1, h1, ops.tile_weights_reg(w2, 0), s2, h2, ops.tile_weights_reg(w3, 0), s3, h3]
            plan = {"version": version, "tensors": tensors, "table": ops.pointer_table(tensors)}
        cache[key] = plan
    tcat = torch.zeros((b, 256 + 3), dtype=torch.float32, device=object_point_cloud.device)
    tcat[:, 256:] = one_hot_vec
    obj2d = object_point_cloud.contiguous().reshape(b * m, 3)
    if "stream" in plan:
        ops.v1_obj_chain_bf16(obj2d, m, False, None, plan["stream"], plan["table"], tcat)
    else:
        ops.v1_obj_chain(obj2d, m, False, None, plan["table"], tcat)
    return tcat


def get_center_regression_net(object_point_cloud, one_hot_vec, is_training, bn_decay, end_points, fused=False):
    """T-Net.  Reference: model_util.py:322-357.  fused: run the convolution stack and
    the max-pool as one chain kernel (object point count a multiple of 32)."""
    num_point = object_point_cloud.shape[1]
    if fused and num_point % 32 == 0:
        tf_util._inference_only(is_training)
        net = _tnet_stack_fused(object_point_cloud, one_hot_vec)
        net = tf_util.fully_connected(net, 256, scope='fc1-stage1', bn=True, is_training=is_training,
                                      bn_decay=bn_decay)
        net = tf_util.fully_connected(net, 128, scope='fc2-stage1', bn=True, is_training=is_training,
                                      bn_decay=bn_decay)
        predicted_center = tf_util.fully_connected(net, 3, activation_fn=None, scope='fc3-stage1')
        return predicted_center, end_points
    net = object_point_cloud.unsqueeze(2)
    net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv-reg1-stage1', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv-reg2-stage1', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 256, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv-reg3-stage1', bn_decay=bn_decay)
    net = tf_util.max_pool2d(net, [num_point, 1], padding='VALID', scope='maxpool-stage1')
    net = net.reshape(net.shape[0], math.prod(net.shape[1:]))
    net = torch.cat([net, one_hot_vec], dim=1)
    net = tf_util.fully_connected(net, 256, scope='fc1-stage1', bn=True, is_training=is_training,
                                  bn_decay=bn_decay)
    net = tf_util.fully_connected(net, 128, scope='fc2-stage1', bn=True, is_training=is_training,
                                  bn_decay=bn_decay)
    predicted_center = tf_util.fully_connected(net, 3, activation_fn=None, scope='fc3-stage1')
    return predicted_center, end_points


# Record parsing (models/model_util.py:89-144) lives in fpn.frustum_records; the
# reference's names stay importable from here.
from .frustum_records import (parse_data, parse_frustum_point_record,  # noqa: E402,F401
                              parse_frustum_point_test_record, parse_test_data)
