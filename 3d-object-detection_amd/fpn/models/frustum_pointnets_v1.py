"""Frustum PointNets v1: counterpart of models/frustum_pointnets_v1.py.

`get_model(point_cloud, one_hot_vec, is_training, bn_decay=None)` keeps the
reference's contract (:141-188): (B,N,4) float32 frustum points and (B,3) one-hot
class vectors in, the `end_points` dict out.  Two executions of the same graph
are offered:

  * planned (default): fpn.plan.V1Plan - fused layers, hoisted conv6, chunked;
  * literal=True: the reference's layer-by-layer text on top of fpn.tf_util
    (materialises the tiled 1091-channel tensor exactly as :64-66 does); kept as a
    cross-check of the plan and for callers that build on the sub-network
    functions.

Variables live in the active fpn.variables.VariableStore under the reference's
scope names.  Inference only.
"""
import torch

from .. import plan as _plan
from .. import tf_util, variables
from ..model_util import (NUM_HEADING_BIN, NUM_OBJECT_POINT, NUM_SIZE_CLUSTER,  # noqa: F401
                          get_center_regression_net, parse_output_to_tensors, point_cloud_masking)


def get_instance_seg_v1_net(point_cloud, one_hot_vec, is_training, bn_decay, end_points):
    """3D instance segmentation PointNet.  Reference: frustum_pointnets_v1.py:20-90."""
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    net = point_cloud.unsqueeze(2)
    net = tf_util.conv2d(net, 64, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv1', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 64, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv2', bn_decay=bn_decay)
    point_feat = tf_util.conv2d(net, 64, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                                is_training=is_training, scope='conv3', bn_decay=bn_decay)
    net = tf_util.conv2d(point_feat, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv4', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 1024, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv5', bn_decay=bn_decay)
    global_feat = tf_util.max_pool2d(net, [num_point, 1], padding='VALID', scope='maxpool')
    global_feat = torch.cat([global_feat, one_hot_vec[:, None, None, :]], dim=3)
    global_feat_expand = global_feat.expand(batch_size, num_point, 1, global_feat.shape[3])
    concat_feat = torch.cat([point_feat, global_feat_expand], dim=3)
    net = tf_util.conv2d(concat_feat, 512, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv6', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 256, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv7', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv8', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv9', bn_decay=bn_decay)
    net = tf_util.dropout(net, is_training, 'dp1', keep_prob=0.5)
    logits = tf_util.conv2d(net, 2, [1, 1], padding='VALID', stride=[1, 1], activation_fn=None,
                            scope='conv10')
    logits = logits.squeeze(2)
    return logits, end_points


def get_3d_box_estimation_v1_net(object_point_cloud, one_hot_vec, is_training, bn_decay, end_points):
    """3D box estimation PointNet.  Reference: frustum_pointnets_v1.py:93-138."""
    num_point = object_point_cloud.shape[1]
    net = object_point_cloud.unsqueeze(2)
    net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv-reg1', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv-reg2', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 256, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv-reg3', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 512, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv-reg4', bn_decay=bn_decay)
    net = tf_util.max_pool2d(net, [num_point, 1], padding='VALID', scope='maxpool2')
    net = net.reshape(net.shape[0], -1)
    net = torch.cat([net, one_hot_vec], dim=1)
    net = tf_util.fully_connected(net, 512, scope='fc1', bn=True, is_training=is_training,
                                  bn_decay=bn_decay)
    net = tf_util.fully_connected(net, 256, scope='fc2', bn=True, is_training=is_training,
                                  bn_decay=bn_decay)
    output = tf_util.fully_connected(net, 3 + NUM_HEADING_BIN * 2 + NUM_SIZE_CLUSTER * 4,
                                     activation_fn=None, scope='fc3')
    return output, end_points


def _literal(point_cloud, one_hot_vec, is_training, bn_decay):
    end_points = {}
    logits, end_points = get_instance_seg_v1_net(point_cloud, one_hot_vec, is_training, bn_decay, end_points)
    end_points['mask_logits'] = logits
    object_point_cloud_xyz, mask_xyz_mean, end_points = point_cloud_masking(point_cloud, logits, end_points)
    end_points['mask_xyz_mean'] = mask_xyz_mean
    center_delta, end_points = get_center_regression_net(object_point_cloud_xyz, one_hot_vec,
                                                         is_training, bn_decay, end_points)
    end_points['center_delta'] = center_delta
    stage1_center = center_delta + mask_xyz_mean
    end_points['stage1_center'] = stage1_center
    object_point_cloud_xyz_new = object_point_cloud_xyz - center_delta.unsqueeze(1)
    output, end_points = get_3d_box_estimation_v1_net(object_point_cloud_xyz_new, one_hot_vec,
                                                      is_training, bn_decay, end_points)
    end_points['box_output'] = output
    end_points = parse_output_to_tensors(output, end_points)
    end_points['center'] = end_points['center_boxnet'] + stage1_center
    return end_points


_plans = {}


def plan_for(store, pc_channels, bn=True, chunk_frustums=None):
    key = (id(store), pc_channels, bn, chunk_frustums)
    p = _plans.get(key)
    if p is None or p.store is not store:
        p = _plan.V1Plan(store, pc_channels, bn, chunk_frustums)
        _plans[key] = p
    return p


def get_model(point_cloud, one_hot_vec, is_training, bn_decay=None, literal=False):
    """Reference: frustum_pointnets_v1.py:141-188.  Returns the end_points dict:
    mask_logits (B,N,2), mask (B,N), stage1_center (B,3), center_boxnet (B,3),
    heading_scores (B,12), heading_residuals_normalized (B,12), heading_residuals
    (B,12), size_scores (B,8), size_residuals_normalized (B,8,3), size_residuals
    (B,8,3), center (B,3); plus object_point_indices (B,512), mask_xyz_mean,
    center_delta and box_output (B,59) for inspection."""
    tf_util._inference_only(is_training)
    if literal:
        return _literal(point_cloud.contiguous(), one_hot_vec.contiguous(), is_training, bn_decay)
    store = variables.get_default_store()
    return plan_for(store, point_cloud.shape[2]).forward(point_cloud.contiguous(), one_hot_vec.contiguous())
