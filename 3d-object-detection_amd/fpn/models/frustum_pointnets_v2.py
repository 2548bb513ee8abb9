"""Frustum PointNets v2: the PointNet++ (set abstraction / feature propagation)
variant of the detector, composed from fpn.pointnet_util.

The reference ships the modules (models/pointnet_util.py) but no v2 model file; the
topology here is this build's (hyper-parameters in v2_config.json next to this file).
The masking step, the T-Net, the output parsing and the get_model contract are the v1
ones (models/frustum_pointnets_v1.py:141-188, models/model_util.py).
"""
import json
import os

import torch

from .. import tf_util
from ..model_util import (NUM_HEADING_BIN, NUM_SIZE_CLUSTER, get_center_regression_net,
                          parse_output_to_tensors, point_cloud_masking)
from ..pointnet_util import (pointnet_fp_module, pointnet_fp_module_with_head, pointnet_sa_module,
                             pointnet_sa_module_msg)

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "v2_config.json")) as _f:
    CONFIG = json.load(_f)


def get_instance_seg_v2_net(point_cloud, one_hot_vec, is_training, bn_decay, end_points, cfg=None):
    """(B,N,3+C) -> logits (B,N,2): SA-MSG x2, SA group_all, FP x3, two 1-wide convolutions."""
    s = (cfg or CONFIG)["seg"]
    l0_xyz = point_cloud[:, :, 0:3].contiguous()
    l0_points = point_cloud[:, :, 3:].contiguous() if point_cloud.shape[2] > 3 else None
    l1_xyz, l1_points = pointnet_sa_module_msg(l0_xyz, l0_points, s["sa1"]["npoint"], s["sa1"]["radius"],
                                               s["sa1"]["nsample"], s["sa1"]["mlp"], is_training, bn_decay,
                                               scope='layer1')
    l2_xyz, l2_points = pointnet_sa_module_msg(l1_xyz, l1_points, s["sa2"]["npoint"], s["sa2"]["radius"],
                                               s["sa2"]["nsample"], s["sa2"]["mlp"], is_training, bn_decay,
                                               scope='layer2')
    l3_xyz, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                              mlp=s["sa3"]["mlp"], mlp2=None, group_all=True,
                                              is_training=is_training, bn_decay=bn_decay, scope='layer3')
    l3_points = torch.cat([l3_points, one_hot_vec[:, None, :]], dim=2)
    l2_points = pointnet_fp_module(l2_xyz, l3_xyz, l2_points, l3_points, s["fp1"], is_training, bn_decay,
                                   scope='fa_layer1')
    l1_points = pointnet_fp_module(l1_xyz, l2_xyz, l1_points, l2_points, s["fp2"], is_training, bn_decay,
                                   scope='fa_layer2')
    l0_feat = point_cloud if l0_points is not None else l0_xyz      # = [l0_xyz | l0_points]
    # last feature propagation + conv1d(128) + dropout (identity at inference, keep_prob
    # s["dropout_keep"] in training) + conv1d(2): one kernel when the shapes fit
    logits = pointnet_fp_module_with_head(l0_xyz, l1_xyz, l0_feat, l1_points, s["fp3"], 128, 2, is_training,
                                          bn_decay, scope='fa_layer3', head_scope='conv1d-fc1',
                                          out_scope='conv1d-fc2')
    return logits, end_points


def get_3d_box_estimation_v2_net(object_point_cloud, one_hot_vec, is_training, bn_decay, end_points, cfg=None):
    """(B,M,3) -> (B,59): SA x2, SA group_all, three fully connected layers."""
    c = (cfg or CONFIG)["box"]
    l1_xyz, l1_points, _ = pointnet_sa_module(object_point_cloud, None, npoint=c["sa1"]["npoint"],
                                              radius=c["sa1"]["radius"], nsample=c["sa1"]["nsample"],
                                              mlp=c["sa1"]["mlp"], mlp2=None, group_all=False,
                                              is_training=is_training, bn_decay=bn_decay, scope='ssg-layer1')
    l2_xyz, l2_points, _ = pointnet_sa_module(l1_xyz, l1_points, npoint=c["sa2"]["npoint"],
                                              radius=c["sa2"]["radius"], nsample=c["sa2"]["nsample"],
                                              mlp=c["sa2"]["mlp"], mlp2=None, group_all=False,
                                              is_training=is_training, bn_decay=bn_decay, scope='ssg-layer2')
    _, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                         mlp=c["sa3"]["mlp"], mlp2=None, group_all=True,
                                         is_training=is_training, bn_decay=bn_decay, scope='ssg-layer3')
    net = l3_points.reshape(l3_points.shape[0], l3_points.shape[1] * l3_points.shape[2])
    net = torch.cat([net, one_hot_vec], dim=1)
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
    net = tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='fc2', bn_decay=bn_decay)
    output = tf_util.fully_connected(net, 3 + NUM_HEADING_BIN * 2 + NUM_SIZE_CLUSTER * 4, activation_fn=None,
                                     scope='fc3')
    return output, end_points


def get_model(point_cloud, one_hot_vec, is_training, bn_decay=None, cfg=None):
    """Same contract as the v1 get_model (models/frustum_pointnets_v1.py:141-188)."""
    tf_util._inference_only(is_training)
    point_cloud = point_cloud.contiguous()
    one_hot_vec = one_hot_vec.contiguous()
    end_points = {}
    logits, end_points = get_instance_seg_v2_net(point_cloud, one_hot_vec, is_training, bn_decay, end_points, cfg)
    end_points['mask_logits'] = logits
    object_point_cloud_xyz, mask_xyz_mean, end_points = point_cloud_masking(point_cloud, logits, end_points)
    end_points['mask_xyz_mean'] = mask_xyz_mean
    from .. import pointnet_util
    center_delta, end_points = get_center_regression_net(object_point_cloud_xyz, one_hot_vec, is_training,
                                                         bn_decay, end_points, fused=pointnet_util.FUSE_SA)
    end_points['center_delta'] = center_delta
    stage1_center = center_delta + mask_xyz_mean
    end_points['stage1_center'] = stage1_center
    object_point_cloud_xyz_new = (object_point_cloud_xyz - center_delta.unsqueeze(1)).contiguous()
    output, end_points = get_3d_box_estimation_v2_net(object_point_cloud_xyz_new, one_hot_vec, is_training,
                                                      bn_decay, end_points, cfg)
    end_points['box_output'] = output
    end_points = parse_output_to_tensors(output, end_points)
    end_points['center'] = end_points['center_boxnet'] + stage1_center
    return end_points
