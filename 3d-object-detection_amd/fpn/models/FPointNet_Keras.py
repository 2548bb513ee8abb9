"""Deterministic Keras-style detector: counterpart of models/FPointNet_Keras.py.

The reference's Keras re-statement is the v1 graph on a 3-channel point cloud with
batch norm commented out (FPointNet_Keras.py:25-37,108): every layer is convolution /
dense + ReLU.  It runs here through the same planned kernels as v1 with bn=False.
Variable scopes are the v1 ones (conv1..conv10, conv-reg*-stage1, fc*-stage1,
conv-reg1..4, fc1..3) in the order the Keras layers are created.
"""
from .. import plan as _plan
from .. import tf_util, variables
from ._keras_like import DetectionModel, TrainingModelStub

_cache = {}


def _plan_for(store, channels):
    key = (id(store), channels)
    p = _cache.get(key)
    if p is None or p.store is not store:
        p = _plan.V1Plan(store, channels, bn=False)
        _cache[key] = p
    return p


def get_model(point_cloud, one_hot_vec, is_training, bn_decay=None):
    """end_points dict, same keys as models/frustum_pointnets_v1.py:141-188."""
    tf_util._inference_only(is_training)
    store = variables.get_default_store()
    return _plan_for(store, point_cloud.shape[2]).forward(point_cloud.contiguous(), one_hot_vec.contiguous())


def Frustum_Pointnet_Model(point_cloud_shape=(1024, 3), one_hot_vec_shape=(3,), mask_label_shape=(1024,),
                           center_label_shape=(3,), heading_class_label_shape=(),
                           heading_residual_label_shape=(), size_class_label_shape=(),
                           size_residual_label_shape=(3,), batch_size=32):
    """Reference: FPointNet_Keras.py:108-191.  Returns (training_model, det_model)."""
    store = variables.get_default_store()
    p = _plan_for(store, point_cloud_shape[1])
    return TrainingModelStub(), DetectionModel(p.forward, store, store.device)
