"""Counterpart of models/frustum_pointnet.py: the reference's older functional Keras
build (Flipout convolutions + batch norm, nine inputs, `frustum_pointnet(...)` ->
(train_model, detection_model) with eleven detection outputs, :108-295).  That file
cannot be imported even where TensorFlow exists (it imports a module `fp_loss` that
is not in the repository, :12), so only its contract is reproduced: the constructor
name, the (train, detection) pair and the order of the eleven outputs.  The network
behind it is the Flipout plan of fpn.bayes (same topology; the reference's batch norm
layers there have no trained statistics to load and are treated as identity).
"""
from .. import bayes as _bayes
from .. import tf_util, variables
from ._keras_like import DetectionModel, TrainingModelStub

DETECTION_OUTPUTS = ("mask_logits", "mask", "stage1_center", "center_boxnet", "heading_scores",
                     "heading_residuals_normalized", "heading_residuals", "size_scores",
                     "size_residuals_normalized", "size_residuals", "center")   # frustum_pointnet.py:290-295


class _Detection11(DetectionModel):
    OUTPUTS = DETECTION_OUTPUTS


def get_model(point_cloud, one_hot_vec, is_training, bn_decay=None):
    tf_util._inference_only(is_training)
    store = variables.get_default_store()
    p = _bayes.BayesPlan(store, point_cloud.shape[2])
    return {k: v[0] for k, v in p.forward(point_cloud.contiguous(), one_hot_vec.contiguous(), 1).items()}


def frustum_pointnet(num_point=1024, num_channels=3, batch_size=32, **unused):
    """-> (train_model, detection_model)."""
    store = variables.get_default_store()
    p = _bayes.BayesPlan(store, num_channels)

    def forward(pc, oh):
        return {k: v[0] for k, v in p.forward(pc, oh, 1).items()}
    return TrainingModelStub(), _Detection11(forward, store, store.device)
