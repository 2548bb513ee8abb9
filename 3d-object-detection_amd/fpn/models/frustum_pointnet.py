"""Counterpart of models/frustum_pointnet.py: the reference's older functional Keras
build (Flipout convolutions + batch norm, nine inputs, `frustum_pointnet(...)` ->
(train_model, detection_model) with eleven detection outputs, :108-295).  That file
cannot be imported even where TensorFlow exists (it imports a module `fp_loss` that
is not in the repository, :12), so only its contract is reproduced: the constructor
name, the (train, detection) pair and the order of the eleven outputs (:293-295) - and the
network itself, read from the file's text:
  * Flipout in every layer (conv_bn_act / Dense_bn_act with the default is_bayes=True,
    :15-41), the two output layers included (:203,225);
  * Keras BatchNormalization after every layer but the last (default bn=True: conv10 :150
    and fc3-stage1 :203 have it; the 59-wide output layer :225 is built with bn=False);
  * the box stack reads the object points as gathered, NOT shifted by the T-net's centre
    (:210-214 forms object_point_cloud_xyz_new and then expands object_point_cloud); the
    final centre is still boxnet_center + stage1_center (:253).
It runs on the Flipout plan of fpn.bayes with bn=True, flipout_outputs=True,
recentre_box_input=False.  Variables: the plan's scopes '<scope>/weights', '/rho',
'/biases', '<scope>/bn/{gamma,beta,moving_mean,moving_variance}' = the Keras layers
'<scope>' (kernel_posterior loc / untransformed scale, bias_posterior loc) and
'<scope>_BN'; the last layer, which the file leaves at Dense_bn_act's default name 'fc'
(:225), is 'fc3' here as in the other model files.
"""
from .. import bayes as _bayes
from .. import tf_util, variables
from ._keras_like import DetectionModel, TrainingModelStub

DETECTION_OUTPUTS = ("mask", "mask_logits", "center", "stage1_center", "center_boxnet", "heading_scores",
                     "heading_residuals_normalized", "heading_residuals", "size_scores",
                     "size_residuals_normalized", "size_residuals")   # frustum_pointnet.py:293-295
PLAN_OPTIONS = dict(bn=True, flipout_outputs=True, recentre_box_input=False)


class _Detection11(DetectionModel):
    OUTPUTS = DETECTION_OUTPUTS


def get_model(point_cloud, one_hot_vec, is_training, bn_decay=None):
    tf_util._inference_only(is_training)
    store = variables.get_default_store()
    p = _bayes.BayesPlan(store, point_cloud.shape[2], **PLAN_OPTIONS)
    return {k: v[0] for k, v in p.forward(point_cloud.contiguous(), one_hot_vec.contiguous(), 1).items()}


def frustum_pointnet(point_cloud_shape=(1024, 4), one_hot_vec_shape=(3,), g_mean_size_tensor_shape=(3, 3),
                     mask_label_shape=(1024,), center_label_shape=(3,), heading_class_label_shape=(),
                     heading_residual_label_shape=(), size_class_label_shape=(), size_residual_label_shape=(3,)):
    """-> (train_model, detection_model); the reference's signature (:108-110).  Only
    point_cloud_shape matters for inference: (points, channels), xyz first.  The detection
    model's `predict` takes the reference's nine-input list too (the first two are read, the
    seven label / mean-size inputs only feed the loss, :272-288) and any number of points
    (the reference pools over a fixed 1024, :125,135)."""
    store = variables.get_default_store()
    p = _bayes.BayesPlan(store, int(point_cloud_shape[1]), **PLAN_OPTIONS)

    def forward(pc, oh):
        return {k: v[0] for k, v in p.forward(pc, oh, 1).items()}
    return TrainingModelStub(), _Detection11(forward, store, store.device)
