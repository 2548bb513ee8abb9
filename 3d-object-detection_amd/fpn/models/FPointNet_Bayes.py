"""Bayesian (Flipout) detector: counterpart of models/FPointNet_Bayes.py.

`get_model(point_cloud, one_hot_vec, is_training, bn_decay=None, nsamples=1)` returns
the v1 end_points dict of ONE stochastic forward pass (nsamples=1, what a `predict`
call of the reference's det_model gives) or, with nsamples=S, the same tensors with a
leading sample axis plus end_points['mc_statistics'] (B,30), the ensemble statistics of
fpn.uncertainty.  `Frustum_Pointnet_Model(...)` keeps the reference constructor
(:140-223): (training_model, det_model), each det_model call drawing fresh noise.
"""
from .. import bayes as _bayes
from .. import tf_util, uncertainty, variables
from ._keras_like import DetectionModel, TrainingModelStub

_cache = {}


def _plan_for(store, channels, mc_dropout=False):
    key = (id(store), channels, mc_dropout)
    p = _cache.get(key)
    if p is None or p.store is not store:
        p = _bayes.BayesPlan(store, channels, mc_dropout=mc_dropout)
        _cache[key] = p
    return p


def get_model(point_cloud, one_hot_vec, is_training, bn_decay=None, nsamples=1, generator=None, noise=None,
              mc_dropout=False):
    tf_util._inference_only(is_training)
    store = variables.get_default_store()
    p = _plan_for(store, point_cloud.shape[2], mc_dropout)
    ep = p.forward(point_cloud.contiguous(), one_hot_vec.contiguous(), nsamples, noise, generator)
    if nsamples == 1 and noise is None:
        return {k: v[0] for k, v in ep.items()}
    ep["mc_statistics"] = uncertainty.mc_statistics(ep["box_output"].contiguous(), ep["stage1_center"].contiguous())
    return ep


def Frustum_Pointnet_Model(point_cloud_shape=(1024, 3), one_hot_vec_shape=(3,), mask_label_shape=(1024,),
                           center_label_shape=(3,), heading_class_label_shape=(),
                           heading_residual_label_shape=(), size_class_label_shape=(),
                           size_residual_label_shape=(3,), batch_size=32):
    """Reference: FPointNet_Bayes.py:140-223."""
    store = variables.get_default_store()
    p = _plan_for(store, point_cloud_shape[1])

    def forward(pc, oh):
        return {k: v[0] for k, v in p.forward(pc, oh, 1).items()}
    return TrainingModelStub(), DetectionModel(forward, store, store.device)
