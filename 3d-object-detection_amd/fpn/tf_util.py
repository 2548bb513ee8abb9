"""Layer wrappers: counterpart of the part of models/tf_util.py the hot path uses
(conv1d, conv2d, fully_connected, max_pool2d, dropout), inference mode.

Signatures follow the reference; tensors are NHWC torch tensors on a ROCm device.
A layer is ONE call into libfpn_hip.so: the 1x1 convolution (or matmul), bias,
inference batch norm and ReLU are fused (the reference issues them as separate TF
ops, tf_util.py:170-184).  Variables come from the active VariableStore under the
same scoped names TF would give them.
"""
import torch

from . import ops, variables


def relu(x):
    """Default activation_fn (the reference passes tf.nn.relu)."""
    return torch.relu(x)


def _inference_only(is_training):
    if isinstance(is_training, torch.Tensor):
        is_training = bool(is_training.item())
    if is_training:
        raise NotImplementedError(
            "training mode (batch statistics, dropout masks, gradients) is outside this inference path")


def _dense(x2d, cin, cout, scope, bn, activation_fn):
    store = variables.get_default_store()
    name = variables.scoped(scope)
    w = store.get_variable(name + "/weights", (cin, cout), "xavier")
    store.get_variable(name + "/biases", (cout,), "zeros")
    if bn:
        store.get_variable(name + "/bn/gamma", (cout,), "ones")
        store.get_variable(name + "/bn/beta", (cout,), "zeros")
        store.get_variable(name + "/bn/moving_mean", (cout,), "zeros")
        store.get_variable(name + "/bn/moving_variance", (cout,), "ones")
    scale, shift = store.folded(name, bn)
    fused_relu = activation_fn is relu
    y = ops.shared_mlp(x2d, w, scale, shift, relu=fused_relu)
    if activation_fn is not None and not fused_relu:
        y = activation_fn(y)
    return y


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME',
           data_format='NHWC', use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=relu,
           bn=False, bn_decay=None, is_training=None):
    """Reference: tf_util.conv2d, models/tf_util.py:120-185.  Only what the point
    networks use is implemented: 1x1 kernels, unit stride, NHWC."""
    _inference_only(is_training)
    if list(kernel_size) != [1, 1] or list(stride) != [1, 1] or data_format != 'NHWC':
        raise NotImplementedError("conv2d: only 1x1 kernels with unit stride in NHWC are on the hot path")
    if inputs.dim() != 4:
        raise ValueError("conv2d expects a 4-D BxHxWxC tensor")
    b, h, w_, c = inputs.shape
    x2d = inputs.contiguous().reshape(b * h * w_, c)
    y = _dense(x2d, c, num_output_channels, scope, bn, activation_fn)
    return y.reshape(b, h, w_, num_output_channels)


def conv1d(inputs, num_output_channels, kernel_size, scope, stride=1, padding='SAME',
           data_format='NHWC', use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=relu,
           bn=False, bn_decay=None, is_training=None):
    """Reference: tf_util.conv1d, models/tf_util.py:52-115; kernel size 1 only."""
    _inference_only(is_training)
    if kernel_size != 1 or stride != 1 or data_format != 'NHWC':
        raise NotImplementedError("conv1d: only kernel_size=1, stride=1, NHWC is on the hot path")
    if inputs.dim() != 3:
        raise ValueError("conv1d expects a 3-D BxLxC tensor")
    b, l, c = inputs.shape
    y = _dense(inputs.contiguous().reshape(b * l, c), c, num_output_channels, scope, bn, activation_fn)
    return y.reshape(b, l, num_output_channels)


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=None,
                    activation_fn=relu, bn=False, bn_decay=None, is_training=None):
    """Reference: tf_util.fully_connected, models/tf_util.py:327-363."""
    _inference_only(is_training)
    if inputs.dim() != 2:
        raise ValueError("fully_connected expects a 2-D BxN tensor")
    return _dense(inputs.contiguous(), inputs.shape[1], num_outputs, scope, bn, activation_fn)


def max_pool2d(inputs, kernel_size, scope, stride=[2, 2], padding='VALID'):
    """Reference: tf_util.max_pool2d, models/tf_util.py:366-389.  The point
    networks only ever pool over the whole point axis (kernel [N,1], VALID), which
    is what is implemented; the models' planned path fuses this pooling into the
    preceding layer instead of calling it."""
    b, h, w_, c = inputs.shape
    if list(kernel_size) != [h, 1] or padding != 'VALID' or w_ != 1:
        raise NotImplementedError("max_pool2d: only the global pool over points ([N,1], VALID) is on the hot path")
    x2d = inputs.contiguous().reshape(b * h, c)
    return ops.max_pool_rows(x2d, h).reshape(b, 1, 1, c)


def dropout(inputs, is_training, scope, keep_prob=0.5, noise_shape=None):
    """Reference: tf_util.dropout, models/tf_util.py:594-615: identity at inference."""
    _inference_only(is_training)
    return inputs
