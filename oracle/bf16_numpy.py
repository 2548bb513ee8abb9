"""float64 restatement of the ROUNDING MODEL of the bf16 layer-chain kernels (the v1
topology, deterministic and Flipout).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product never imports this module.

The bf16 path is the network of models/frustum_pointnets_v1.py:20-188 (or of
models/FPointNet_Bayes.py:71-223) with roundings to bfloat16 (8 significant bits, round
to nearest even) inserted at stated places and nothing else changed:

  * weights of every convolution but the first of a stack and conv10: the batch-norm scale
    is multiplied into the columns in float32, then the product is rounded once;
  * activations handed from one convolution to the next (after shift and ReLU), including
    point_feat; NOT rounded: the maxima over points (global features), conv9's output, the
    shifts, conv10 and every fully connected layer (all float32 in the kernels);
  * the first convolution of every stack reads coordinates in metres and multiplies SPLIT
    operands: x = xh + xl, w = wh + wl, shift = sh + sl with every part a bfloat16 number,
    product = xh wh + xl wh + xh wl (what is dropped is below 2^-16 of |x| |w|).

Everything between the roundings is evaluated here in float64 (the kernels accumulate in
float32: 1e-6 of difference).  A kernel that matches this model to a few 1e-4 of the
output's scale computes what it claims to compute, for ANY weights and inputs - that is the
parity check of the kernels (tests/test_gpu_bf16.py).  How far the rounding model itself is
from the unrounded network is a property of the weights, not of the kernels, and is stated
and sampled separately.

PARITY UNPINNED by the reference, like oracle/v1_numpy.py: the reference has no reduced-
precision path and no forward values."""
import numpy as np

from . import v1_numpy as V

f32 = np.float32
PLAIN = ("fc3-stage1", "fc3")      # the two output layers: no batch norm, no ReLU, never Flipout


def rb(x):
    """Round float32(x) to the nearest bfloat16 (ties to even); returned as float64."""
    u = np.ascontiguousarray(np.asarray(x, f32)).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(f32).astype(np.float64)


def split(x):
    """x -> (hi, lo), both bfloat16 numbers, hi = rb(x), lo = rb(float32(x - hi))."""
    x = np.asarray(x, f32)
    hi = rb(x)
    return hi, rb((x - hi.astype(f32)).astype(f32))


def fold(params, scope, bn):
    """(scale, shift) in float32 as fpn.variables.VariableStore.folded forms them
    (float64 arithmetic rounded once); scale None without batch norm."""
    bias = params[scope + "/biases"].astype(np.float64)
    if not bn:
        return None, params[scope + "/biases"].astype(f32)
    inv = params[scope + "/bn/gamma"].astype(np.float64) / np.sqrt(
        params[scope + "/bn/moving_variance"].astype(np.float64) + V.BN_EPS)
    shift = params[scope + "/bn/beta"].astype(np.float64) + (bias - params[scope + "/bn/moving_mean"].astype(np.float64)) * inv
    return inv.astype(f32), shift.astype(f32)


def scaled(w, scale):
    """float32 product of the weights with the column scale (w (...,cin,cout))."""
    w = np.asarray(w, f32)
    return w if scale is None else (w * np.asarray(scale, f32)).astype(f32)


def first_layer(x, w, scale, shift):
    """x (B,R,K) float32 values; w (K,C) or (B,K,C); -> bf16-valued (B,R,C)."""
    wh, wl = split(scaled(w, scale))
    xh, xl = split(x)
    sh, sl = split(shift)
    y = xh @ wh + xl @ wh + xh @ wl + (sh + sl)
    return rb(np.maximum(y, 0.0))


def layer(a, w, scale, shift, keep=False):
    """a bf16-valued (B,R,K); shift (C,) or (B,1,C); -> rb(relu(a . rb(w scale) + shift)),
    or the unrounded float value with keep=True (conv9)."""
    y = np.maximum(a @ rb(scaled(w, scale)) + np.asarray(shift, np.float64), 0.0)
    return y if keep else rb(y)


def pooled_layer(a, w, scale, shift):
    """Last layer of a stack: relu(max over the rows of (a . rb(w scale)) + shift), float."""
    return np.maximum((a @ rb(scaled(w, scale))).max(axis=1) + np.asarray(shift, np.float64), 0.0)


class Weights:
    """What the layers of one forward pass multiply with.  conv(scope) -> (w, scale, shift)
    with w (cin,cout) shared or (B,cin,cout) per frustum; dense(x, scope, relu) evaluates a
    fully connected layer (float arithmetic, no rounding)."""

    def __init__(self, params, bn):
        self.params, self.bn = params, bn

    def conv(self, scope):
        scale, shift = fold(self.params, scope, self.bn and scope != "conv10")
        return self.params[scope + "/weights"].astype(f32), scale, shift

    def dense(self, x, scope, relu):
        return V.dense(x, self.params, scope, self.bn and scope not in PLAIN, relu, np.float64)

    def summed_magnitude(self, x, scope):
        """sum_k |x_k| |w_kj| + |b_j| of a deterministic output layer."""
        return np.abs(x) @ np.abs(self.params[scope + "/weights"].astype(np.float64)) + \
            np.abs(self.params[scope + "/biases"].astype(np.float64))

    def conv6_frustum_part(self, g):
        """[global_feat | one_hot] (B,1027) -> (B,512): the frustum-constant inputs of conv6
        through their rows of W6, scale and shift of conv6 applied, no ReLU."""
        w, scale, shift = self.conv("conv6")
        y = g @ w[64:].astype(np.float64)
        return (y if scale is None else y * scale.astype(np.float64)) + shift.astype(np.float64)


FIRST = ("conv1", "conv-reg1-stage1", "conv-reg1")      # first layers of the stacks: float32 weights, split operands
NO_BN = ("fc3",)                                        # models/frustum_pointnet.py:126


class FlipoutWeights(Weights):
    """One Monte-Carlo sample: noise_s[scope] = (dW (cin,cout), s_in (B,cin), s_out (B,cout));
    the convolutions see W_eff[b] = W + diag(s_in[b]) dW diag(s_out[b]) in float32 (signs are
    +-1: the sum is the one rounding).  bn: batch norm after every layer but the last
    (models/frustum_pointnet.py); the product then scales mean and perturbation by the
    batch-norm scale in float32 BEFORE it forms the effective weights of a bf16 stream (the
    column scale commutes with the output signs), and multiplies conv10's float32 effective
    weights by it; a stack's first layer keeps unscaled effective weights and the kernel
    multiplies (then splits) them."""

    def __init__(self, params, noise_s, bn=False):
        super().__init__(params, bn)
        self.noise = noise_s

    def _fold(self, scope):
        return fold(self.params, scope, self.bn and scope not in NO_BN)

    def conv(self, scope):
        dw, si, so = self.noise[scope]
        scale, shift = self._fold(scope)
        w = self.params[scope + "/weights"].astype(f32)
        dw = np.asarray(dw, f32)
        pre = scale is not None and scope not in FIRST and scope != "conv10"
        if pre:
            w, dw = (w * scale).astype(f32), (dw * scale).astype(f32)
        eff = (w.astype(np.float64)[None] + si[:, :, None] * dw.astype(np.float64)[None] * so[:, None, :]).astype(f32)
        if scale is not None and scope == "conv10":
            eff = (eff * scale).astype(f32)
        return eff, (scale if scope in FIRST else None), shift

    def dense(self, x, scope, relu):
        scale, shift = self._fold(scope)
        w = self.params[scope + "/weights"].astype(np.float64)
        y = x @ w
        if scope in self.noise:
            dw, si, so = self.noise[scope]
            y = y + ((x * si) @ np.asarray(dw, np.float64)) * so
        y = (y if scale is None else y * scale.astype(np.float64)) + shift.astype(np.float64)
        return np.maximum(y, 0.0) if relu else y

    def summed_magnitude(self, x, scope):
        scale, shift = self._fold(scope)
        w = np.abs(self.params[scope + "/weights"].astype(np.float64))
        if scale is not None:
            w = w * np.abs(scale.astype(np.float64))
        return np.abs(x) @ w + np.abs(shift.astype(np.float64))

    def conv6_frustum_part(self, g):
        dw, si, so = self.noise["conv6"]
        scale, shift = self._fold("conv6")
        w = self.params["conv6/weights"].astype(np.float64)[64:]
        y = g @ w + ((g * si[:, 64:]) @ np.asarray(dw, np.float64)[64:]) * so
        return (y if scale is None else y * scale.astype(np.float64)) + shift.astype(np.float64)


def _rows(w, lo, hi):
    return w[..., lo:hi, :]


def forward(pc, one_hot, weights, mask_override=None, dropout_keep=None, recentre_box_input=True):
    """The v1 topology under the rounding model.  pc (B,N,C) float32, one_hot (B,3);
    weights: Weights or FlipoutWeights; dropout_keep (B,N,128) bool or None (kept activations
    of conv9 doubled, the others zero: layers.Dropout(0.5) active, FPointNet_Bayes.py:92).
    -> the end points of oracle.v1_numpy.get_model (float64), and under "summed_magnitude"
    for mask_logits, center_delta, stage1_center, box_output and center the magnitude of what
    the output's last layer sums, sum_k |a_k| |w_kj| + |b_j| (the scale against which the
    rounding error of a sum is judged: with random weights the sum itself can be a hundredth
    of it)."""
    B, N, _ = pc.shape
    M = V.NUM_OBJECT_POINT
    one_hot = np.asarray(one_hot, np.float64)
    W = weights

    # segmentation net: conv1..conv5 + max (frustum_pointnets_v1.py:41-61)
    a = first_layer(np.asarray(pc, f32), *W.conv("conv1"))
    a = layer(a, *W.conv("conv2"))
    pf = layer(a, *W.conv("conv3"))
    a = layer(pf, *W.conv("conv4"))
    g = pooled_layer(a, *W.conv("conv5"))
    g = np.concatenate([g, one_hot], 1)                                   # :64
    # conv6: point part on bf16 operands, the frustum part (float) is its shift (:64-71)
    w6, scale6, _ = W.conv("conv6")
    gshift = W.conv6_frustum_part(g)[:, None, :]
    a = layer(pf, _rows(w6, 0, 64), scale6, gshift)
    a = layer(a, *W.conv("conv7"))
    a = layer(a, *W.conv("conv8"))
    a = layer(a, *W.conv("conv9"), keep=True)
    if dropout_keep is not None:
        a = a * np.asarray(dropout_keep, np.float64).reshape(B, N, 128) * 2.0
    w10, _, b10 = W.conv("conv10")
    logits = a @ w10.astype(np.float64) + b10.astype(np.float64)         # float32 in the kernel
    scale = {"mask_logits": np.abs(a) @ np.abs(w10.astype(np.float64)) + np.abs(b10.astype(np.float64))}

    obj, mean, mask, idx = V.point_cloud_masking(pc, logits, dtype=np.float32, mask_override=mask_override)
    ep = {"mask_logits": logits, "mask": mask, "object_point_indices": idx, "mask_xyz_mean": mean.astype(np.float64)}

    def stack(x, scopes):
        t = first_layer(x, *W.conv(scopes[0]))
        for s in scopes[1:-1]:
            t = layer(t, *W.conv(s))
        return np.concatenate([pooled_layer(t, *W.conv(scopes[-1])), one_hot], 1)

    # T-Net (model_util.py:322-357)
    t = stack(obj.astype(f32), [s for s, _ in V.TNET_CONV])
    for s, _ in V.TNET_FC:
        t = W.dense(t, s, True)
    delta = W.dense(t, "fc3-stage1", False)
    ep["center_delta"] = delta
    ep["stage1_center"] = delta + ep["mask_xyz_mean"]
    scale["center_delta"] = W.summed_magnitude(t, "fc3-stage1")
    scale["stage1_center"] = scale["center_delta"] + np.abs(ep["mask_xyz_mean"])
    # box net (frustum_pointnets_v1.py:93-138); the kernel subtracts the float32 centre while loading
    # (recentre_box_input=False: models/frustum_pointnet.py:214 feeds the un-shifted points)
    obj_new = (obj.astype(f32) - delta.astype(f32)[:, None, :]).astype(f32) if recentre_box_input else obj.astype(f32)
    t = stack(obj_new, [s for s, _ in V.BOX_CONV])
    for s, _ in V.BOX_FC:
        t = W.dense(t, s, True)
    out = W.dense(t, "fc3", False)
    ep["box_output"] = out
    V.parse_output(out, ep, np.float64)
    ep["center"] = ep["center_boxnet"] + ep["stage1_center"]
    scale["box_output"] = W.summed_magnitude(t, "fc3")
    scale["center"] = scale["box_output"][:, :3] + scale["stage1_center"]
    ep["summed_magnitude"] = scale
    return ep
