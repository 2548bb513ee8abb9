"""numpy restatement of the Flipout network of models/FPointNet_Bayes.py and of the
Monte-Carlo ensemble statistics of the project report.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the Flipout
arithmetic lives in tensorflow_probability (tfp.layers.Convolution2DFlipout /
DenseFlipout; version not pinned by the reference, not present in its tree, not
installed here), and the reference holds no test or value for these layers.  What
is restated is the published estimator as tfp implements it,
    y = x.W + b + ((x*s_in).dW)*s_out,
evaluated in its textbook two-product form, with the noise (dW, s_in, s_out) given
explicitly so that the HIP path can be compared on identical draws.
"""
import numpy as np

from . import v1_numpy as V

PLAIN = ("fc3-stage1", "fc3")


NO_BN = ("fc3",)       # models/frustum_pointnet.py:225: the last layer alone is built with bn=False


def batch_norm(y, params, scope, dtype=np.float64):
    """Keras BatchNormalization in inference mode (models/frustum_pointnet.py:23-24,36-37; epsilon
    1e-3, the Keras default): (y - mean) * gamma / sqrt(var + eps) + beta."""
    inv = params[scope + "/bn/gamma"].astype(dtype) / np.sqrt(params[scope + "/bn/moving_variance"].astype(dtype) + V.BN_EPS)
    return y * inv + (params[scope + "/bn/beta"].astype(dtype) - params[scope + "/bn/moving_mean"].astype(dtype) * inv)


def flipout_dense(x, params, scope, dw, s_in, s_out, rows_per_group, relu, dtype=np.float64, bn=False):
    """x (rows,cin); dw (cin,cout); s_in (groups,cin); s_out (groups,cout)."""
    w = params[scope + "/weights"].astype(dtype)
    b = params[scope + "/biases"].astype(dtype)
    si = np.repeat(s_in.astype(dtype), rows_per_group, axis=0)
    so = np.repeat(s_out.astype(dtype), rows_per_group, axis=0)
    x = x.astype(dtype)
    y = x @ w + b + ((x * si) @ dw.astype(dtype)) * so
    if bn:
        y = batch_norm(y, params, scope, dtype)
    return np.maximum(y, 0) if relu else y


def forward_sample(pc, one_hot, params, noise_s, dtype=np.float64, mask_override=None, dropout_keep=None, bn=False,
                   recentre_box_input=True):
    """One Monte-Carlo pass (FPointNet_Bayes.py:167-185).  noise_s[scope] =
    (dW (cin,cout), s_in (B,cin), s_out (B,cout)) for every Flipout layer.
    dropout_keep (B,N,128) bool, if given: layers.Dropout(rate=0.5) before the 2-channel
    head (:92) ACTIVE, as in Monte-Carlo dropout: kept activations are scaled by
    1 / (1 - rate) = 2, the others are zero.
    bn=True and noise for 'fc3-stage1' / 'fc3': the network of models/frustum_pointnet.py:108-295
    instead - batch norm after every layer but the last (conv10 and fc3-stage1 included: their
    conv_bn_act / Dense_bn_act calls keep the default bn=True, :51,104) and Flipout output layers."""
    B, N, _ = pc.shape
    M = V.NUM_OBJECT_POINT

    def conv(x, scope, rpg, relu=True):
        dw, si, so = noise_s[scope]
        return flipout_dense(x, params, scope, dw, si, so, rpg, relu, dtype, bn=bn and scope not in NO_BN)

    def output_layer(x, scope):
        if scope in noise_s:
            return conv(x, scope, 1, relu=False)
        y = V.dense(x, params, scope, False, False, dtype)
        return batch_norm(y, params, scope, dtype) if bn and scope not in NO_BN else y

    x = pc.reshape(B * N, -1)
    a = conv(x, "conv1", N)
    a = conv(a, "conv2", N)
    pf = conv(a, "conv3", N)
    a = conv(pf, "conv4", N)
    a = conv(a, "conv5", N)
    g = a.reshape(B, N, -1).max(1)                                        # MaxPool2D :82
    g = np.concatenate([g, one_hot.astype(dtype)], 1)                    # :84
    cat = np.concatenate([pf.reshape(B, N, -1), np.broadcast_to(g[:, None], (B, N, g.shape[1]))], 2)  # :85-86
    a = conv(cat.reshape(B * N, -1), "conv6", N)                         # :88
    a = conv(a, "conv7", N)
    a = conv(a, "conv8", N)
    a = conv(a, "conv9", N)
    if dropout_keep is not None:
        a = a * np.asarray(dropout_keep, dtype).reshape(B * N, -1) * 2.0   # :92
    logits = conv(a, "conv10", N, relu=False).reshape(B, N, 2)           # :93
    obj, mean, mask, idx = V.point_cloud_masking(pc, logits, dtype=dtype, mask_override=mask_override)
    t = conv(obj.reshape(B * M, 3), "conv-reg1-stage1", M)
    t = conv(t, "conv-reg2-stage1", M)
    t = conv(t, "conv-reg3-stage1", M)
    t = np.concatenate([t.reshape(B, M, -1).max(1), one_hot.astype(dtype)], 1)
    t = conv(t, "fc1-stage1", 1)
    t = conv(t, "fc2-stage1", 1)
    delta = output_layer(t, "fc3-stage1")                                # layers.Dense(3) :136
    stage1 = delta + mean
    # models/frustum_pointnet.py:210-214 forms the re-centred points and then feeds the box
    # stack the un-shifted ones (recentre_box_input=False); FPointNet_Bayes.py:99-101 shifts
    box_in = obj - delta[:, None, :] if recentre_box_input else obj
    t = conv(box_in.reshape(B * M, 3), "conv-reg1", M)
    t = conv(t, "conv-reg2", M)
    t = conv(t, "conv-reg3", M)
    t = conv(t, "conv-reg4", M)
    t = np.concatenate([t.reshape(B, M, -1).max(1), one_hot.astype(dtype)], 1)
    t = conv(t, "fc1", 1)
    t = conv(t, "fc2", 1)
    out = output_layer(t, "fc3")                                         # layers.Dense(59) :115
    ep = {"mask_logits": logits, "mask": mask, "object_point_indices": idx, "mask_xyz_mean": mean,
          "center_delta": delta, "stage1_center": stage1, "box_output": out}
    V.parse_output(out, ep, dtype)
    ep["center"] = ep["center_boxnet"] + stage1
    return ep


# --- ensemble statistics ---------------------------------------------------------

def decode_detection(box_output, stage1_center):
    """Per-sample detection vector v = (cx,cy,cz, l,w,h, heading): centre =
    center_boxnet + stage1_center (frustum_pointnets_v1.py:186); size = mean size of
    the arg-max size cluster + its residual; heading = bin centre of the arg-max
    heading bin + its residual (scripts/provider.py:73-82 class2angle without the
    to_label_format wrap, :101-106 class2size)."""
    NH, NS = V.NUM_HEADING_BIN, V.NUM_SIZE_CLUSTER
    out = np.asarray(box_output, np.float64)
    center = out[..., 0:3] + np.asarray(stage1_center, np.float64)
    hs = out[..., 3:3 + NH]
    hr = out[..., 3 + NH:3 + 2 * NH] * (np.pi / NH)
    ss = out[..., 3 + 2 * NH:3 + 2 * NH + NS]
    sr = out[..., 3 + 2 * NH + NS:].reshape(out.shape[:-1] + (NS, 3)) * V.G_MEAN_SIZE_ARR
    hc = hs.argmax(-1)
    sc = ss.argmax(-1)
    heading = hc * (2 * np.pi / NH) + np.take_along_axis(hr, hc[..., None], -1)[..., 0]
    size = V.G_MEAN_SIZE_ARR[sc] + np.take_along_axis(sr, sc[..., None, None].repeat(3, -1), -2)[..., 0, :]
    return np.concatenate([center, size, heading[..., None]], -1), hs, ss


def mc_statistics(box_output, stage1_center):
    """box_output (S,B,59), stage1_center (S,B,3) -> per-frustum ensemble statistics
    of the report (section 3): mean detection (7), total variance = trace of
    C = 1/S sum v v^T - mean mean^T (eq. 8), and the Shannon entropies of the
    sample-mean softmax probabilities of the heading and size classifiers."""
    v, hs, ss = decode_detection(box_output, stage1_center)
    mean = v.mean(0)
    tv = ((v ** 2).mean(0) - mean ** 2).sum(-1)
    ph = V.softmax(hs).mean(0)
    ps = V.softmax(ss).mean(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        eh = -np.sum(np.where(ph > 0, ph * np.log(ph), 0.0), -1)
        es = -np.sum(np.where(ps > 0, ps * np.log(ps), 0.0), -1)
    return {"mean": mean, "total_variance": tv, "heading_entropy": eh, "size_entropy": es,
            "heading_prob": ph, "size_prob": ps}
