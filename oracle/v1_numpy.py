"""numpy restatement of the Frustum-PointNet v1 inference graph.

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py as the checker / CPU baseline.  The product package
never imports this module.

PARITY UNPINNED by the reference's own tests: the reference holds no forward-value
test, golden vector or checkpoint for this graph, and TensorFlow is not installed,
so the restatement below cannot be compared with an executed reference.  It follows
the reference text layer by layer; each function cites the lines it restates.

Parameters are a dict of numpy arrays keyed by the reference's variable scopes:
  '<scope>/weights' (Cin,Cout)   [TF stores conv kernels as (1,1,Cin,Cout)]
  '<scope>/biases'  (Cout,)
  '<scope>/bn/{gamma,beta,moving_mean,moving_variance}' (Cout,)   when the layer has BN
"""
import numpy as np

NUM_HEADING_BIN = 12     # models/model_util.py:14
NUM_SIZE_CLUSTER = 8     # models/model_util.py:15
NUM_OBJECT_POINT = 512   # models/model_util.py:16
BN_EPS = 1e-3            # tf.contrib.layers.batch_norm default, models/tf_util.py:527-531

# models/model_util.py:34-48, class order Car, Van, Truck, Pedestrian,
# Person_sitting, Cyclist, Tram, Misc; each row is (l, w, h)
G_MEAN_SIZE_ARR = np.array([
    [3.88311640418, 1.62856739989, 1.52563191462],
    [5.06763659, 1.9007158, 2.20532825],
    [10.13586957, 2.58549199, 3.2520595],
    [0.84422524, 0.66068622, 1.76255119],
    [0.80057803, 0.5983815, 1.27450867],
    [1.76282397, 0.59706367, 1.73698127],
    [16.17150617, 2.53246914, 3.53079012],
    [3.64300781, 1.54298177, 1.92320313]], dtype=np.float64)

SEG_LAYERS = [("conv1", 64), ("conv2", 64), ("conv3", 64), ("conv4", 128), ("conv5", 1024)]
SEG_TAIL = [("conv6", 512), ("conv7", 256), ("conv8", 128), ("conv9", 128)]
TNET_CONV = [("conv-reg1-stage1", 128), ("conv-reg2-stage1", 128), ("conv-reg3-stage1", 256)]
TNET_FC = [("fc1-stage1", 256), ("fc2-stage1", 128)]
BOX_CONV = [("conv-reg1", 128), ("conv-reg2", 128), ("conv-reg3", 256), ("conv-reg4", 512)]
BOX_FC = [("fc1", 512), ("fc2", 256)]


def dense(x, params, scope, bn, relu, dtype):
    """tf_util.conv2d with a 1x1 kernel (models/tf_util.py:120-185) and
    tf_util.fully_connected (:327-363) are the same arithmetic on the last axis:
    matmul, bias_add (:176 / :356), optional batch norm in inference mode
    (:178-181 -> :527-531; tf.nn.batch_normalization form
    x*inv + (beta - mean*inv) with inv = gamma*rsqrt(var+eps)), optional ReLU."""
    w = params[scope + "/weights"].astype(dtype)
    y = x.astype(dtype) @ w.reshape(-1, w.shape[-1])
    y = y + params[scope + "/biases"].astype(dtype)
    if bn:
        gamma = params[scope + "/bn/gamma"].astype(dtype)
        beta = params[scope + "/bn/beta"].astype(dtype)
        mean = params[scope + "/bn/moving_mean"].astype(dtype)
        var = params[scope + "/bn/moving_variance"].astype(dtype)
        inv = gamma / np.sqrt(var + dtype(BN_EPS))
        y = y * inv + (beta - mean * inv)
    if relu:
        y = np.maximum(y, dtype(0))
    return y


def summed_magnitude(x, params, scope):
    """sum_k |x_k| |w_kj| + |b_j| for a layer without batch norm: the magnitude of what the
    layer's output sums.  The rounding error of a sum is judged against it (with random
    weights the sum itself can be a hundredth of it), tests of the bf16 path scale by it."""
    w = np.abs(params[scope + "/weights"].astype(np.float64))
    return np.abs(np.asarray(x, np.float64)) @ w.reshape(-1, w.shape[-1]) + np.abs(params[scope + "/biases"].astype(np.float64))


def instance_seg_v1(point_cloud, one_hot, params, bn=True, dtype=np.float32):
    """get_instance_seg_v1_net, models/frustum_pointnets_v1.py:20-90."""
    B, N, _ = point_cloud.shape
    net = point_cloud
    feats = {}
    for scope, _ in SEG_LAYERS:                       # :41-60
        net = dense(net, params, scope, bn, True, dtype)
        feats[scope] = net
    point_feat = feats["conv3"]                       # :49
    global_feat = net.max(axis=1)                     # :61  (B,1024)
    global_feat = np.concatenate([global_feat, one_hot.astype(dtype)], axis=1)    # :64
    expand = np.broadcast_to(global_feat[:, None, :], (B, N, global_feat.shape[1]))  # :65
    net = np.concatenate([point_feat, expand], axis=2)                            # :66
    for scope, _ in SEG_TAIL:                         # :68-83
        net = dense(net, params, scope, bn, True, dtype)
    # dropout :84 is the identity at inference (models/tf_util.py:612-614)
    logits = dense(net, params, "conv10", False, False, dtype)                    # :86-89
    return logits


def select_object_points(mask, npoints=NUM_OBJECT_POINT, policy="first", rng=None):
    """mask_to_indices inside tf_gather_object_pc, models/model_util.py:66-82.

    policy 'reference': numpy's global-RNG choice/shuffle exactly as written there
      (pass rng=np.random or a RandomState).
    policy 'first' (what this build specifies): with more than npoints positives
      keep the first npoints in index order; with fewer, cycle through them.  For
      count <= npoints the selected SET equals the reference's for every RNG
      state, and everything downstream is a per-point MLP followed by a max over
      points, so the result does not depend on order or repetition.
    policy 'spread' (the build's other option for count > npoints): positive number
      floor(t * count / npoints) for slot t, evenly spaced over the positives.
    An empty mask gives all-zero indices in all of them (:67,71)."""
    B, _ = mask.shape
    idx = np.zeros((B, npoints), np.int32)
    for i in range(B):
        pos = np.where(mask[i] > 0.5)[0]
        if len(pos) == 0:
            continue
        if policy == "reference":
            if len(pos) > npoints:
                choice = rng.choice(len(pos), npoints, replace=False)
            else:
                choice = rng.choice(len(pos), npoints - len(pos), replace=True)
                choice = np.concatenate((np.arange(len(pos)), choice))
            rng.shuffle(choice)
        elif len(pos) <= npoints:
            choice = np.arange(npoints) % len(pos)
        elif policy == "spread":
            choice = (np.arange(npoints, dtype=np.int64) * len(pos)) // npoints
        else:
            choice = np.arange(npoints)
        idx[i] = pos[choice]
    return idx


def point_cloud_masking(point_cloud, logits, policy="first", rng=None, dtype=np.float32,
                        mask_override=None):
    """point_cloud_masking, models/model_util.py:275-319 (xyz_only=True)."""
    mask = (logits[:, :, 0] < logits[:, :, 1]).astype(dtype)          # :292-294
    if mask_override is not None:
        mask = mask_override.astype(dtype)
    count = mask.sum(axis=1, keepdims=True)                            # :295
    xyz = point_cloud[:, :, :3].astype(dtype)                          # :297
    mean = (mask[:, :, None] * xyz).sum(axis=1) / np.maximum(count, dtype(1))   # :298-302
    stage1 = xyz - mean[:, None, :]                                    # :305-306
    idx = select_object_points(mask, NUM_OBJECT_POINT, policy, rng)    # :315
    obj = np.take_along_axis(stage1, idx[:, :, None].astype(np.int64), axis=1)
    return obj, mean, mask, idx


def center_regression_net(obj_pc, one_hot, params, bn=True, dtype=np.float32, magnitude=None):
    """get_center_regression_net (T-Net), models/model_util.py:322-357.
    magnitude: a dict that receives summed_magnitude of the output layer under 'center_delta'."""
    net = obj_pc
    for scope, _ in TNET_CONV:                        # :335-346
        net = dense(net, params, scope, bn, True, dtype)
    net = net.max(axis=1)                             # :347-349
    net = np.concatenate([net, one_hot.astype(dtype)], axis=1)        # :350
    for scope, _ in TNET_FC:                          # :351-354
        net = dense(net, params, scope, bn, True, dtype)
    if magnitude is not None:
        magnitude["center_delta"] = summed_magnitude(net, params, "fc3-stage1")
    return dense(net, params, "fc3-stage1", False, False, dtype)      # :355


def box_estimation_v1(obj_pc, one_hot, params, bn=True, dtype=np.float32):
    """get_3d_box_estimation_v1_net, models/frustum_pointnets_v1.py:93-138."""
    net = obj_pc
    for scope, _ in BOX_CONV:                         # :108-123
        net = dense(net, params, scope, bn, True, dtype)
    net = net.max(axis=1)                             # :124-126
    net = np.concatenate([net, one_hot.astype(dtype)], axis=1)        # :127
    for scope, _ in BOX_FC:                           # :128-131
        net = dense(net, params, scope, bn, True, dtype)
    return dense(net, params, "fc3", False, False, dtype)             # :136


def parse_output(output, end_points, dtype=np.float32):
    """parse_output_to_tensors, models/model_util.py:211-243."""
    NH, NS = NUM_HEADING_BIN, NUM_SIZE_CLUSTER
    B = output.shape[0]
    end_points["center_boxnet"] = output[:, 0:3]
    end_points["heading_scores"] = output[:, 3:3 + NH]
    hrn = output[:, 3 + NH:3 + 2 * NH]
    end_points["heading_residuals_normalized"] = hrn
    end_points["heading_residuals"] = hrn * dtype(np.pi / NH)           # :230
    end_points["size_scores"] = output[:, 3 + 2 * NH:3 + 2 * NH + NS]
    srn = output[:, 3 + 2 * NH + NS:3 + 2 * NH + 4 * NS].reshape(B, NS, 3)
    end_points["size_residuals_normalized"] = srn
    end_points["size_residuals"] = srn * G_MEAN_SIZE_ARR.astype(np.float32).astype(dtype)[None]  # :240
    return end_points


def get_model(point_cloud, one_hot, params, bn=True, policy="first", rng=None,
              dtype=np.float32, mask_override=None):
    """get_model, models/frustum_pointnets_v1.py:141-188, inference mode.
    bn=False gives the Keras variant's graph (models/FPointNet_Keras.py:25-37,
    batch norm commented out)."""
    dtype = np.dtype(dtype).type
    ep = {}
    logits = instance_seg_v1(point_cloud, one_hot, params, bn, dtype)            # :159-162
    ep["mask_logits"] = logits
    obj, mean, mask, idx = point_cloud_masking(point_cloud, logits, policy, rng, dtype,
                                               mask_override)                    # :166-167
    ep["mask"] = mask
    ep["object_point_indices"] = idx
    ep["mask_xyz_mean"] = mean
    delta = center_regression_net(obj, one_hot, params, bn, dtype)               # :170-172
    ep["center_delta"] = delta
    stage1_center = delta + mean                                                 # :173
    ep["stage1_center"] = stage1_center
    obj_new = obj - delta[:, None, :]                                            # :176-177
    output = box_estimation_v1(obj_new, one_hot, params, bn, dtype)              # :180-182
    ep["box_output"] = output
    parse_output(output, ep, dtype)                                              # :185
    ep["center"] = ep["center_boxnet"] + stage1_center                           # :186
    return ep


def softmax(x):
    """scripts/test.py:38-43."""
    p = np.exp(x - np.max(x, axis=-1, keepdims=True))
    return p / np.sum(p, axis=-1, keepdims=True)


def inference_postprocess(logits, centers, heading_scores, heading_residuals,
                          size_scores, size_residuals):
    """The arithmetic of scripts/test.py:inference, :125-162, for one batch:
    float64 buffers (np.zeros default dtype :125-131), softmax/argmax scoring
    :145-152, class picks and residual gathers :157-160.  mask_mean_prob divides by
    the positive count without a guard (:148): a frustum with no positive point
    yields NaN, as in the reference."""
    logits = np.asarray(logits, np.float64)
    centers = np.asarray(centers, np.float64)
    heading_scores64 = np.asarray(heading_scores, np.float64)
    heading_residuals = np.asarray(heading_residuals, np.float64)
    size_scores64 = np.asarray(size_scores, np.float64)
    size_residuals = np.asarray(size_residuals, np.float64)
    # the score block runs on the model's own float32 outputs (:145-150)
    bl = np.asarray(logits, np.float32)
    seg_prob = softmax(bl)[:, :, 1]
    seg_mask = np.argmax(bl, 2)
    with np.errstate(invalid="ignore", divide="ignore"):
        mask_mean_prob = np.sum(seg_prob * seg_mask, 1) / np.sum(seg_mask, 1)
    heading_prob = np.max(softmax(np.asarray(heading_scores, np.float32)), 1)
    size_prob = np.max(softmax(np.asarray(size_scores, np.float32)), 1)
    scores = ((mask_mean_prob + heading_prob + size_prob) / 3).astype(np.float64)
    heading_cls = np.argmax(heading_scores64, 1)
    size_cls = np.argmax(size_scores64, 1)
    n = logits.shape[0]
    heading_res = np.array([heading_residuals[i, heading_cls[i]] for i in range(n)])
    size_res = np.vstack([size_residuals[i, size_cls[i], :] for i in range(n)])
    return np.argmax(logits, 2), centers, heading_cls, heading_res, size_cls, size_res, scores


# mean sizes (l,w,h) by size-cluster index, scripts/provider.py:25-33 (= models/model_util.py:38-48)
_MEAN_SIZE = np.array([[3.88311640418, 1.62856739989, 1.52563191462], [5.06763659, 1.9007158, 2.20532825],
                       [10.13586957, 2.58549199, 3.2520595], [0.84422524, 0.66068622, 1.76255119],
                       [0.80057803, 0.5983815, 1.27450867], [1.76282397, 0.59706367, 1.73698127],
                       [16.17150617, 2.53246914, 3.53079012], [3.64300781, 1.54298177, 1.92320313]])


def label_fields(centers, heading_cls, heading_res, size_cls, size_res, rot_angle, trig="exact"):
    """The label conversion of scripts/provider.py:366-373 for a batch, written out
    element by element in float64 (no shared code with the product's fpn/provider.py):
      l,w,h = mean_size[size_cls] + size_res                                 (:101-104)
      angle = heading_cls * (2 pi / 12) + heading_res; angle > pi: -2 pi     (:73-82)
      ry    = angle + rot_angle                                              (:370)
      tx,tz = the centre's x,z rotated about y by -rot_angle                 (:35-48, :371)
      ty    = centre y + h/2                                                 (:372)
    -> (B,7) columns h,w,l,tx,ty,tz,ry.
    rot_angle is a float32 array (the record files hold float32): the reference evaluates
    np.cos / np.sin on the float32 scalar and gets float32 numbers.  trig="exact": the
    correctly rounded float32 cosine and sine (float64 evaluation rounded once);
    trig="numpy": numpy's own float32 routines (within an ulp or two of the former)."""
    centers = np.asarray(centers, np.float64)
    size = _MEAN_SIZE[np.asarray(size_cls)] + np.asarray(size_res, np.float64)
    angle = np.asarray(heading_cls).astype(np.float64) * (2 * np.pi / 12.0) + np.asarray(heading_res, np.float64)
    angle = np.where(angle > np.pi, angle - 2 * np.pi, angle)
    rot = np.asarray(rot_angle, np.float32)
    neg = -rot
    if trig == "exact":
        c = np.cos(neg.astype(np.float64)).astype(np.float32).astype(np.float64)
        s = np.sin(neg.astype(np.float64)).astype(np.float32).astype(np.float64)
    else:
        c = np.cos(neg).astype(np.float64)
        s = np.sin(neg).astype(np.float64)
    x, y, z = centers[:, 0], centers[:, 1], centers[:, 2]
    # row vector [x z] times transpose([[c,-s],[s,c]]) = [x*c + z*(-s), x*s + z*c]
    tx = x * c + z * (-s)
    tz = x * s + z * c
    h = size[:, 2]
    return np.stack([h, size[:, 1], size[:, 0], tx, y + h / 2.0, tz, angle + rot.astype(np.float64)], axis=1)
