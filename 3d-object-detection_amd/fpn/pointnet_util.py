"""PointNet++ layers: counterpart of models/pointnet_util.py (sample_and_group :21,
sample_and_group_all :58, pointnet_sa_module :86, pointnet_sa_module_msg :155,
pointnet_fp_module :198), inference mode.

Same signatures and return values; tensors are torch tensors on a ROCm device.
Inside the modules the work is fused where the reference issues separate ops:
  * grouping: gather(xyz) - centre, gather(features) and their concatenation are one
    kernel (fpn_group_concat);
  * every MLP layer is one call (matmul + bias + batch norm + ReLU), and the max over
    the nsample axis is fused into the last layer's epilogue;
  * feature propagation: three_nn, the inverse-distance weights and
    three_interpolate are one kernel (fpn_three_nn_interpolate).
Concatenation orders follow the reference exactly: SSG [xyz, features] (:49), MSG
[features, xyz] (:183), group_all [xyz, features] (:77), FP [interpolated, points1]
(:218).
"""
import torch

from . import _lib, ops, tf_util, variables
from .tf_ops.tf_grouping import group_point, knn_point, query_ball_point, query_ball_point_multi
from .tf_ops.tf_interpolate import three_nn_interpolate
from .tf_ops.tf_sampling import farthest_point_sample, gather_point

# Set-abstraction modules with max pooling and a three-layer MLP run as one fused kernel
# when a kernel exists for their shape; False forces the per-layer path (used by tests).
FUSE_SA = True
# Operand type of the fused kernel's matrix products: "fp32", or "bf16" (bf16 weights and
# activations between the layers, fp32 accumulation; the per-layer path stays fp32).
SA_PRECISION = "fp32"
# Under SA_PRECISION = "bf16": the interpolated features of the last feature propagation reach
# the fused head kernel as bf16 (False: as float, rounded by the head kernel - same results).
INTERP_BF16 = True


def group_concat(xyz, new_xyz, points, idx, xyz_first=True):
    """(B,N,3), (B,M,3), (B,N,C) or None, (B,M,ns) -> (B,M,ns,3+C): neighbour
    coordinates relative to the query joined with neighbour features."""
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    c = 0 if points is None else points.shape[2]
    out = torch.empty((b, m, ns, 3 + c), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.load().fpn_group_concat(b, n, m, ns, c, int(bool(xyz_first)), _lib.ptr(xyz),
                                                _lib.ptr(new_xyz), _lib.ptr(points), _lib.ptr(idx),
                                                _lib.ptr(out), _lib.stream_of(xyz)))
    return out


def _sample_and_query(npoint, radius, nsample, xyz, knn):
    """The sampling half of sample_and_group (pointnet_util.py:39-43): new_xyz, idx."""
    new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
    if knn:
        _, idx = knn_point(nsample, xyz, new_xyz)
    else:
        idx, _ = query_ball_point(radius, nsample, xyz, new_xyz)
    return new_xyz, idx


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True, sampled=None):
    """Reference: pointnet_util.py:21-55.  Returns new_xyz (B,npoint,3), new_points
    (B,npoint,nsample,3+C), idx (B,npoint,nsample), grouped_xyz (B,npoint,nsample,3).
    `sampled` = (new_xyz, idx) already computed for these arguments (skips the sampling
    and the neighbour search)."""
    xyz = xyz.contiguous()
    new_xyz, idx = sampled if sampled is not None else _sample_and_query(npoint, radius, nsample, xyz, knn)
    if points is not None and use_xyz:
        new_points = group_concat(xyz, new_xyz, points.contiguous(), idx, xyz_first=True)
        grouped_xyz = new_points[..., :3]
    else:
        grouped_xyz = group_concat(xyz, new_xyz, None, idx)
        new_points = grouped_xyz if points is None else group_point(points.contiguous(), idx)
    return new_xyz, new_points, idx, grouped_xyz


_GROUP_ALL_CONSTANTS = {}


def _group_all_constants(b, n, device, dense=False):
    """The centroid (0,0,0) and the index table 0..n-1 of sample_and_group_all: the same
    for every call of a shape, so they are made once.  The index table is handed out as
    an expanded (stride-0) view, which torch refuses to write in place (`dense`: as a table
    with a row per frustum, which a kernel can read; do not write it); the caller gets its
    own copy of the centroid."""
    key = (b, n, device, dense)
    hit = _GROUP_ALL_CONSTANTS.get(key)
    if hit is None:
        if len(_GROUP_ALL_CONSTANTS) >= 64:
            _GROUP_ALL_CONSTANTS.clear()
        idx = torch.arange(n, dtype=torch.int32, device=device).reshape(1, 1, n).expand(b, 1, n)
        hit = (torch.zeros((b, 1, 3), dtype=torch.float32, device=device), idx.contiguous() if dense else idx)
        _GROUP_ALL_CONSTANTS[key] = hit
    return hit


def sample_and_group_all(xyz, points, use_xyz=True):
    """Reference: pointnet_util.py:58-83: one group holding every point, centroid (0,0,0),
    no translation."""
    b, n, _ = xyz.shape
    new_xyz, idx = _group_all_constants(b, n, xyz.device)
    new_xyz = new_xyz.clone()
    grouped_xyz = xyz.reshape(b, 1, n, 3)
    if points is not None:
        new_points = torch.cat([xyz, points], dim=2) if use_xyz else points
        new_points = new_points.unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def _layer_params(store, name, cin, cout, bn):
    """Create (or fetch) the variables of one conv2d layer under `name` and return its
    weight with the folded epilogue constants (scale may be None)."""
    w = store.get_variable(name + "/weights", (cin, cout), "xavier")
    store.get_variable(name + "/biases", (cout,), "zeros")
    if bn:
        store.get_variable(name + "/bn/gamma", (cout,), "ones")
        store.get_variable(name + "/bn/beta", (cout,), "zeros")
        store.get_variable(name + "/bn/moving_mean", (cout,), "zeros")
        store.get_variable(name + "/bn/moving_variance", (cout,), "ones")
    scale, shift = store.folded(name, bn)
    return w, scale, shift


def _round_up(v, q):
    return (v + q - 1) // q * q


def _sa_chain_plan(store, names, cf, widths, bn, xyz_first):
    """Weights of a 3-layer set-abstraction MLP in the form fpn_sa_chain_f32 reads them
    (padded widths, rows of the first layer reordered to [features | xyz] / [xyz |
    features], chunked), or None when no fused kernel covers the shape.  Cached on the
    store until its variables change."""
    if len(widths) != 3:
        return None
    # padded widths: the middle layer to a multiple of 32 where a kernel exists for that
    # (96 stays 96), else of 64
    pad = None
    supported = _lib.load().fpn_sa_chain_bf16_supported if SA_PRECISION == "bf16" else _lib.load().fpn_sa_chain_supported
    # wide features go through their rows of the first layer once per point, outside the kernel
    hoist = HOIST_FEATURES and cf >= 32
    narrow = (_round_up(widths[0], 32), _round_up(widths[1], 32), _round_up(widths[2], 64))
    for mid in (0, 32, 64):
        # (first the widths rounded to whole 32-channel tiles only: a kernel exists for 32-32-64 in fp32)
        cand = narrow if mid == 0 else (_round_up(widths[0], 64), _round_up(widths[1], mid), _round_up(widths[2], 128))
        if mid == 0 and (cf >= 32 or not supported(cf, *cand)):
            continue
        if hoist and supported(0, *cand):
            pad = cand
            break
        if supported(cf, *cand):
            pad = cand
            hoist = False
            break
    if pad is None:
        # the variables are created by the per-layer path
        return None
    cache = store.__dict__.setdefault("_sa_chain_cache", {})
    key = (names[0], cf, xyz_first, SA_PRECISION, hoist)
    hit = cache.get(key)
    if hit is not None and hit["version"] == store.version:
        return hit
    cins = (3 + cf, widths[0], widths[1])
    params = [_layer_params(store, names[i], cins[i], widths[i], bn) for i in range(3)]
    version = store.version          # creating variables may have bumped it
    dev = params[0][0].device

    def padded(w, scale, shift, rows, cols):
        wp = torch.zeros((rows, cols), dtype=torch.float32, device=dev)
        wp[:w.shape[0], :w.shape[1]] = w
        sp = torch.zeros(cols, dtype=torch.float32, device=dev)
        sp[:w.shape[1]] = 1.0 if scale is None else scale
        hp = torch.zeros(cols, dtype=torch.float32, device=dev)
        hp[:w.shape[1]] = shift
        return wp, sp, hp
    w1, s1, h1 = params[0]
    xyz_rows = w1[:3] if xyz_first else w1[cf:cf + 3]
    feat_rows = w1[3:] if xyz_first else w1[:cf]
    if hoist:
        w1k, s1p, h1p = padded(xyz_rows, s1, h1, 3, pad[0])
    elif cf >= 32:
        w1k, s1p, h1p = padded(torch.cat([feat_rows, xyz_rows], 0), s1, h1, cf + 32, pad[0])
        w1k = ops.tile_weights_reg(w1k, 0)
    else:
        w1k, s1p, h1p = padded(torch.cat([xyz_rows, feat_rows], 0), s1, h1, 3 + cf, pad[0])
    w2p, s2p, h2p = padded(*params[1], pad[0], pad[1])
    w3p, s3p, h3p = padded(*params[2], pad[1], pad[2])
    if SA_PRECISION == "bf16":
        # one bf16 weight stream (scales folded into the columns); fp32 first layer when
        # it reads the coordinates and features straight from memory (cf < 32)
        if hoist:
            stream = ops.chain_stream_bf16([(w2p, s2p), (w3p, s3p)])
            tensors = [w1k.contiguous(), s1p, h1p, h2p, h3p]
        elif cf >= 32:
            # the kernel feeds every coordinate as two bf16 numbers (rows cf..cf+2 and cf+4..cf+6): the same weights twice
            w1p = padded(torch.cat([feat_rows, xyz_rows, torch.zeros_like(xyz_rows[:1]), xyz_rows], 0), s1, h1, cf + 32,
                         pad[0])[0]
            stream = ops.chain_stream_bf16([(w1p, s1p), (w2p, s2p), (w3p, s3p)])
            tensors = [h1p, h2p, h3p]
        else:
            if pad[1] % 64:
                # the stream pairs output tiles: the middle layer goes in with its columns zero-padded to a
                # multiple of 64 (the kernel leaves the padding tile out), the last layer with pad[1] rows
                w2s, s2s, _ = padded(*params[1], pad[0], _round_up(pad[1], 64))
                stream = ops.chain_stream_bf16([(w2s, s2s), (w3p, s3p)])
            else:
                stream = ops.chain_stream_bf16([(w2p, s2p), (w3p, s3p)])
            tensors = [w1k.contiguous(), s1p, h1p, h2p, h3p]
        plan = {"version": version, "tensors": tensors, "table": ops.pointer_table(tensors), "pad": pad,
                "stream": stream}
        if hoist:       # the features' rows of the first layer, scaled like the stream's weights, columns padded
            plan["w1f"] = padded(feat_rows, None, h1, cf, pad[0])[0].contiguous()
            plan["s1"] = s1p
            plan["w1f_bf16"] = ops.pack_mlp_bf16(plan["w1f"])
    else:
        tensors = [w1k.contiguous(), s1p, h1p, ops.tile_weights_reg(w2p, 0), s2p, h2p, ops.tile_weights_reg(w3p, 0),
                   s3p, h3p]
        plan = {"version": version, "tensors": tensors, "table": ops.pointer_table(tensors), "pad": pad}
        if hoist:       # the features' rows of the first layer, columns padded like the layer: no scale, no shift
            plan["w1f"] = padded(feat_rows, None, h1, cf, pad[0])[0].contiguous()
    cache[key] = plan
    return plan


# A set-abstraction layer is linear before its batch norm and ReLU, and a point is gathered
# into many balls: wide point features are multiplied with their rows of the first layer ONCE
# PER POINT and the fused kernel gathers that product (fpn_sa_chain_pre_f32) instead of the
# features.  False: the kernel gathers the features and multiplies per gathered row (the
# reference's order of operations; the same results up to fp32 rounding).
HOIST_FEATURES = True

# A test switch: fill with NaN every set-abstraction output that is handed to a kernel
# without being cleared (the kernel is trusted to store all of it).
POISON_UNCLEARED = False

# Under SA_PRECISION = "bf16": point features wider than 32 channels are converted to the
# kernel's operand form once per module instead of once per gathered row (False: gathered as
# float and converted in the kernel - same results).
PACK_FEATURES = True


def _pack_features(points):
    """(B,N,C) float32 features, C a multiple of 32 -> fpn_sa_pack_features_bf16's table."""
    points = points.contiguous()
    b, n, c = points.shape
    packed = torch.empty((b, n, c), dtype=torch.bfloat16, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.load().fpn_sa_pack_features_bf16(b * n, c, _lib.ptr(points), _lib.ptr(packed),
                                                         _lib.stream_of(points)))
    return packed


def _sa_chain(xyz, new_xyz, points, idx, widths, scope_fmt, bn, out2d, xyz_first, zeroed=False, shared=None):
    """Grouping + 3-layer MLP + max over the samples in one kernel (fpn_sa_chain_f32).
    out2d (b*m, widths[-1]) may be a column slice of a wider buffer; the kernel merges its
    maxima into it, so it must start at zero (`zeroed`: the caller has cleared it).  Returns
    False when the shape has no fused kernel (the caller then runs the per-layer path).
    `shared`: a dict the scales of a multi-scale module pass along - the packed bf16 feature
    table ("packed", made by the first scale that gathers features) and the first-layer products
    of all scales ("pre": {first-layer scope: (matrix, column offset)}, see _msg_products)."""
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    cf = 0 if points is None else points.shape[2]
    if ns % 32 != 0 or len(widths) != 3 or widths[2] % 64 != 0:     # (the kernels write whole pairs of 32-wide tiles)
        return False
    store = variables.get_default_store()
    names = [variables.scoped(scope_fmt % i) for i in range(3)]
    plan = _sa_chain_plan(store, names, cf, widths, bn, xyz_first)
    if plan is None:
        return False
    pad = plan["pad"]
    hoisted = "w1f" in plan
    if not zeroed:
        # the persistent kernels store every ball's row; the others merge maxima into a cleared buffer
        ask = _lib.load().fpn_sa_chain_bf16_overwrites if "stream" in plan else _lib.load().fpn_sa_chain_overwrites
        if not ask(b * m * ns, 0 if hoisted else cf, pad[0], pad[1], pad[2]):
            out2d.zero_()
        elif POISON_UNCLEARED:
            out2d.fill_(float("nan"))
    ldo = out2d.stride(0) if out2d.shape[0] > 1 else max(out2d.stride(0), out2d.shape[1])
    obs = ops.launch_observer
    tok = None
    if obs is not None:       # algorithmic work: the unpadded layer widths (hoisted: the features' rows of the
        # first layer are another launch's work, once per point)
        macs = (3 + (0 if hoisted else cf)) * widths[0] + widths[0] * widths[1] + widths[1] * widths[2]
        tok = obs.begin("sa_chain%s[cf=%d,%d-%d-%d%s]" % ("_bf16" if "stream" in plan else "", 0 if hoisted else cf, *widths,
                                                           ",pre=%d" % cf if hoisted else ""),
                        b * m * ns, 0, 0, macs=b * m * ns * macs)
    with torch.cuda.device(xyz.device):
        pre = None
        if hoisted:
            made = (shared or {}).get("pre", {}).get(names[0])
            if made is not None:        # a column block of the module's one matrix of products
                pre = made[0][:, made[1]:made[1] + pad[0]]
            elif "stream" in plan:
                pre = ops.shared_mlp(points.reshape(b * n, cf), plan["w1f"], plan["s1"], None, relu=False,
                                     wt_bf16=plan["w1f_bf16"])
            else:
                # (:184 before :187-190 in the reference's order: per gathered row; here per point, then gathered)
                pre = ops.shared_mlp(points.reshape(b * n, cf), plan["w1f"], None, None, relu=False)
        if "stream" in plan and hoisted:
            _lib.check(_lib.load().fpn_sa_chain_pre_bf16(b, n, m, ns, 0, pad[0], pad[1], pad[2], _lib.ptr(xyz),
                                                         _lib.ptr(new_xyz), None, _lib.ptr(pre), pre.stride(0), _lib.ptr(idx),
                                                         _lib.ptr(plan["stream"]), plan["table"], _lib.ptr(out2d), ldo,
                                                         int(widths[2]), _lib.stream_of(xyz)))
        elif "stream" in plan:
            packed = None
            if PACK_FEATURES and cf >= 32 and cf % 32 == 0:
                holder = shared if shared is not None else {}
                if holder.get("packed") is None:
                    holder["packed"] = _pack_features(points)
                packed = holder["packed"]
            entry = _lib.load().fpn_sa_chain_bf16 if packed is None else _lib.load().fpn_sa_chain_bf16p
            _lib.check(entry(b, n, m, ns, cf, pad[0], pad[1], pad[2], _lib.ptr(xyz), _lib.ptr(new_xyz),
                             _lib.ptr(points if packed is None else packed), _lib.ptr(idx), _lib.ptr(plan["stream"]),
                             plan["table"], _lib.ptr(out2d), ldo, int(widths[2]), _lib.stream_of(xyz)))
        elif hoisted:
            _lib.check(_lib.load().fpn_sa_chain_pre_f32(b, n, m, ns, 0, pad[0], pad[1], pad[2], _lib.ptr(xyz),
                                                        _lib.ptr(new_xyz), None, _lib.ptr(pre), pre.stride(0), _lib.ptr(idx),
                                                        plan["table"], _lib.ptr(out2d), ldo, int(widths[2]),
                                                        _lib.stream_of(xyz)))
        else:
            _lib.check(_lib.load().fpn_sa_chain_f32(b, n, m, ns, cf, pad[0], pad[1], pad[2], _lib.ptr(xyz),
                                                    _lib.ptr(new_xyz), _lib.ptr(points), _lib.ptr(idx), plan["table"],
                                                    _lib.ptr(out2d), ldo, int(widths[2]), _lib.stream_of(xyz)))
    if obs is not None:
        obs.end(tok)
    return True


# Rows from which a per-layer call takes the bf16 kernel under SA_PRECISION = "bf16".  Zero:
# the operand type of a layer does not depend on the batch size, so a frustum's results do
# not either (with few rows the fp32 few-row forms would be some microseconds faster).
BF16_LAYER_MIN_ROWS = 0


def _packed_bf16(store, name, w):
    """pack_mlp_bf16 of a layer's weights, cached on the store until its variables change
    (`name`: the cache key - the layer's scope, plus a suffix for a row slice)."""
    cache = store.__dict__.setdefault("_mlp_bf16_cache", {})
    hit = cache.get(name)
    if hit is None or hit[0] != store.version:
        hit = (store.version, ops.pack_mlp_bf16(w.contiguous()))
        cache[name] = hit
    return hit[1]


def _mlp_stack(x2d, widths, scope_fmt, bn, pool_rows=0, out=None, first=0, coords_in_first=False):
    """Shared MLP over the rows of x2d; the pooling over groups of pool_rows rows is
    fused into the last layer when the kernel supports that group size.  `out`, if
    given, receives the final result (it may be a column slice of a wider buffer).
    `first`: index of widths[0] in the scope names (layers before it ran elsewhere).
    Under SA_PRECISION = "bf16" layers with many rows run on bf16 operands, except one whose
    input holds coordinates (`coords_in_first`: the first layer's does)."""
    store = variables.get_default_store()
    fuse_pool = pool_rows and pool_rows % 32 == 0 and (pool_rows <= 128 and 128 % pool_rows == 0 or pool_rows % 128 == 0)
    for i, cout in enumerate(widths):
        name = variables.scoped(scope_fmt % (first + i))
        w, scale, shift = _layer_params(store, name, x2d.shape[1], cout, bn)
        last = i == len(widths) - 1
        wt = None
        if SA_PRECISION == "bf16" and x2d.shape[0] >= BF16_LAYER_MIN_ROWS and not (i == 0 and coords_in_first):
            wt = _packed_bf16(store, name, w)
        x2d = ops.shared_mlp(x2d, w, scale, shift, relu=True, pool_rows=pool_rows if (last and fuse_pool) else 0,
                             out=out if (last and (fuse_pool or not pool_rows)) else None, wt_bf16=wt)
    if pool_rows and not fuse_pool:
        x2d = ops.max_pool_rows(x2d, pool_rows, out=out)
    return x2d


def pointnet_sa_module(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, is_training, bn_decay,
                       scope, bn=True, pooling='max', knn=False, use_xyz=True, use_nchw=False):
    """Set abstraction.  Reference: pointnet_util.py:86-153.  Returns new_xyz
    (B,npoint,3), new_points (B,npoint,mlp[-1] or mlp2[-1]), idx (B,npoint,nsample)."""
    tf_util._inference_only(is_training)
    with variables.variable_scope(scope):
        sampled = None
        if not group_all and pooling == 'max' and mlp2 is None and FUSE_SA and (points is None or use_xyz):
            # grouping, MLP and max-pool in one kernel (:106-126) when the shape has one
            xyz = xyz.contiguous()
            sampled = _sample_and_query(npoint, radius, nsample, xyz, knn)
            new_xyz, idx = sampled
            pooled = torch.empty((xyz.shape[0] * npoint, mlp[-1]), dtype=torch.float32, device=xyz.device)
            if _sa_chain(xyz, new_xyz, None if points is None else points.contiguous(), idx.contiguous(), mlp,
                         'conv%d', bn, pooled, xyz_first=True):
                return new_xyz, pooled.reshape(xyz.shape[0], npoint, pooled.shape[-1]), idx
        if group_all and SA_PRECISION == "bf16" and FUSE_SA and points is not None and use_xyz \
                and pooling == 'max' and mlp2 is None and len(mlp) == 3:
            # group-all (:62-84, :108-109) as ONE ball per frustum of the fused kernel: every point in index order,
            # centre at the origin (the coordinates are not translated)
            b, n, _ = xyz.shape
            new_xyz, table = _group_all_constants(b, n, xyz.device, dense=True)
            pooled = torch.empty((b, mlp[-1]), dtype=torch.float32, device=xyz.device)
            if _sa_chain(xyz.contiguous(), new_xyz, points.contiguous(), table, mlp, 'conv%d', bn, pooled, xyz_first=True):
                return new_xyz.clone(), pooled.reshape(b, 1, pooled.shape[-1]), _group_all_constants(b, n, xyz.device)[1]
        if group_all and SA_PRECISION == "bf16" and FUSE_SA and points is not None and use_xyz \
                and pooling == 'max' and len(mlp) >= 2:
            # group-all in bf16: the first layer's input is [xyz | features] (:77); the three coordinate
            # columns go through the fp32 kernel (K = 3), the features through the bf16 kernel with that
            # product as a per-row bias - the concatenated tensor is not made
            b, n, _ = xyz.shape
            cfeat = points.shape[2]
            store = variables.get_default_store()
            name0 = variables.scoped('conv0')
            w, scale, shift = _layer_params(store, name0, 3 + cfeat, mlp[0], bn)
            new_xyz, idx = _group_all_constants(b, n, xyz.device)
            part = ops.shared_mlp(xyz.contiguous().reshape(b * n, 3), w[:3], None, None, relu=False)
            y = ops.shared_mlp(points.contiguous().reshape(b * n, cfeat), w[3:], scale, shift, relu=True, group_bias=part,
                               rows_per_group=1, wt_bf16=_packed_bf16(store, name0 + "#features", w[3:]))
            pooled = _mlp_stack(y, mlp[1:], 'conv%d', bn, pool_rows=n, first=1)
            if mlp2 is not None:
                pooled = _mlp_stack(pooled, mlp2, 'conv_post_%d', bn)
            return new_xyz.clone(), pooled.reshape(b, 1, pooled.shape[-1]), idx
        if group_all:
            nsample = xyz.shape[1]
            new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, use_xyz)
        else:
            # (the fused kernel declined the shape: its samples and ball indices are reused)
            new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, nsample, xyz, points,
                                                                     knn, use_xyz, sampled=sampled)
        b, m, ns, c = new_points.shape
        x2d = new_points.contiguous().reshape(b * m * ns, c)
        has_xyz = points is None or use_xyz       # the first layer's input holds coordinates
        if pooling == 'max':
            pooled = _mlp_stack(x2d, mlp, 'conv%d', bn, pool_rows=ns, coords_in_first=has_xyz)   # :116-126
        else:
            feat = _mlp_stack(x2d, mlp, 'conv%d', bn, coords_in_first=has_xyz)
            feat = feat.reshape(b, m, ns, feat.shape[-1])
            if pooling == 'avg':
                pooled = feat.mean(dim=2)
            elif pooling == 'weighted_avg':                                              # :129-135
                dists = torch.linalg.vector_norm(grouped_xyz, dim=-1, keepdim=True)
                exp_dists = torch.exp(-dists * 5)
                weights = exp_dists / exp_dists.sum(dim=2, keepdim=True)
                pooled = (feat * weights).sum(dim=2)
            elif pooling == 'max_and_avg':                                               # :136-139
                pooled = torch.cat([feat.mean(dim=2), feat.amax(dim=2)], dim=-1)
            else:
                raise ValueError("unknown pooling %r" % pooling)
            pooled = pooled.reshape(b * m, pooled.shape[-1]).contiguous()
        if mlp2 is not None:
            pooled = _mlp_stack(pooled, mlp2, 'conv_post_%d', bn)                        # :142-150
        return new_xyz, pooled.reshape(b, m, pooled.shape[-1]), idx


def _msg_products(points, names_list, mlp_list, bn):
    """The first-layer products (HOIST_FEATURES) of every scale of a multi-scale module in ONE
    launch over the points: the scales' feature rows of W1 side by side as the columns of one
    matrix.  Returns {first-layer scope: (products (b*n, total), column offset)} for the scales
    whose fused kernel takes products; {} when there are fewer than two."""
    b, n, cf = points.shape
    store = variables.get_default_store()
    plans = []
    for names, widths in zip(names_list, mlp_list):
        plan = _sa_chain_plan(store, names, cf, widths, bn, False) if len(widths) == 3 and widths[2] % 64 == 0 else None
        if plan is not None and "w1f" in plan:
            plans.append((names[0], plan))
    if len(plans) < 2:
        return {}
    cache = store.__dict__.setdefault("_msg_products_cache", {})
    key = (tuple(nm for nm, _ in plans), SA_PRECISION)
    hit = cache.get(key)
    if hit is None or hit["version"] != store.version:
        w = torch.cat([pl["w1f"] for _, pl in plans], dim=1).contiguous()
        bf16 = "stream" in plans[0][1]
        hit = {"version": store.version, "w": w,
               "scale": torch.cat([pl["s1"] for _, pl in plans]).contiguous() if bf16 else None,
               "wt": ops.pack_mlp_bf16(w) if bf16 else None}
        cache[key] = hit
    pre = ops.shared_mlp(points.reshape(b * n, cf), hit["w"], hit["scale"], None, relu=False, wt_bf16=hit["wt"])
    made, off = {}, 0
    for nm, pl in plans:
        made[nm] = (pre, off)
        off += pl["w1f"].shape[1]
    return made


def pointnet_sa_module_msg(xyz, points, npoint, radius_list, nsample_list, mlp_list, is_training, bn_decay,
                           scope, bn=True, use_xyz=True, use_nchw=False):
    """Multi-scale-grouping set abstraction.  Reference: pointnet_util.py:155-195.
    Returns new_xyz (B,npoint,3), new_points (B,npoint,sum of mlp[k][-1])."""
    tf_util._inference_only(is_training)
    with variables.variable_scope(scope):
        xyz = xyz.contiguous()
        new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
        b, m = new_xyz.shape[0], new_xyz.shape[1]
        width = sum(w[-1] for w in mlp_list)
        # (every scale fills its columns: a fused kernel that merges maxima clears them first, _sa_chain)
        out = torch.empty((b * m, width), dtype=torch.float32, device=xyz.device)
        shared = {}         # what the scales have in common: the packed feature table, the first-layer products
        if FUSE_SA and points is not None and use_xyz:
            points = points.contiguous()
            shared["pre"] = _msg_products(points, [[variables.scoped('conv%d_%d' % (i, j)) for j in range(3)]
                                                   for i in range(len(mlp_list))], mlp_list, bn)
        col = 0
        # all radii in one sweep over the points (the same index tables as one query per radius, :178)
        balls = query_ball_point_multi(radius_list, nsample_list, xyz, new_xyz, counts=False)
        for i, (radius, nsample) in enumerate(zip(radius_list, nsample_list)):
            idx = balls[i][0]
            w_i = mlp_list[i][-1]
            if (points is None or use_xyz) and FUSE_SA and \
                    _sa_chain(xyz, new_xyz, None if points is None else points.contiguous(), idx, mlp_list[i],
                              'conv%d_' % i + '%d', bn, out[:, col:col + w_i], xyz_first=False,
                              shared=shared):                                                      # :176-192 in one kernel
                col += w_i
                continue
            if points is not None and use_xyz:
                grouped = group_concat(xyz, new_xyz, points.contiguous(), idx, xyz_first=False)   # :183
            elif points is not None:
                grouped = group_point(points.contiguous(), idx)
            else:
                grouped = group_concat(xyz, new_xyz, None, idx)
            x2d = grouped.reshape(b * m * nsample, grouped.shape[-1])
            _mlp_stack(x2d, mlp_list[i], 'conv%d_' % i + '%d', bn, pool_rows=nsample,
                       out=out[:, col:col + w_i], coords_in_first=points is None or use_xyz)       # :187-194
            col += w_i
        return new_xyz, out.reshape(b, m, width)


def pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=True):
    """Feature propagation.  Reference: pointnet_util.py:198-228.  xyz1 (B,n,3) dense,
    xyz2 (B,m,3) sparse, points1 (B,n,c1) or None, points2 (B,m,c2) -> (B,n,mlp[-1])."""
    tf_util._inference_only(is_training)
    with variables.variable_scope(scope):
        if FUSE_SA and xyz2.shape[1] == 1 and points1 is not None and len(mlp) >= 1:
            # ONE known point (the level above was a group-all set abstraction): three_nn gives it
            # weight (1/d)/(1/d) = 1 and the two missing neighbours 1/inf = 0 (tf_interpolate.cpp:60-103),
            # so every dense point receives points2[:,0,:] exactly and its share of the first layer is one
            # row per frustum - added as a per-frustum bias, as the v1 seg net does with its global
            # feature; neither the interpolated tensor nor the concatenation (:218) is made
            b, n, c1 = points1.shape
            c2 = points2.shape[2]
            store = variables.get_default_store()
            w, scale, shift = _layer_params(store, variables.scoped('conv_0'), c2 + c1, mlp[0], bn)
            part = ops.shared_mlp(points2[:, 0, :].contiguous(), w[:c2], None, None, relu=False)
            wt = None
            if SA_PRECISION == "bf16" and b * n >= BF16_LAYER_MIN_ROWS:
                wt = _packed_bf16(store, variables.scoped('conv_0') + "#own", w[c2:])
            y = ops.shared_mlp(points1.contiguous().reshape(b * n, c1), w[c2:], scale, shift, relu=True,
                               group_bias=part, rows_per_group=n, wt_bf16=wt)
            if len(mlp) > 1:
                y = _mlp_stack(y, mlp[1:], 'conv_%d', bn, first=1)
            return y.reshape(b, n, y.shape[-1])
        if FUSE_SA and HOIST_FEATURES and points1 is not None and len(mlp) >= 1 and mlp[0] <= points2.shape[2]:
            # interpolation (:210-215) and the first convolution before its batch norm (:222-226) are linear: the
            # KNOWN points go through the interpolated features' rows of W1 once, the products are interpolated and
            # enter the first layer over the dense points' own features as a per-row bias - the concatenation
            # (:218) is not made
            b, n, c1 = points1.shape
            m, c2 = points2.shape[1], points2.shape[2]
            store = variables.get_default_store()
            name0 = variables.scoped('conv_0')
            w, scale, shift = _layer_params(store, name0, c2 + c1, mlp[0], bn)
            bf16 = SA_PRECISION == "bf16" and b * n >= BF16_LAYER_MIN_ROWS
            known = ops.shared_mlp(points2.contiguous().reshape(b * m, c2), w[:c2], None, None, relu=False,
                                   wt_bf16=_packed_bf16(store, name0 + "#known", w[:c2]) if bf16 else None)
            part = three_nn_interpolate(xyz1.contiguous(), xyz2.contiguous(), known.reshape(b, m, mlp[0]))
            y = ops.shared_mlp(points1.contiguous().reshape(b * n, c1), w[c2:], scale, shift, relu=True,
                               group_bias=part.reshape(b * n, mlp[0]), rows_per_group=1,
                               wt_bf16=_packed_bf16(store, name0 + "#own", w[c2:]) if bf16 else None)
            if len(mlp) > 1:
                y = _mlp_stack(y, mlp[1:], 'conv_%d', bn, first=1)
            return y.reshape(b, n, y.shape[-1])
        interpolated = three_nn_interpolate(xyz1.contiguous(), xyz2.contiguous(), points2.contiguous())  # :210-215
        new_points1 = torch.cat([interpolated, points1], dim=2) if points1 is not None else interpolated  # :218
        b, n, c = new_points1.shape
        y = _mlp_stack(new_points1.reshape(b * n, c).contiguous(), mlp, 'conv_%d', bn)                  # :222-226
        return y.reshape(b, n, y.shape[-1])


def _fp_head_plan(store, names, cb, bn):
    """Weights of the fused feature-propagation head in the form fpn_fp_head_chain_*
    reads them; cached on the store until its variables change."""
    cache = store.__dict__.setdefault("_fp_head_cache", {})
    key = (names[0], cb, SA_PRECISION, HOIST_FEATURES, INTERP_BF16)
    hit = cache.get(key)
    if hit is not None and hit["version"] == store.version:
        return hit
    cins = (128 + cb, 128, 128)
    params = [_layer_params(store, names[i], cins[i], 128, bn if i < 2 else True) for i in range(3)]
    w4 = store.get_variable(names[3] + "/weights", (128, 2), "xavier")
    b4 = store.get_variable(names[3] + "/biases", (2,), "zeros")
    version = store.version
    dev = w4.device
    ones = torch.ones(128, dtype=torch.float32, device=dev)
    w1 = torch.zeros((160, 128), dtype=torch.float32, device=dev)
    w1[:128 + cb] = params[0][0]
    ws = [w1, params[1][0].contiguous(), params[2][0].contiguous()]
    scales = [ones if p[1] is None else p[1] for p in params]
    shifts = [p[2] for p in params]
    if SA_PRECISION == "bf16":
        stream = ops.chain_stream_bf16(list(zip(ws, scales)))
        tensors = shifts + [w4.contiguous(), b4]
        plan = {"version": version, "tensors": tensors, "table": ops.pointer_table(tensors), "stream": stream}
        if HOIST_FEATURES and cb <= 8 and INTERP_BF16:
            # the interpolated features' rows of the first layer meet the KNOWN points (fpn_fp_head_chain_pre_bf16)
            w1a = params[0][0][:128].contiguous()
            w1b = params[0][0][128:128 + cb].contiguous() if cb else scales[0]
            pre_tensors = [w1b, scales[0], shifts[0], shifts[1], shifts[2], w4.contiguous(), b4]
            plan["pre"] = {"w1a": w1a, "w1a_bf16": ops.pack_mlp_bf16(w1a), "scale": scales[0], "tensors": pre_tensors,
                           "table": ops.pointer_table(pre_tensors),
                           "stream": ops.chain_stream_bf16(list(zip(ws[1:], scales[1:])))}
    else:
        tensors = []
        for w, sc, sh in zip(ws, scales, shifts):
            tensors += [ops.tile_weights_reg(w, 0), sc, sh]
        tensors += [w4.contiguous(), b4]
        plan = {"version": version, "tensors": tensors, "table": ops.pointer_table(tensors)}
        if HOIST_FEATURES and cb <= 8:
            # the interpolated features' rows of the first layer meet the KNOWN points (fpn_fp_head_chain_pre_f32)
            w1a = params[0][0][:128].contiguous()
            w1b = params[0][0][128:128 + cb].contiguous() if cb else scales[0]
            pre_tensors = [w1b] + tensors[1:]
            plan["pre"] = {"w1a": w1a, "tensors": pre_tensors, "table": ops.pointer_table(pre_tensors)}
    cache[key] = plan
    return plan


def pointnet_fp_module_with_head(xyz1, xyz2, points1, points2, mlp, head_width, num_out, is_training, bn_decay,
                                 scope, head_scope, out_scope, bn=True):
    """Feature propagation (pointnet_fp_module, reference pointnet_util.py:198-228) followed
    by the two 1-wide convolutions that end a PointNet++ segmentation net:
    conv1d(head_width) + batch norm + ReLU under `head_scope`, (dropout: identity at
    inference,) conv1d(num_out) without activation under `out_scope`.  Returns
    (B,n,num_out).  When the shapes fit - mlp == [128,128], head_width == 128, num_out == 2,
    128 interpolated channels, at most 32 own channels - everything after the
    interpolation is ONE kernel and neither the concatenated input nor any hidden layer
    reaches memory; otherwise the layers run one call each."""
    tf_util._inference_only(is_training)
    b, n, _ = xyz1.shape
    cb = 0 if points1 is None else points1.shape[2]
    fused = (FUSE_SA and list(mlp) == [128, 128] and head_width == 128 and num_out == 2 and points2.shape[2] == 128
             and cb <= 32 and (b * n) % 32 == 0)
    if not fused:
        feat = pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=bn)
        net = tf_util.conv1d(feat, head_width, 1, padding='VALID', bn=True, is_training=is_training, scope=head_scope,
                             bn_decay=bn_decay)
        return tf_util.conv1d(net, num_out, 1, padding='VALID', activation_fn=None, scope=out_scope)
    store = variables.get_default_store()
    with variables.variable_scope(scope):
        names = [variables.scoped('conv_0'), variables.scoped('conv_1')]
    names += [variables.scoped(head_scope), variables.scoped(out_scope)]
    plan = _fp_head_plan(store, names, cb, bn)
    rows = b * n
    logits = torch.empty((b, n, 2), dtype=torch.float32, device=xyz1.device)
    xb = None if points1 is None else points1.contiguous()
    lib = _lib.load()
    obs = ops.launch_observer
    if "pre" in plan:
        # interpolation and the first layer (before its batch norm) are both linear: the known points go through
        # their rows of W1 once, the PRODUCTS are interpolated (:210-215 after :222-226 for those rows)
        m = xyz2.shape[1]
        hp = plan["pre"]
        bf16 = "stream" in hp
        known = ops.shared_mlp(points2.contiguous().reshape(b * m, 128), hp["w1a"], hp["scale"] if bf16 else None, None,
                               relu=False, wt_bf16=hp.get("w1a_bf16"))
        pre = three_nn_interpolate(xyz1.contiguous(), xyz2.contiguous(), known.reshape(b, m, 128),
                                   out_dtype=torch.bfloat16 if bf16 else torch.float32)
        tok = obs.begin("fp_head_chain_pre" + ("_bf16" if bf16 else ""), rows, 0, 0,
                        macs=rows * (cb * 128 + 2 * 128 * 128 + 128 * 2)) if obs is not None else None
        with torch.cuda.device(xyz1.device):
            if bf16:
                _lib.check(lib.fpn_fp_head_chain_pre_bf16(rows, cb, _lib.ptr(pre), _lib.ptr(xb), _lib.ptr(hp["stream"]),
                                                          hp["table"], _lib.ptr(logits), _lib.stream_of(xyz1)))
            else:
                _lib.check(lib.fpn_fp_head_chain_pre_f32(rows, cb, _lib.ptr(pre), _lib.ptr(xb), hp["table"],
                                                         _lib.ptr(logits), _lib.stream_of(xyz1)))
        if obs is not None:
            obs.end(tok)
        return logits
    with variables.variable_scope(scope):
        # the bf16 kernel rounds its inputs to bf16 as it reads them: rounded on the way OUT of the
        # interpolation instead, the same values cross memory in half the bytes (INTERP_BF16)
        half = SA_PRECISION == "bf16" and INTERP_BF16
        interpolated = three_nn_interpolate(xyz1.contiguous(), xyz2.contiguous(), points2.contiguous(),
                                            out_dtype=torch.bfloat16 if half else torch.float32)   # :210-215
    tok = obs.begin("fp_head_chain_bf16" if "stream" in plan else "fp_head_chain", rows, 0, 0,
                    macs=rows * ((128 + cb) * 128 + 2 * 128 * 128 + 128 * 2)) if obs is not None else None
    with torch.cuda.device(xyz1.device):
        if "stream" in plan:
            entry = lib.fpn_fp_head_chain_bf16x if interpolated.dtype == torch.bfloat16 else lib.fpn_fp_head_chain_bf16
            _lib.check(entry(rows, 128, cb, _lib.ptr(interpolated), _lib.ptr(xb), _lib.ptr(plan["stream"]),
                             plan["table"], _lib.ptr(logits), _lib.stream_of(xyz1)))
        else:
            _lib.check(lib.fpn_fp_head_chain_f32(rows, 128, cb, _lib.ptr(interpolated), _lib.ptr(xb), plan["table"],
                                                 _lib.ptr(logits), _lib.stream_of(xyz1)))
    if obs is not None:
        obs.end(tok)
    return logits
