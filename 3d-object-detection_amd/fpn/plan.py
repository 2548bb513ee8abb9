"""Planned execution of the Frustum-PointNet v1 graph.

`V1Plan` runs the same network as models/frustum_pointnets_v1.py:141-188 (and its
Keras / Flipout re-statements) as a fixed sequence of libfpn_hip.so calls:

  * every layer is one fused call (matmul + bias + folded batch norm + ReLU), the
    global max-pools are fused into the layer before them and write straight into
    the concat buffers that the next layer reads, so no pooled / tiled / concatenated
    tensor is ever materialised;
  * conv6 is split: its input is [point_feat(64) | global_feat(1024) | one_hot(3)]
    (frustum_pointnets_v1.py:64-66) and 1027 of those 1091 channels are the same
    for every point of a frustum, so their product with W6[64:1091] is computed
    once per frustum and enters the per-point layer as a per-frustum bias;
  * the batch is processed in chunks of frustums sized so that a chunk's
    activations stay resident in the 256 MiB Infinity Cache between layers.

In Flipout mode (models/FPointNet_Bayes.py) S Monte-Carlo samples of every frustum
are stacked along the row axis and go through the same kernels in one pass.
"""
import torch

from . import model_util, ops

SEG_HEAD = [("conv1", 64), ("conv2", 64), ("conv3", 64), ("conv4", 128), ("conv5", 1024)]
SEG_TAIL = [("conv6", 512), ("conv7", 256), ("conv8", 128), ("conv9", 128)]
TNET_CONV = [("conv-reg1-stage1", 128), ("conv-reg2-stage1", 128), ("conv-reg3-stage1", 256)]
TNET_FC = [("fc1-stage1", 256), ("fc2-stage1", 128)]
BOX_CONV = [("conv-reg1", 128), ("conv-reg2", 128), ("conv-reg3", 256), ("conv-reg4", 512)]
BOX_FC = [("fc1", 512), ("fc2", 256)]
BOX_OUT = 3 + model_util.NUM_HEADING_BIN * 2 + model_util.NUM_SIZE_CLUSTER * 4   # 59


def layer_table(pc_channels):
    """(scope, cin, cout, has_bn_slot, relu) for every layer of the v1 topology."""
    t = []
    cin = pc_channels
    for s, c in SEG_HEAD:
        t.append((s, cin, c, True, True)); cin = c
    cin = 64 + 1024 + 3
    for s, c in SEG_TAIL:
        t.append((s, cin, c, True, True)); cin = c
    t.append(("conv10", cin, 2, False, False))
    cin = 3
    for s, c in TNET_CONV:
        t.append((s, cin, c, True, True)); cin = c
    cin += 3
    for s, c in TNET_FC:
        t.append((s, cin, c, True, True)); cin = c
    t.append(("fc3-stage1", cin, 3, False, False))
    cin = 3
    for s, c in BOX_CONV:
        t.append((s, cin, c, True, True)); cin = c
    cin += 3
    for s, c in BOX_FC:
        t.append((s, cin, c, True, True)); cin = c
    t.append(("fc3", cin, BOX_OUT, False, False))
    return t


def init_variables(store, pc_channels=4, bn=True):
    """Create every variable of the v1 topology in `store` (Xavier weights, zero
    biases, identity batch norm), in the reference's scope names."""
    for scope, cin, cout, bn_slot, _ in layer_table(pc_channels):
        store.get_variable(scope + "/weights", (cin, cout), "xavier")
        store.get_variable(scope + "/biases", (cout,), "zeros")
        if bn and bn_slot:
            store.get_variable(scope + "/bn/gamma", (cout,), "ones")
            store.get_variable(scope + "/bn/beta", (cout,), "zeros")
            store.get_variable(scope + "/bn/moving_mean", (cout,), "zeros")
            store.get_variable(scope + "/bn/moving_variance", (cout,), "ones")
    return store


class V1Plan:
    def __init__(self, store, pc_channels=4, bn=True, chunk_frustums=None, chain=True, precision="fp32"):
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (precision,))
        if precision == "bf16" and not chain:
            raise ValueError("bf16 exists for the layer-chain kernels only")
        self.precision = precision    # operand type of the chain kernels' matrix products
        self.store = store
        self.pc_channels = pc_channels
        self.bn = bn
        self.chunk_frustums = chunk_frustums
        self.chain = chain      # use the layer-chain kernels for the segmentation net
        init_variables(store, pc_channels, bn)
        self._version = None
        self._prepare()

    # folded parameters, re-derived when the store changes
    def _prepare(self):
        st = self.store
        self.p = {}
        for scope, cin, cout, bn_slot, relu in layer_table(self.pc_channels):
            scale, shift = st.folded(scope, self.bn and bn_slot)
            self.p[scope] = (st[scope + "/weights"], scale, shift, relu)
        w6 = st["conv6/weights"]
        self.w6_pt = w6[:64].contiguous()       # rows of point_feat
        self.w6_glob = w6[64:].contiguous()     # rows of [global_feat | one_hot]
        # pointer tables of the chain kernels ({w, scale, shift} per layer; a layer
        # without batch norm gets an explicit scale of ones)
        if not self.chain:
            self._version = st.version
            return

        # The chain kernels keep activations in registers and read their weights in 32x32
        # chunks (fpn_tile_weights_reg_f32; conv8 and conv9 input-tile major, all others
        # output-tile major); the first layer of each stack (K = 3 or 4) takes plain w.
        def reg(scope, jmajor=0, w=None, tiled=True):
            wt, scale, shift, _ = self.p[scope]
            if scale is None:
                scale = torch.ones_like(shift)
            wt = (wt if w is None else w).contiguous()
            return [ops.tile_weights_reg(wt, jmajor) if tiled else wt, scale, shift]
        self._head_tensors = reg("conv1", tiled=False) + sum([reg(s) for s in ("conv2", "conv3", "conv4", "conv5")], [])
        self._tnet_tensors = sum([reg(s, tiled=i > 0) for i, (s, _) in enumerate(TNET_CONV)], [])
        self._box_tensors = sum([reg(s, tiled=i > 0) for i, (s, _) in enumerate(BOX_CONV)], [])
        # conv6's shift reaches the tail kernel folded into the per-frustum term
        self._tail_tensors = reg("conv6", 0, self.w6_pt)[:2] + reg("conv7", 0) + reg("conv8", 1) + reg("conv9", 1) + \
            [self.p["conv10"][0], self.p["conv10"][2]]
        self._head_table = ops.pointer_table(self._head_tensors)
        self._tail_table = ops.pointer_table(self._tail_tensors)
        self._tnet_table = ops.pointer_table(self._tnet_tensors)
        self._box_table = ops.pointer_table(self._box_tensors)
        if self.precision == "bf16":
            # bf16 chains: one weight stream per kernel (batch-norm scale folded into the
            # columns), fp32 first layers and shifts
            P = self.p
            def stream(scopes, first=None):
                layers = [first] if first is not None else []
                return ops.chain_stream_bf16(layers + [(P[s][0].contiguous(), P[s][1]) for s in scopes])
            def consts(first, scopes, extra=()):
                t = (reg(first, tiled=False) if first else []) + [P[s][2] for s in scopes] + list(extra)
                return t, ops.pointer_table(t)
            self._head_ws = stream(("conv2", "conv3", "conv4", "conv5"))
            self._head_ct, self._head_consts = consts("conv1", ("conv2", "conv3", "conv4", "conv5"))
            self._tail_ws = stream(("conv7", "conv8", "conv9"), first=(self.w6_pt, P["conv6"][1]))
            self._tail_ct, self._tail_consts = consts(None, ("conv7", "conv8", "conv9"), (P["conv10"][0], P["conv10"][2]))
            tn = [s for s, _ in TNET_CONV]
            bx = [s for s, _ in BOX_CONV]
            self._tnet_ws = stream(tn[1:])
            self._tnet_ct, self._tnet_consts = consts(tn[0], tn[1:])
            self._box_ws = stream(bx[1:])
            self._box_ct, self._box_consts = consts(bx[0], bx[1:])
        self._version = st.version

    def _layer(self, x, scope, **kw):
        w, scale, shift, relu = self.p[scope]
        return ops.shared_mlp(x, w, scale, shift, relu=relu, **kw)

    def _pooled_layer(self, x, scope, pool_rows, out):
        """Layer + max over groups of pool_rows rows: fused into the layer's epilogue for the
        group sizes the kernel pools (32, 64, multiples of 128), a separate pass otherwise."""
        if pool_rows in (32, 64) or pool_rows % 128 == 0:
            return self._layer(x, scope, pool_rows=pool_rows, out=out)
        return ops.max_pool_rows(self._layer(x, scope), pool_rows, out=out)

    def default_chunk(self, n):
        # largest per-point activation set alive at once is conv6's 512 floats out +
        # conv7's 256: ~3.6 KB per point with the smaller buffers; a chunk of 1 GiB
        # keeps the launch count low (the layers are compute-bound, so spilling the
        # Infinity Cache costs little) while bounding the transient footprint
        if self.chain:
            # the chain kernels keep activations in registers: per point only point_feat
            # (256 B), the logits and the mask reach memory
            return max(1, min(4096, (1 << 30) // (n * 320)))
        per_frustum = n * 3700
        return max(1, min(1024, (1 << 30) // per_frustum))

    def forward(self, point_cloud, one_hot_vec):
        if self._version != self.store.version:
            self._prepare()
        b, n, c = point_cloud.shape
        if c != self.pc_channels:
            raise ValueError("point cloud has %d channels, plan was built for %d" % (c, self.pc_channels))
        if n % 32 != 0:
            raise ValueError("the fused max-pool needs the number of points per frustum to be a multiple of 32")
        chunk = self.chunk_frustums or self.default_chunk(n)
        outs = []
        for s in range(0, max(b, 1), chunk):      # (an empty batch still makes one, empty, pass)
            outs.append(self._forward_chunk(point_cloud[s:s + chunk], one_hot_vec[s:s + chunk]))
        ep = {}
        for k in outs[0]:
            ep[k] = outs[0][k] if len(outs) == 1 else torch.cat([o[k] for o in outs], dim=0)
        return ep

    def _forward_chunk(self, pc, one_hot):
        b, n, c = pc.shape
        dev = pc.device
        rows = b * n
        m = model_util.NUM_OBJECT_POINT
        x = pc.reshape(rows, c)

        # --- instance segmentation net (frustum_pointnets_v1.py:20-90) -------------
        bf16 = self.precision == "bf16"
        if self.chain:
            # the three [pooled features | one-hot] rows of a frustum lie in ONE zeroed buffer (the chain kernels
            # accumulate their maxima in place): one fill instead of three, and every part starts on a 16-byte
            # boundary with a row stride of whole 16-byte groups, which the dense layers read as vectors
            cat = torch.zeros((b, 1028 + 260 + 516), dtype=torch.float32, device=dev)
            gcat, tcat, bcat = cat[:, :1024 + 3], cat[:, 1028:1028 + 256 + 3], cat[:, 1288:1288 + 512 + 3]
        else:
            gcat = torch.empty((b, 1024 + 3), dtype=torch.float32, device=dev)
        gcat[:, 1024:] = one_hot                                        # concat one-hot :64
        if self.chain:
            point_feat = torch.empty((rows, 64), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)
            if bf16:
                ops.v1_seg_head_bf16(x, n, self._head_ws, self._head_consts, point_feat, gcat)
            else:
                ops.v1_seg_head(x, n, self._head_table, point_feat, gcat)  # conv1..conv5 + maxpool :41-61
            # frustum-constant part of conv6, folded into its shift:
            # ([global_feat | one_hot] . W6[64:]) * scale6 + shift6
            _, scale6, shift6, _ = self.p["conv6"]
            gbias = ops.shared_mlp(gcat, self.w6_glob, scale6, shift6, relu=False)
            logits = torch.empty((b, n, 2), dtype=torch.float32, device=dev)
            if bf16:
                ops.v1_seg_tail_bf16(point_feat, n, gbias, self._tail_ws, self._tail_consts, logits)
            else:
                ops.v1_seg_tail(point_feat, n, gbias, self._tail_table, logits)   # conv6..conv10 :68-89
            del point_feat
        else:
            a = self._layer(x, "conv1")
            a = self._layer(a, "conv2")
            point_feat = self._layer(a, "conv3")
            a = self._layer(point_feat, "conv4")
            self._pooled_layer(a, "conv5", n, gcat[:, :1024])             # conv5 + maxpool :57-62
            del a
            gbias = ops.shared_mlp(gcat, self.w6_glob, None, None, relu=False)
            _, scale6, shift6, _ = self.p["conv6"]
            a = ops.shared_mlp(point_feat, self.w6_pt, scale6, shift6, relu=True,
                               group_bias=gbias, rows_per_group=n)          # conv6 :68-71
            del point_feat
            a = self._layer(a, "conv7")
            a = self._layer(a, "conv8")
            a = self._layer(a, "conv9")
            logits = self._layer(a, "conv10").reshape(b, n, 2)             # dropout = identity :84-89
            del a

        # --- masking (model_util.py:275-319) -----------------------------------------
        mask, mean, sel, obj = ops.mask_gather(pc, logits, m)

        # --- T-Net (model_util.py:322-357) ---------------------------------------------
        if not self.chain:
            tcat = torch.empty((b, 256 + 3), dtype=torch.float32, device=dev)
        tcat[:, 256:] = one_hot
        obj2d = obj.reshape(b * m, 3)
        if self.chain:
            if bf16:
                ops.v1_obj_chain_bf16(obj2d, m, False, None, self._tnet_ws, self._tnet_consts, tcat)
            else:
                ops.v1_obj_chain(obj2d, m, False, None, self._tnet_table, tcat)   # conv stack + maxpool :335-347
        else:
            t = self._layer(obj2d, "conv-reg1-stage1")
            t = self._layer(t, "conv-reg2-stage1")
            self._pooled_layer(t, "conv-reg3-stage1", m, tcat[:, :256])
        t = self._layer(tcat, "fc1-stage1")
        t = self._layer(t, "fc2-stage1")
        center_delta = self._layer(t, "fc3-stage1")
        stage1_center = center_delta + mean                              # frustum_pointnets_v1.py:173

        # --- box estimation net (frustum_pointnets_v1.py:93-138) ------------------------
        if not self.chain:
            bcat = torch.empty((b, 512 + 3), dtype=torch.float32, device=dev)
        bcat[:, 512:] = one_hot
        if self.chain:
            # the kernel subtracts center_delta while loading the tile (:176-177)
            if bf16:
                ops.v1_obj_chain_bf16(obj2d, m, True, center_delta, self._box_ws, self._box_consts, bcat)
            else:
                ops.v1_obj_chain(obj2d, m, True, center_delta, self._box_table, bcat)   # :108-124
        else:
            obj_new = (obj - center_delta[:, None, :]).reshape(b * m, 3)     # :176-177
            t = self._layer(obj_new, "conv-reg1")
            t = self._layer(t, "conv-reg2")
            t = self._layer(t, "conv-reg3")
            self._pooled_layer(t, "conv-reg4", m, bcat[:, :512])
        t = self._layer(bcat, "fc1")
        t = self._layer(t, "fc2")
        output = self._layer(t, "fc3")

        ep = {"mask_logits": logits, "mask": mask, "object_point_indices": sel, "mask_xyz_mean": mean,
              "center_delta": center_delta, "stage1_center": stage1_center, "box_output": output}
        model_util.parse_output_to_tensors(output, ep)
        ep["center"] = ep["center_boxnet"] + stage1_center               # :186
        return ep
