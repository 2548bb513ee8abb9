"""Flipout (Bayesian) execution of the v1 topology: the network of
models/FPointNet_Bayes.py:71-223.

Layer kinds (FPointNet_Bayes.py:54-68,93,115,136):
  * every convolution, including the 2-channel segmentation head, is a
    tfp.layers.Convolution2DFlipout(filters, 1, 1, "valid") (+ ReLU except the head);
  * the hidden fully connected layers are tfp.layers.DenseFlipout + ReLU;
  * the two output layers (Dense(59), Dense(3)) are deterministic;
  * no batch norm; Dropout(0.5) before the segmentation head is the identity in
    `predict` (Keras inference mode) and can be switched on as Monte-Carlo dropout.

Flipout arithmetic (the tfp layers are not part of the reference tree; this is the
published estimator, Wen et al. 2018, as tfp implements it):
    y = x . W_mean + b + ((x * s_in) . dW) * s_out,     dW = softplus(rho) * eps
with ONE eps ~ N(0,1) draw per forward call shared by the batch, and independent
Rademacher sign vectors s_in (cin), s_out (cout) per example.  One forward call = one
Monte-Carlo sample; S samples are S calls with fresh noise.

Variables per Flipout layer: '<scope>/weights' (posterior mean, tfp
kernel_posterior_loc), '<scope>/rho' (tfp kernel_posterior_untransformed_scale),
'<scope>/biases' (bias_posterior_loc, deterministic in tfp's default).
"""
import numpy as np
import os

import torch

from . import model_util, ops, plan as _plan

FLIPOUT_EPS = float(np.finfo(np.float32).eps)   # tfp adds eps to softplus(rho)

# scopes of the deterministic output layers
PLAIN = ("fc3-stage1", "fc3")


# the one layer of models/frustum_pointnet.py without batch norm (:225; every other layer,
# conv10 and fc3-stage1 included, is built with the default bn=True, :15,30,150,203)
NO_BN = ("fc3",)


def init_variables(store, pc_channels=3, rho_mean=-3.0, rho_std=0.1, seed=3, bn=False, flipout_outputs=False):
    """Weights of the v1 topology plus a 'rho' per Flipout layer (N(rho_mean, rho_std), tfp's
    default untransformed-scale initialiser).  bn: batch-norm variables for every layer but
    the last (models/frustum_pointnet.py); flipout_outputs: the two output layers are Flipout
    layers too (there: DenseFlipout by default, :30-41,203,225; in FPointNet_Bayes.py they
    are plain Dense, :115,136)."""
    _plan.init_variables(store, pc_channels, bn=False)
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    for scope, cin, cout, _, _ in _plan.layer_table(pc_channels):
        if bn and scope not in NO_BN:
            store.get_variable(scope + "/bn/gamma", (cout,), "ones")
            store.get_variable(scope + "/bn/beta", (cout,), "zeros")
            store.get_variable(scope + "/bn/moving_mean", (cout,), "zeros")
            store.get_variable(scope + "/bn/moving_variance", (cout,), "ones")
        if (scope in PLAIN and not flipout_outputs) or (scope + "/rho") in store:
            continue
        store[scope + "/rho"] = torch.randn((cin, cout), generator=g) * rho_std + rho_mean
    return store


def unpack_keep_bits(bits):
    """(..., 4) int32 keep bits -> (..., 128) float 0/1: bit c of word j -> channel 32j + c."""
    shifts = torch.arange(32, device=bits.device, dtype=torch.int32)
    return ((bits.unsqueeze(-1) >> shifts) & 1).reshape(bits.shape[:-1] + (128,)).float()


class Noise(dict):
    """noise[scope] = (eps (S,cin,cout), sign_in (S,B,cin), sign_out (S,B,cout)) [+ "dropout"].
    When drawn by BayesPlan.sample_noise the per-scope tensors are VIEWS into shared buffers
    (one generator call each): eps_all (S, E), the layers' perturbation draws back to back
    (scope k at column offsets[k]); the sign matrices of all layers, each (S, B, width) in one
    piece, made from one random bit per sign (ops.flipout_signs).  The chain path then scales
    all draws of a sample block by softplus(rho) in one launch instead of one per layer."""
    eps_all = None
    offsets = None
    # The int32 words all signs were spread from and, per scope, where its signs start: sign_pos[scope] = (first
    # input bit, bits per row, first output bit, bits per row).  The chain path builds its per-copy weights from
    # these bits where it can (same values as the float views, a thirtieth of the bytes): whoever edits the float
    # sign tensors of a drawn Noise must set sign_bits to None.
    sign_bits = None
    sign_pos = None


class BayesPlan:
    """S Monte-Carlo forward passes of the Flipout network."""

    def __init__(self, store, pc_channels=3, chunk_frustums=None, mc_dropout=False, chain=True, precision="fp32",
                 bn=False, flipout_outputs=False, recentre_box_input=True):
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (precision,))
        if precision == "bf16" and not chain:
            raise ValueError("bf16 exists for the layer-chain kernels only")
        self.precision = precision
        self.store = store
        self.pc_channels = pc_channels
        self.chunk_frustums = chunk_frustums
        self.mc_dropout = mc_dropout
        # chain: run the convolution stacks through the register-chain kernels on
        # per-frustum effective weights (the points of a frustum share their Flipout
        # signs, so W_mean + diag(s_in) dW diag(s_out) is ONE matrix per frustum and the
        # layer is a plain product with it).  Monte-Carlo dropout (before the 2-channel
        # head) is a keep-bit mask applied inside the tail kernel.
        self.chain = chain
        # bn: inference batch norm after every layer but the last; flipout_outputs: the two output
        # layers draw weights too - together the network of models/frustum_pointnet.py:108-295
        # recentre_box_input=False: the box stack reads the object points as the T-net saw them, not shifted
        # by its centre estimate (models/frustum_pointnet.py:210-214 forms the shifted points and does not use them)
        self.bn = bn
        self.flipout_outputs = flipout_outputs
        self.recentre_box_input = recentre_box_input
        init_variables(store, pc_channels, bn=bn, flipout_outputs=flipout_outputs)
        self.table = _plan.layer_table(pc_channels)
        self.flip_scopes = [scope for scope, _, _, _, _ in self.table if flipout_outputs or scope not in PLAIN]
        self._ones = {}
        self._w6_glob = None
        self._sig_all = None

    # ---- noise -------------------------------------------------------------
    def sample_noise(self, nsamples, batch, generator=None, points=None):
        """All randomness of `nsamples` forward passes over `batch` frustums:
        noise[scope] = (eps (S,cin,cout), sign_in (S,batch,cin), sign_out (S,batch,cout))
        and, with mc_dropout, noise["dropout"] = (S,batch,points,4) int32 keep bits of
        conv9's 128 outputs (bit c of word j keeps channel 32j + c; rate 0.5)."""
        dev = self.store.device
        noise = Noise()
        flip = [(scope, cin, cout) for scope, cin, cout, _, _ in self.table if scope in self.flip_scopes]
        up4 = lambda v: (v + 3) // 4 * 4          # noqa: E731  (every slot starts on a 16-byte boundary)
        total_e = sum(up4(cin * cout) for _, cin, cout in flip)
        # one call per kind of draw: all layers' N(0,1) perturbations; one random bit per sign, spread to +-1 floats
        # by the library (drawn as floats - random integers, times two, minus one - the signs were three launches
        # over 41 M values for 32 samples x 128 frustums).  Each layer's sign matrices lie back to back,
        # (S, batch, width) with the row length rounded up to whole 16-byte groups (what the weight builders read)
        # except for the fully connected layers, whose kernel wants dense matrices: no copies on the way there.
        eps_all = torch.randn((nsamples, total_e), device=dev, generator=generator)
        copies = nsamples * batch
        row = lambda scope, width: width if scope.startswith("fc") else up4(width)      # noqa: E731
        total_s = sum(up4(copies * row(k, cin)) + up4(copies * row(k, cout)) for k, cin, cout in flip)
        bits = torch.randint(-(1 << 31), 1 << 31, ((total_s + 31) // 32,), dtype=torch.int32, device=dev,
                             generator=generator)
        sign_all = ops.flipout_signs(bits, total_s)
        noise.eps_all, noise.offsets = eps_all, {}
        noise.sign_bits, noise.sign_pos = bits, {}
        e = w = 0
        for scope, cin, cout in flip:
            noise.offsets[scope] = e
            ri, ro = row(scope, cin), row(scope, cout)
            noise.sign_pos[scope] = (w, ri, w + up4(copies * ri), ro)
            si = sign_all[w:w + copies * ri].view(nsamples, batch, ri)[:, :, :cin]
            w += up4(copies * ri)
            so = sign_all[w:w + copies * ro].view(nsamples, batch, ro)[:, :, :cout]
            w += up4(copies * ro)
            noise[scope] = (eps_all[:, e:e + cin * cout].unflatten(1, (cin, cout)), si, so)
            e += up4(cin * cout)
        if self.mc_dropout:
            if points is None:
                raise ValueError("sample_noise needs the number of points per frustum for the dropout mask")
            # (all 32 bits of every word are fair: the range is the whole of int32)
            noise["dropout"] = torch.randint(-(1 << 31), 1 << 31, (nsamples, batch, points, 4), dtype=torch.int32,
                                             device=dev, generator=generator)
        return noise

    def _sigma(self, scope):
        return torch.nn.functional.softplus(self.store[scope + "/rho"]) + FLIPOUT_EPS

    def _sigma_all(self):
        """softplus(rho) + eps of every Flipout layer, flattened back to back in table order
        (the column order of Noise.eps_all); cached until the store's variables change."""
        hit = self._sig_all
        if hit is None or hit[0] != self.store.version:
            parts = []
            for scope, cin, cout, _, _ in self.table:
                if scope in self.flip_scopes:
                    parts.append(self.store[scope + "/rho"].reshape(-1))
                    if (cin * cout) % 4:      # the slots of Noise.eps_all start on 16-byte boundaries
                        parts.append(parts[-1].new_zeros(4 - (cin * cout) % 4))
            hit = self._sig_all = (self.store.version, torch.nn.functional.softplus(torch.cat(parts)) + FLIPOUT_EPS)
        return hit[1]

    # ---- layers --------------------------------------------------------------
    def _fold(self, scope):
        """(scale or None, shift) of a layer's epilogue: bias and, with bn, the inference
        batch norm folded (fpn.variables.VariableStore.folded)."""
        return self.store.folded(scope, self.bn and scope not in NO_BN)

    def _scale(self, scope):
        """The epilogue scale as a tensor (ones without batch norm)."""
        scale, shift = self._fold(scope)
        return scale if scale is not None else self._one(shift.shape[0])

    def _flip(self, x, scope, dw, si, so, rows_per_group, relu=True, pool_rows=0, out=None, **kw):
        """One Flipout layer on the rows of x; dw (cin,cout), si (groups,cin), so
        (groups,cout)."""
        scale, shift = self._fold(scope)
        return ops.shared_mlp_flipout(x, self.store[scope + "/weights"], dw.reshape(1, *dw.shape), x.shape[0],
                                      si, so, rows_per_group, scale=scale, shift=shift,
                                      relu=relu, pool_rows=pool_rows, out=out, **kw)

    def _plain(self, x, scope, out=None):
        scale, shift = self._fold(scope)
        return ops.shared_mlp(x, self.store[scope + "/weights"], scale, shift, relu=False, out=out)

    def default_chunk(self, n):
        return max(1, min(1024, (1 << 30) // (n * 3700)))

    def forward(self, point_cloud, one_hot_vec, nsamples=1, noise=None, generator=None):
        """-> dict of tensors with a leading sample axis: mask_logits (S,B,N,2),
        box_output (S,B,59), stage1_center (S,B,3), center (S,B,3), mask (S,B,N), ..."""
        b, n, c = point_cloud.shape
        if n % 32 != 0:
            raise ValueError("the fused max-pool needs the number of points per frustum to be a multiple of 32")
        if noise is None:
            noise = self.sample_noise(nsamples, b, generator, n)
        drop = noise.get("dropout") if self.mc_dropout else None
        if self.mc_dropout and drop is None:
            raise ValueError("mc_dropout needs noise['dropout'] (see sample_noise)")
        shared = noise if isinstance(noise, Noise) and noise.eps_all is not None else None
        noise = {k: v for k, v in noise.items() if k != "dropout"}
        if self.chain and 3 <= c <= 8:
            return self._finish(self._forward_chain(point_cloud, one_hot_vec, nsamples, noise, drop, shared))
        chunk = self.chunk_frustums or self.default_chunk(n)
        per_sample = []
        for s in range(nsamples):
            dws = {k: self._sigma(k) * v[0][s] for k, v in noise.items()}
            outs = []
            for lo in range(0, b, chunk):
                hi = min(b, lo + chunk)
                sl = {k: (dws[k], v[1][s, lo:hi].contiguous(), v[2][s, lo:hi].contiguous())
                      for k, v in noise.items()}
                outs.append(self._forward_chunk(point_cloud[lo:hi], one_hot_vec[lo:hi], sl,
                                                None if drop is None else drop[s, lo:hi]))
            per_sample.append({k: torch.cat([o[k] for o in outs], 0) if len(outs) > 1 else outs[0][k]
                               for k in outs[0]})
        return {k: torch.stack([p[k] for p in per_sample], 0) for k in per_sample[0]}

    # ---- chain path -----------------------------------------------------------
    def _one(self, width):
        t = self._ones.get(width)
        if t is None or t.device != self.store.device:
            t = self._ones[width] = torch.ones(width, dtype=torch.float32, device=self.store.device)
        return t

    # frustum copies (sample x frustum pairs) per pass of the chain path: each carries
    # 2.4 MB of effective weights (1.2 MB as a bf16 stream) and about 0.4 MB of activations,
    # 11 GB / 6 GB a pass.  Measured on 32 samples x 128 frustums x 1024 points, bf16:
    # 256 / 512 / 1024 / 2048 / 4096 copies -> 11.1 / 15.5 / 18.0 / 19.4 / 19.8 k frustums/s
    # (fewer, longer launches; fp32: 3.8 / 3.9 / 3.9 k from 1024 on)
    CHAIN_COPIES = 4096
    SAMPLE_BLOCK = 64

    @staticmethod
    def _finish(r):
        """Derived end points from the per-pass results, once for all samples: (S,B,...)."""
        s_, b_ = r["box_output"].shape[:2]
        r["stage1_center"] = r["center_delta"] + r["mask_xyz_mean"]
        flat = {}
        model_util.parse_output_to_tensors(r["box_output"].reshape(s_ * b_, 59), flat)
        for k, v in flat.items():
            r[k] = v.reshape((s_, b_) + tuple(v.shape[1:]))
        r["center"] = r["center_boxnet"] + r["stage1_center"]
        return r

    def _forward_chain(self, point_cloud, one_hot_vec, nsamples, noise, drop=None, shared=None):
        """All samples of a block of frustums stacked along the frustum axis: a pass of
        the chain kernels works on (samples x frustums) independent "frustum copies",
        each with its own effective weights.  Returns the per-pass results stacked to
        (S,B,...): mask_logits, mask, object_point_indices, mask_xyz_mean, center_delta,
        box_output (the derived end points are formed once at the end, _finish)."""
        b, n, c = point_cloud.shape
        budget = self.chunk_frustums or self.CHAIN_COPIES
        dev = point_cloud.device
        m = model_util.NUM_OBJECT_POINT
        z = lambda *shape, dtype=torch.float32: torch.empty((nsamples, b) + shape, dtype=dtype, device=dev)   # noqa: E731
        result = {"mask_logits": z(n, 2), "mask": z(n), "object_point_indices": z(m, dtype=torch.int32),
                  "mask_xyz_mean": z(3), "center_delta": z(3), "box_output": z(59)}
        if b == 0 or nsamples == 0:      # an empty batch: the end points with zero frustums, nothing to launch
            return result
        sig = None if shared is not None else {k: self._sigma(k) for k in noise}
        # the perturbations sigma * eps of a block of samples are formed once (one launch per
        # block when the draws share a buffer, else one per layer), not per pass; a block holds
        # at most SAMPLE_BLOCK samples x 9 MB
        for q0 in range(0, nsamples, self.SAMPLE_BLOCK):
            q1 = min(nsamples, q0 + self.SAMPLE_BLOCK)
            if shared is not None:
                dw_all = shared.eps_all[q0:q1] * self._sigma_all()
                dws = {k: dw_all[:, shared.offsets[k]:shared.offsets[k] + v[0].shape[1] * v[0].shape[2]]
                       .unflatten(1, tuple(v[0].shape[1:])) for k, v in noise.items()}
            else:
                dws = {k: sig[k] * v[0][q0:q1] for k, v in noise.items()}
            for lo in range(0, b, budget):
                hi = min(b, lo + budget)
                bb = hi - lo
                whole = lo == 0 and hi == b              # (then the slices below are views)
                group = max(1, budget // bb)             # samples per pass
                pcs, ohs = point_cloud[lo:hi].contiguous(), one_hot_vec[lo:hi].contiguous()
                rep = None
                for s0 in range(q0, q1, group):
                    s1 = min(q1, s0 + group)
                    sp = s1 - s0
                    direct = whole or sp == 1
                    nz = {k: (dws[k][s0 - q0:s1 - q0],
                              v[1][s0:s1].reshape(sp * bb, -1) if whole else v[1][s0:s1, lo:hi].reshape(sp * bb, -1),
                              v[2][s0:s1].reshape(sp * bb, -1) if whole else v[2][s0:s1, lo:hi].reshape(sp * bb, -1))
                          for k, v in noise.items()}
                    # the same signs as bits, where the pass's copies are consecutive rows of the sign matrices
                    if shared is not None and shared.sign_bits is not None and whole \
                            and os.environ.get("FPN_FLIPOUT_BITS", "1") != "0":
                        nz["__bits__"] = (shared.sign_bits, s0 * b, shared.sign_pos)
                    if rep is None or rep[0] != sp:      # the repeated inputs are the same for every pass of a block
                        rep = (sp, pcs.repeat(sp, 1, 1) if sp > 1 else pcs, ohs.repeat(sp, 1) if sp > 1 else ohs)
                    dm = None if drop is None else drop[s0:s1, lo:hi].reshape(sp * bb * n, 4).contiguous()
                    # the pass writes straight into the result buffers where its copies are
                    # contiguous there (the whole batch, or one sample of a frustum block)
                    outs = None
                    if direct:
                        outs = {k: (v[s0:s1] if whole else v[s0, lo:hi]).reshape((sp * bb,) + tuple(v.shape[2:]))
                                for k, v in result.items()}
                    ep = self._forward_chunk_chain(rep[1], rep[2], nz, sp, dm, outs)
                    if not direct:
                        for k, v in ep.items():
                            result[k][s0:s1, lo:hi] = v.reshape((sp, bb) + tuple(v.shape[1:]))
        return result

    def _eff(self, scope, nz, layout, rows=None):
        """Per-copy effective weights of a layer in a chain kernel's layout; rows: use
        only the first `rows` input channels (conv6's point_feat part)."""
        dw, si, so = nz[scope]
        w = self.store[scope + "/weights"]
        if rows is not None:
            w, dw, si = w[:rows], dw[:, :rows], si[:, :rows]
            if si.stride(0) % 4:          # (the kernels read the signs in 16-byte groups)
                si = si.contiguous()
        bits = None
        if "__bits__" in nz and layout != 0:
            words, row0, pos = nz["__bits__"]
            in0, ri, out0, ro = pos[scope]
            bits = (words, in0 + row0 * ri, ri, out0 + row0 * ro, ro)
        return ops.flipout_weights(w, dw, si, so, layout, bits=bits)

    def _chain_tensors(self, scopes, nz, layouts):
        """[w_eff, ones, bias] per layer: the `layers` table of a chain kernel."""
        t = []
        for scope, layout in zip(scopes, layouts):
            t += [self._eff(scope, nz, layout), self._scale(scope), self._fold(scope)[1]]
        return t

    def _eff_head(self, nz, copies):
        """conv10's per-copy effective weights (copies, 128, 2) with its batch-norm scale folded into the
        columns (the tail kernels add a shift after the product, they have no scale slot), and that shift."""
        w10 = self._eff("conv10", nz, 0)
        scale, shift = self._fold("conv10")
        if scale is not None:
            w10 = (w10.view(copies, 128, 2) * scale).reshape(copies, 256)
        return w10, shift

    def _forward_chunk_chain(self, pc, one_hot, nz, samples, drop_mask=None, outs=None):
        """pc (samples * frustums, n, c): the frustum block repeated once per sample;
        nz[scope] = (dw (samples,cin,cout), sign_in (copies,>=cin), sign_out (copies,cout)),
        possibly strided views (sample stride of dw, row strides of the signs).  outs: where
        to write the six results (dense (copies, ...) tensors), else they are allocated."""
        b, n, c = pc.shape
        dev = pc.device
        m = model_util.NUM_OBJECT_POINT
        st = self.store
        outs = outs or {}

        def conv(x, scope, rpg, relu=True, out=None):        # fully connected Flipout layers: one row per copy
            dw, si, so = nz[scope]
            scale, shift = self._fold(scope)
            return ops.shared_mlp_flipout(x, st[scope + "/weights"], dw.contiguous(), x.shape[0] // samples,
                                          si.contiguous(), so.contiguous(), rpg,
                                          scale=scale, shift=shift, relu=relu, out=out)

        bf16 = self.precision == "bf16"

        def scaled(scope, w, dw):
            """bf16 streams hold the batch-norm scale inside the weights (one rounding of w * scale): the
            column scale commutes with the output signs, so mean and perturbation are scaled (fp32) before
            the effective weights are formed."""
            scale, _ = self._fold(scope)
            return (w, dw) if scale is None else (w * scale, dw * scale)

        def sign_bits(scope):          # (bits, first input bit, bits a copy, first output bit, bits a copy) or None
            if "__bits__" not in nz:
                return None
            bits, row0, pos = nz["__bits__"]
            in0, ri, out0, ro = pos[scope]
            return (bits, in0 + row0 * ri, ri, out0 + row0 * ro, ro)

        def streams(scopes, first=None):           # bf16: per-copy weight streams of a kernel
            layers = [] if first is None else [first]
            for k in scopes:
                dw, si, so = nz[k]
                layers.append(scaled(k, st[k + "/weights"], dw) + (si, so, sign_bits(k)))
            return ops.flipout_streams_bf16(b, layers)

        def obj_chain(scopes, box, delta, out):
            if bf16:
                ws, cv = streams(scopes[1:])
                ct = [self._eff(scopes[0], nz, 0), self._scale(scopes[0])] + [self._fold(k)[1] for k in scopes]
                ops.v1_obj_chain_bf16(obj2d, m, box, delta, ws, ops.pointer_table(ct), out, copy_vecs=cv)
            else:
                t = self._chain_tensors(scopes, nz, (0,) + (1,) * (len(scopes) - 1))
                ops.v1_obj_chain(obj2d, m, box, delta, ops.pointer_table(t), out, per_frustum=True)

        # segmentation net (FPointNet_Bayes.py:71-96): conv1..conv5 + max-pool, then
        # conv6..conv10 with the frustum-constant part of conv6 as a per-frustum shift
        # (the three [pooled features | one-hot] rows of a copy in ONE zeroed buffer, as in V1Plan._forward_chunk)
        cat = torch.zeros((b, 1028 + 260 + 516), dtype=torch.float32, device=dev)
        gcat, tcat, bcat = cat[:, :1024 + 3], cat[:, 1028:1028 + 256 + 3], cat[:, 1288:1288 + 512 + 3]
        gcat[:, 1024:] = one_hot
        point_feat = torch.empty((b * n, 64), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)
        if bf16:
            ws, cv = streams(("conv2", "conv3", "conv4", "conv5"))
            head = [self._eff("conv1", nz, 0), self._scale("conv1")] + \
                [self._fold(k)[1] for k in ("conv1", "conv2", "conv3", "conv4", "conv5")]
            ops.v1_seg_head_bf16(pc.reshape(b * n, c), n, ws, ops.pointer_table(head), point_feat, gcat, copy_vecs=cv)
            del ws
        else:
            head = self._chain_tensors(("conv1", "conv2", "conv3", "conv4", "conv5"), nz, (0, 1, 1, 1, 1))
            ops.v1_seg_head(pc.reshape(b * n, c), n, ops.pointer_table(head), point_feat, gcat, per_frustum=True)
        dw6, si6, so6 = nz["conv6"]
        w6 = st["conv6/weights"]
        if self._w6_glob is None or self._w6_glob[0] != st.version:
            self._w6_glob = (st.version, w6[64:].contiguous())       # rows of the frustum-constant inputs
        scale6, shift6 = self._fold("conv6")
        gshift = ops.shared_mlp_flipout(gcat, self._w6_glob[1], dw6[:, 64:].contiguous(), b // samples,
                                        si6[:, 64:].contiguous(), so6.contiguous(), 1, scale=scale6,
                                        shift=shift6, relu=False)
        logits = outs.get("mask_logits")
        if logits is None:
            logits = torch.empty((b, n, 2), dtype=torch.float32, device=dev)
        w10, shift10 = self._eff_head(nz, b)
        if bf16:
            w6p, dw6p = scaled("conv6", w6[:64], dw6[:, :64])
            ws, cv = streams(("conv7", "conv8", "conv9"),
                             first=(w6p, dw6p if dw6p.stride(1) == 512 and dw6p.stride(2) == 1 else dw6p.contiguous(),
                                    si6[:, :64] if si6.stride(0) % 4 == 0 else si6[:, :64].contiguous(), so6,
                                    sign_bits("conv6")))
            tail = [self._fold(k)[1] for k in ("conv7", "conv8", "conv9")] + [w10, shift10]
            ops.v1_seg_tail_bf16(point_feat, n, gshift, ws, ops.pointer_table(tail), logits, copy_vecs=cv,
                                 drop_mask=drop_mask)
            del ws
        else:
            tail = [self._eff("conv6", nz, 1, rows=64), self._scale("conv6")] + \
                self._chain_tensors(("conv7", "conv8", "conv9"), nz, (1, 2, 2)) + [w10, shift10]
            ops.v1_seg_tail(point_feat, n, gshift, ops.pointer_table(tail), logits, per_frustum=True,
                            drop_mask=drop_mask)
        del point_feat, head, tail

        mask, mean, sel, obj = ops.mask_gather(pc, logits, m, mask=outs.get("mask"), mean=outs.get("mask_xyz_mean"),
                                               sel=outs.get("object_point_indices"))
        obj2d = obj.reshape(b * m, 3)

        # T-Net (FPointNet_Bayes.py:120-137)
        tcat[:, 256:] = one_hot
        obj_chain(("conv-reg1-stage1", "conv-reg2-stage1", "conv-reg3-stage1"), False, None, tcat)
        t = conv(tcat, "fc1-stage1", 1)
        t = conv(t, "fc2-stage1", 1)
        if self.flipout_outputs:
            center_delta = conv(t, "fc3-stage1", 1, relu=False, out=outs.get("center_delta"))
        else:
            center_delta = self._plain(t, "fc3-stage1", out=outs.get("center_delta"))

        # box net (FPointNet_Bayes.py:99-117); the kernel subtracts center_delta on load
        bcat[:, 512:] = one_hot
        obj_chain(("conv-reg1", "conv-reg2", "conv-reg3", "conv-reg4"), True,
                  center_delta if self.recentre_box_input else None, bcat)
        t = conv(bcat, "fc1", 1)
        t = conv(t, "fc2", 1)
        if self.flipout_outputs:
            output = conv(t, "fc3", 1, relu=False, out=outs.get("box_output"))
        else:
            output = self._plain(t, "fc3", out=outs.get("box_output"))
        return {"mask_logits": logits, "mask": mask, "object_point_indices": sel, "mask_xyz_mean": mean,
                "center_delta": center_delta, "box_output": output}

    def _forward_chunk(self, pc, one_hot, nz, drop=None):
        b, n, c = pc.shape
        dev = pc.device
        m = model_util.NUM_OBJECT_POINT
        x = pc.reshape(b * n, c)

        def conv(x, scope, rpg, **kw):
            dw, si, so = nz[scope]
            return self._flip(x, scope, dw, si, so, rpg, **kw)

        # segmentation net (FPointNet_Bayes.py:71-96)
        a = conv(x, "conv1", n)
        a = conv(a, "conv2", n)
        point_feat = conv(a, "conv3", n)
        a = conv(point_feat, "conv4", n)
        gcat = torch.empty((b, 1024 + 3), dtype=torch.float32, device=dev)
        conv(a, "conv5", n, pool_rows=n, out=gcat[:, :1024])
        gcat[:, 1024:] = one_hot
        del a
        # conv6 on [point_feat | global_feat | one_hot]: the frustum-constant 1027
        # inputs are multiplied once per frustum (one Flipout product on B rows with
        # per-row signs) and enter the per-point product as a per-frustum bias
        dw6, si6, so6 = nz["conv6"]
        w6 = self.store["conv6/weights"]
        gbias = ops.shared_mlp_flipout(gcat, w6[64:].contiguous(), dw6[64:].reshape(1, 1027, 512).contiguous(),
                                       b, si6[:, 64:].contiguous(), so6, 1, scale=None, shift=None, relu=False)
        scale6, shift6 = self._fold("conv6")
        a = ops.shared_mlp_flipout(point_feat, w6[:64].contiguous(), dw6[:64].reshape(1, 64, 512).contiguous(),
                                   b * n, si6[:, :64].contiguous(), so6, n, scale=scale6,
                                   shift=shift6, relu=True, group_bias=gbias)
        del point_feat
        a = conv(a, "conv7", n)
        a = conv(a, "conv8", n)
        a = conv(a, "conv9", n)
        if drop is not None:  # layers.Dropout(rate=0.5), active (FPointNet_Bayes.py:92)
            a = a * unpack_keep_bits(drop).reshape(b * n, 128) * 2.0
        logits = conv(a, "conv10", n, relu=False).reshape(b, n, 2)
        del a

        mask, mean, sel, obj = ops.mask_gather(pc, logits, m)

        # T-Net (FPointNet_Bayes.py:120-137)
        t = conv(obj.reshape(b * m, 3), "conv-reg1-stage1", m)
        t = conv(t, "conv-reg2-stage1", m)
        tcat = torch.empty((b, 256 + 3), dtype=torch.float32, device=dev)
        conv(t, "conv-reg3-stage1", m, pool_rows=m, out=tcat[:, :256])
        tcat[:, 256:] = one_hot
        t = conv(tcat, "fc1-stage1", 1)
        t = conv(t, "fc2-stage1", 1)
        center_delta = conv(t, "fc3-stage1", 1, relu=False) if self.flipout_outputs else self._plain(t, "fc3-stage1")
        stage1_center = center_delta + mean
        obj_new = ((obj - center_delta[:, None, :]) if self.recentre_box_input else obj).reshape(b * m, 3)

        # box net (FPointNet_Bayes.py:99-117)
        t = conv(obj_new, "conv-reg1", m)
        t = conv(t, "conv-reg2", m)
        t = conv(t, "conv-reg3", m)
        bcat = torch.empty((b, 512 + 3), dtype=torch.float32, device=dev)
        conv(t, "conv-reg4", m, pool_rows=m, out=bcat[:, :512])
        bcat[:, 512:] = one_hot
        t = conv(bcat, "fc1", 1)
        t = conv(t, "fc2", 1)
        output = conv(t, "fc3", 1, relu=False) if self.flipout_outputs else self._plain(t, "fc3")

        ep = {"mask_logits": logits, "mask": mask, "object_point_indices": sel, "mask_xyz_mean": mean,
              "center_delta": center_delta, "stage1_center": stage1_center, "box_output": output}
        model_util.parse_output_to_tensors(output, ep)
        ep["center"] = ep["center_boxnet"] + stage1_center
        return ep
