"""Thin tensor-level callers of the shared-MLP and masking entry points."""
import ctypes

import torch

from . import _lib


# Optional observer of shared-MLP launches (the benchmark's per-kernel timer):
# an object with begin(kernel_name, rows, cin, cout) -> token and end(token).
launch_observer = None


def mlp_kernel_name(cout, rows=None, cin=None, pool_rows=0):
    """Which configuration fpn_shared_mlp_f32 launches: the split-K kernel for few rows
    (at most 1024 tiles of 32 x 64, no fused pooling), else 128-row tiles by layer width."""
    if rows is not None and cin is not None and not pool_rows and cin >= 32 and \
            -(-rows // 32) * -(-cout // 64) <= 1024:
        return "shared_mlp_fewrow"
    return "shared_mlp_128x128" if cout > 64 else ("shared_mlp_128x64" if cout > 32 else "shared_mlp_128x32")


def _check_layer_operands(x, w, cin, cout, scale, shift, group_bias, groups, extra=()):
    """A raw-pointer ABI reads whatever it is given: weights must be a dense float32
    (cin,cout) matrix on x's device, the per-channel constants dense (cout,) vectors, a
    per-group bias a dense (groups,cout) table; `extra`: (name, tensor, shape) triples."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError("x must be a float32 tensor on a ROCm device (got %s on %s)" % (x.dtype, x.device))
    _lib.require_device_tensor("w", w, torch.float32, 2)
    if tuple(w.shape) != (cin, cout):
        raise ValueError("w has shape %s, expected (%d, %d)" % (tuple(w.shape), cin, cout))
    named = [("scale", scale, (cout,)), ("shift", shift, (cout,))]
    if group_bias is not None:
        named.append(("group_bias", group_bias, (groups, cout)))
    named += list(extra)
    for name, t, shape in named:
        if t is None:
            continue
        _lib.require_device_tensor(name, t, torch.float32, len(shape))
        if tuple(t.shape) != tuple(shape):
            raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    _lib.same_device(x, w, scale, shift, group_bias, *[t for _, t, _ in extra])


def pack_mlp_bf16(w):
    """w (cin,cout) float32 -> the layer's weights as fpn_shared_mlp_bf16 reads them:
    (cout, cin rounded up to 64) bf16, transposed, zero beyond cin (fpn_mlp_pack_bf16)."""
    _lib.require_device_tensor("w", w, torch.float32, 2)
    if not w.is_contiguous():
        raise ValueError("w must be a dense (cin,cout) matrix")
    cin, cout = w.shape
    wt = torch.empty((cout, (cin + 63) // 64 * 64), dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().fpn_mlp_pack_bf16(cin, cout, _lib.ptr(w), _lib.ptr(wt), _lib.stream_of(w)))
    return wt


def shared_mlp(x, w, scale=None, shift=None, relu=True, pool_rows=0, group_bias=None,
               rows_per_group=0, out=None, cin=None, wt_bf16=None):
    """One fused layer through fpn_shared_mlp_f32 - or, with wt_bf16 = pack_mlp_bf16(w),
    through fpn_shared_mlp_bf16 (bf16 operands, fp32 accumulation and epilogue; for layers
    whose inputs are features, not coordinates).

    x (rows, ldx) float32 with the layer's inputs in columns [0,cin) (cin defaults
    to w.shape[0]); w (cin,cout); scale/shift (cout,) or None; group_bias
    (rows/rows_per_group, cout) or None.  pool_rows > 0 max-pools groups of
    pool_rows consecutive rows.  `out` may be a (rows_out, ldy) view into a wider
    buffer (stride(0) = ldy >= cout, unit column stride).
    """
    lib = _lib.load()
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D tensor with unit column stride")
    rows, ldx = x.shape[0], x.stride(0) if x.shape[0] > 1 else max(x.stride(0), x.shape[1])
    cin = w.shape[0] if cin is None else cin
    cout = w.shape[1]
    if x.shape[1] < cin or w.shape[0] != cin:
        raise ValueError("x has %d columns, w is %s" % (x.shape[1], tuple(w.shape)))
    if group_bias is not None and rows_per_group < 1:
        raise ValueError("group_bias needs rows_per_group >= 1")
    _check_layer_operands(x, w, cin, cout, scale, shift, group_bias,
                          -(-rows // rows_per_group) if group_bias is not None else 0)
    rows_out = rows // pool_rows if pool_rows else rows
    if out is None:
        out = torch.empty((rows_out, cout), dtype=torch.float32, device=x.device)
    if out.shape[0] != rows_out or out.shape[1] < cout or out.stride(1) != 1 or out.dtype != torch.float32 \
            or out.device != x.device:
        raise ValueError("out has shape %s, expected float32 (%d, >=%d) on %s" % (tuple(out.shape), rows_out, cout, x.device))
    ldy = out.stride(0) if out.shape[0] > 1 else max(out.stride(0), out.shape[1])
    ws = None
    ws_bytes = lib.fpn_shared_mlp_workspace_bytes(rows, cout, pool_rows)
    if ws_bytes:
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    obs = launch_observer
    if wt_bf16 is not None:
        if wt_bf16.dtype != torch.bfloat16 or tuple(wt_bf16.shape) != (cout, (cin + 63) // 64 * 64) \
                or not wt_bf16.is_contiguous() or wt_bf16.device != x.device:
            raise ValueError("wt_bf16 must be pack_mlp_bf16(w): a dense bfloat16 (%d, %d) tensor on %s"
                             % (cout, (cin + 63) // 64 * 64, x.device))
        tok = obs.begin("shared_mlp_bf16", rows, cin, cout) if obs is not None else None
        with torch.cuda.device(x.device):
            _lib.check(lib.fpn_shared_mlp_bf16(rows, cin, cout, _lib.ptr(x), ldx, _lib.ptr(wt_bf16), _lib.ptr(scale),
                                               _lib.ptr(shift), _lib.ptr(group_bias), int(rows_per_group),
                                               int(bool(relu)), int(pool_rows), _lib.ptr(out), ldy,
                                               _lib.ptr(ws), ws_bytes, _lib.stream_of(x)))
        if obs is not None:
            obs.end(tok)
        return out
    tok = obs.begin(mlp_kernel_name(cout, rows, cin, pool_rows), rows, cin, cout) if obs is not None else None
    with torch.cuda.device(x.device):
        _lib.check(lib.fpn_shared_mlp_f32(rows, cin, cout, _lib.ptr(x), ldx, _lib.ptr(w), _lib.ptr(scale),
                                          _lib.ptr(shift), _lib.ptr(group_bias), int(rows_per_group),
                                          int(bool(relu)), int(pool_rows), _lib.ptr(out), ldy,
                                          _lib.ptr(ws), ws_bytes, _lib.stream_of(x)))
    if obs is not None:
        obs.end(tok)
    return out


def shared_mlp_flipout(x, w_mean, w_delta, rows_per_sample, sign_in, sign_out, rows_per_group,
                       scale=None, shift=None, relu=True, pool_rows=0, out=None, group_bias=None):
    """Flipout layer through fpn_shared_mlp_flipout_f32: x (S*rows_per_sample, cin),
    w_mean (cin,cout), w_delta (S,cin,cout), sign_in (groups,cin), sign_out
    (groups,cout) with groups = rows / rows_per_group."""
    lib = _lib.load()
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D tensor with unit column stride")
    rows, ldx = x.shape[0], x.stride(0) if x.shape[0] > 1 else max(x.stride(0), x.shape[1])
    cin, cout = w_mean.shape
    if x.shape[1] < cin:
        raise ValueError("x has %d columns, w_mean is %s" % (x.shape[1], tuple(w_mean.shape)))
    if rows_per_group < 1 or rows_per_sample < 1 or rows % rows_per_sample or rows_per_sample % rows_per_group:
        raise ValueError("rows (%d) must split into samples of %d rows and those into groups of %d"
                         % (rows, rows_per_sample, rows_per_group))
    groups = rows // rows_per_group
    _check_layer_operands(x, w_mean, cin, cout, scale, shift, group_bias, groups,
                          extra=[("w_delta", w_delta, (rows // rows_per_sample, cin, cout)),
                                 ("sign_in", sign_in, (groups, cin)), ("sign_out", sign_out, (groups, cout))])
    rows_out = rows // pool_rows if pool_rows else rows
    if out is None:
        out = torch.empty((rows_out, cout), dtype=torch.float32, device=x.device)
    ldy = out.stride(0) if out.shape[0] > 1 else max(out.stride(0), out.shape[1])
    ws = None
    ws_bytes = lib.fpn_shared_mlp_workspace_bytes(rows, cout, pool_rows)
    if ws_bytes:
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.fpn_shared_mlp_flipout_f32(
            rows, cin, cout, _lib.ptr(x), ldx, _lib.ptr(w_mean), _lib.ptr(w_delta), int(rows_per_sample),
            _lib.ptr(sign_in), _lib.ptr(sign_out), int(rows_per_group), _lib.ptr(group_bias),
            _lib.ptr(scale), _lib.ptr(shift),
            int(bool(relu)), int(pool_rows), _lib.ptr(out), ldy, _lib.ptr(ws), ws_bytes,
            _lib.stream_of(x)))
    return out


def max_pool_rows(x, pool_rows, out=None):
    """x (rows, C) -> (rows/pool_rows, C): max over groups of consecutive rows,
    through fpn_max_pool_rows."""
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError("x must be a 2-D float32 tensor with unit column stride on a ROCm device")
    rows, c = x.shape
    if pool_rows < 1 or rows % pool_rows:
        raise ValueError("rows (%d) must be a multiple of pool_rows (%d)" % (rows, pool_rows))
    ldx = x.stride(0) if rows > 1 else max(x.stride(0), c)
    if out is None:
        out = torch.empty((rows // pool_rows, c), dtype=torch.float32, device=x.device)
    ldy = out.stride(0) if out.shape[0] > 1 else max(out.stride(0), out.shape[1])
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().fpn_max_pool_rows(rows, c, int(pool_rows), _lib.ptr(x), ldx, _lib.ptr(out),
                                                 ldy, _lib.stream_of(x)))
    return out


def pointer_table(tensors):
    """Host array of device pointers (the `layers` argument of the chain kernels).
    The caller keeps `tensors` alive for as long as the table is used."""
    import ctypes
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def tile_weights_reg(w, jmajor):
    """w (cin,cout) row-major -> the register-chain kernels' chunked layout (a flat
    tensor of cin*cout floats) through fpn_tile_weights_reg_f32."""
    _lib.require_device_tensor("w", w, torch.float32, 2)
    cin, cout = w.shape
    out = torch.empty(cin * cout, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().fpn_tile_weights_reg_f32(cin, cout, int(bool(jmajor)), _lib.ptr(w), _lib.ptr(out),
                                                        _lib.stream_of(w)))
    return out


def _ld(t):
    """Row stride of a 2-D tensor whose rows are dense; a single row has none that matters
    (torch may report the stride of the wider tensor it was sliced from): its width."""
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def flipout_signs(bits, n, out=None):
    """+-1 float signs from random bits through fpn_flipout_signs: out[i] = +1 where bit i % 32 of the int32 word
    bits[i // 32] is set, else -1; bits a contiguous int32 tensor of at least (n + 31) // 32 words."""
    if bits.dtype != torch.int32 or not bits.is_contiguous() or bits.numel() * 32 < n:
        raise ValueError("bits must be a contiguous int32 tensor of at least %d words" % ((n + 31) // 32))
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=bits.device)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != n or out.device != bits.device:
        raise ValueError("out must be a contiguous float32 tensor of %d elements on the device of bits" % n)
    with torch.cuda.device(bits.device):
        _lib.check(_lib.load().fpn_flipout_signs(int(n), _lib.ptr(bits), _lib.ptr(out), _lib.stream_of(bits)))
    return out


def flipout_weights(w_mean, w_delta, sign_in, sign_out, layout, out=None, bits=None):
    """Effective Flipout weights per frustum through fpn_flipout_weights_f32:
    W_mean + diag(sign_in[f]) . W_delta[f // per] . diag(sign_out[f]) for every row f of
    the sign matrices, in `layout` (0 row-major, 1 / 2 the chain kernels' chunk layout,
    output-tile / input-tile major).  w_delta (S, cin, cout): S Monte-Carlo samples whose
    frustum copies are stacked along the rows of the sign matrices (per = rows / S); it
    may be a row slice [:, :cin] of a taller stack and sign_in a column slice of a wider
    matrix.  bits (chunked layouts): (int32 tensor, in_bit0, ld_in, out_bit0, ld_out) - the same
    signs as one bit each (fpn_flipout_weights_bits_f32; same matrices).  Returns (rows, cin*cout)."""
    cin, cout = w_mean.shape
    b = sign_in.shape[0]
    s = w_delta.shape[0]
    if w_delta.dim() != 3 or b % s != 0 or w_delta.stride(2) != 1 or w_delta.stride(1) != cout:
        raise ValueError("w_delta must be (samples, cin, cout) with dense matrices and samples dividing %d" % b)
    if out is None:
        out = torch.empty((b, cin * cout), dtype=torch.float32, device=w_mean.device)
    if bits is not None and layout in (1, 2):
        words, in0, ldi, out0, ldo = bits
        if words.dtype != torch.int32 or not words.is_contiguous():
            raise ValueError("sign bits must be a contiguous int32 tensor")
        if max(in0 + (b - 1) * ldi + cin, out0 + (b - 1) * ldo + cout) > 32 * words.numel():
            raise ValueError("sign bits: the last copy's signs lie beyond the tensor")
        with torch.cuda.device(w_mean.device):
            _lib.check(_lib.load().fpn_flipout_weights_bits_f32(b, cin, cout, _lib.ptr(w_mean), _lib.ptr(w_delta), b // s,
                                                                w_delta.stride(0) if s > 1 else 0, _lib.ptr(words), int(in0),
                                                                int(ldi), int(out0), int(ldo), int(layout), _lib.ptr(out),
                                                                _lib.stream_of(w_mean)))
        return out
    with torch.cuda.device(w_mean.device):
        _lib.check(_lib.load().fpn_flipout_weights_f32(b, cin, cout, _lib.ptr(w_mean), _lib.ptr(w_delta), b // s,
                                                       w_delta.stride(0) if s > 1 else 0, _lib.ptr(sign_in),
                                                       _ld(sign_in), _lib.ptr(sign_out), _ld(sign_out), int(layout),
                                                       _lib.ptr(out), _lib.stream_of(w_mean)))
    return out


def v1_seg_head(pc2d, n, table, point_feat, global_max, per_frustum=False):
    """conv1..conv5 + max over each frustum's points through fpn_v1_seg_head_f32;
    global_max (frustums, >=1024) must hold zeros (or a running maximum) in its first
    1024 columns and may be a view into a wider buffer."""
    rows, cpc = pc2d.shape
    obs = launch_observer
    tok = obs.begin("v1_seg_head", rows, 0, 0, macs=rows * (cpc * 64 + 64 * 64 * 2 + 64 * 128 + 128 * 1024)) \
        if obs is not None else None
    with torch.cuda.device(pc2d.device):
        _lib.check(_lib.load().fpn_v1_seg_head_f32(rows, int(n), cpc, _lib.ptr(pc2d), table, int(bool(per_frustum)),
                                                   _lib.ptr(point_feat), _lib.ptr(global_max), _ld(global_max),
                                                   _lib.stream_of(pc2d)))
    if obs is not None:
        obs.end(tok)


def v1_seg_tail(point_feat, n, gbias, table, logits, per_frustum=False, drop_mask=None):
    """conv6..conv10 through fpn_v1_seg_tail_f32; gbias (frustums,512) is conv6's
    shift with the frustum-constant part of the product folded in; drop_mask (rows,4)
    int32 keep bits of conv9's outputs (Monte-Carlo dropout) or None."""
    rows = point_feat.shape[0]
    if drop_mask is not None and (drop_mask.dtype != torch.int32 or tuple(drop_mask.shape) != (rows, 4)
                                  or not drop_mask.is_contiguous()):
        raise ValueError("drop_mask must be a contiguous (rows,4) int32 tensor")
    obs = launch_observer
    tok = obs.begin("v1_seg_tail", rows, 0, 0, macs=rows * (64 * 512 + 512 * 256 + 256 * 128 + 128 * 128 + 128 * 2)) \
        if obs is not None else None
    with torch.cuda.device(point_feat.device):
        _lib.check(_lib.load().fpn_v1_seg_tail_f32(rows, int(n), _lib.ptr(point_feat), _lib.ptr(gbias), table,
                                                   int(bool(per_frustum)), _lib.ptr(drop_mask), _lib.ptr(logits),
                                                   _lib.stream_of(point_feat)))
    if obs is not None:
        obs.end(tok)


def v1_obj_chain(obj2d, m, box, delta, table, feat_max, per_frustum=False):
    """T-Net (box=False) or box-net (box=True) convolution stack + max over the object
    points through fpn_v1_obj_chain_f32; `delta` (frustums,3) or None; feat_max
    (frustums, >=256 / >=512) zeroed (or a running maximum), possibly a view."""
    rows = obj2d.shape[0]
    macs = 3 * 128 + 128 * 128 + 128 * 256 + (256 * 512 if box else 0)
    obs = launch_observer
    tok = obs.begin("v1_obj_chain_box" if box else "v1_obj_chain_tnet", rows, 0, 0, macs=rows * macs) \
        if obs is not None else None
    with torch.cuda.device(obj2d.device):
        _lib.check(_lib.load().fpn_v1_obj_chain_f32(rows, int(m), int(bool(box)), _lib.ptr(obj2d), _lib.ptr(delta),
                                                    table, int(bool(per_frustum)), _lib.ptr(feat_max), _ld(feat_max),
                                                    _lib.stream_of(obj2d)))
    if obs is not None:
        obs.end(tok)


def chain_stream_bf16(layers):
    """The weight stream of a bf16 chain kernel: `layers` is the list of (w (cin,cout)
    fp32, folded batch-norm scale (cout,) or None) in the order the kernel runs them;
    returns one bfloat16 tensor (fpn_chain_pack_bf16 per layer, back to back, plus the
    read-ahead padding)."""
    lib = _lib.load()
    dev = layers[0][0].device
    pad = lib.fpn_chain_pad_units_bf16() * 2048
    out = torch.zeros(sum(w.shape[0] * w.shape[1] for w, _ in layers) + pad, dtype=torch.bfloat16, device=dev)
    off = 0
    with torch.cuda.device(dev):
        for w, scale in layers:
            _lib.require_device_tensor("w", w, torch.float32, 2)
            cin, cout = w.shape
            _lib.check(lib.fpn_chain_pack_bf16(cin, cout, _lib.ptr(w), _lib.ptr(scale),
                                               ctypes.c_void_p(out.data_ptr() + 2 * off), _lib.stream_of(w)))
            off += cin * cout
    return out


def flipout_streams_bf16(copies, layers):
    """Per-copy weight streams of a bf16 chain kernel under Flipout: `layers` is the
    list of (w_mean (cin,cout), w_delta (S,cin,cout), sign_in (copies,>=cin), sign_out
    (copies,cout)) in the order the kernel runs them (fpn_flipout_stream_bf16 per layer).
    An entry may carry a fifth item (bits int32 tensor, in_bit0, ld_in, out_bit0, ld_out): the
    same signs as one bit each (fpn_flipout_stream_bits_bf16 - same units, read from a few
    megabytes instead of 4 bytes a sign).  Returns (stream tensor, copy_vecs)."""
    lib = _lib.load()
    dev = layers[0][0].device
    per = sum(l[0].shape[0] * l[0].shape[1] for l in layers)
    pad = lib.fpn_chain_pad_units_bf16() * 2048
    out = torch.empty(copies * per + pad, dtype=torch.bfloat16, device=dev)
    out[copies * per:].zero_()
    off = 0
    with torch.cuda.device(dev):
        for layer in layers:
            w, dw, si, so = layer[:4]
            cin, cout = w.shape
            s = dw.shape[0]
            if dw.dim() != 3 or copies % s != 0 or dw.stride(2) != 1 or dw.stride(1) != cout:
                raise ValueError("w_delta must be (samples, cin, cout) with dense matrices and samples dividing %d" % copies)
            if len(layer) > 4 and layer[4] is not None:
                bits, in0, ldi, out0, ldo = layer[4]
                if bits.dtype != torch.int32 or not bits.is_contiguous():
                    raise ValueError("sign bits must be a contiguous int32 tensor")
                if max(in0 + (copies - 1) * ldi + cin, out0 + (copies - 1) * ldo + cout) > 32 * bits.numel():
                    raise ValueError("sign bits: the last copy's signs lie beyond the tensor")
                _lib.check(lib.fpn_flipout_stream_bits_bf16(copies, cin, cout, _lib.ptr(w), _lib.ptr(dw), copies // s,
                                                            dw.stride(0) if s > 1 else 0, _lib.ptr(bits), int(in0), int(ldi),
                                                            int(out0), int(ldo), ctypes.c_void_p(out.data_ptr() + 2 * off),
                                                            per // 8, _lib.stream_of(w)))
                off += cin * cout
                continue
            _lib.check(lib.fpn_flipout_stream_bf16(copies, cin, cout, _lib.ptr(w), _lib.ptr(dw), copies // s,
                                                   dw.stride(0) if s > 1 else 0, _lib.ptr(si), _ld(si), _lib.ptr(so),
                                                   _ld(so), ctypes.c_void_p(out.data_ptr() + 2 * off), per // 8,
                                                   _lib.stream_of(w)))
            off += cin * cout
    return out, per // 8


def v1_seg_head_bf16(pc2d, n, wstream, consts, point_feat, global_max, copy_vecs=0):
    """fpn_v1_seg_head_bf16: as v1_seg_head; point_feat (rows,64) bfloat16; copy_vecs > 0:
    one weight stream per frustum (flipout_streams_bf16)."""
    rows, cpc = pc2d.shape
    obs = launch_observer
    tok = obs.begin("v1_seg_head_bf16", rows, 0, 0, macs=rows * (cpc * 64 + 64 * 64 * 2 + 64 * 128 + 128 * 1024)) \
        if obs is not None else None
    with torch.cuda.device(pc2d.device):
        _lib.check(_lib.load().fpn_v1_seg_head_bf16(rows, int(n), cpc, _lib.ptr(pc2d), _lib.ptr(wstream), consts,
                                                    int(copy_vecs), _lib.ptr(point_feat), _lib.ptr(global_max),
                                                    _ld(global_max),
                                                    _lib.stream_of(pc2d)))
    if obs is not None:
        obs.end(tok)


def v1_seg_tail_bf16(point_feat, n, gbias, wstream, consts, logits, copy_vecs=0, drop_mask=None):
    """fpn_v1_seg_tail_bf16: as v1_seg_tail on the bfloat16 point_feat of the bf16 head."""
    rows = point_feat.shape[0]
    if drop_mask is not None and (drop_mask.dtype != torch.int32 or tuple(drop_mask.shape) != (rows, 4)
                                  or not drop_mask.is_contiguous()):
        raise ValueError("drop_mask must be a contiguous (rows,4) int32 tensor")
    obs = launch_observer
    tok = obs.begin("v1_seg_tail_bf16", rows, 0, 0, macs=rows * (64 * 512 + 512 * 256 + 256 * 128 + 128 * 128 + 128 * 2)) \
        if obs is not None else None
    with torch.cuda.device(point_feat.device):
        _lib.check(_lib.load().fpn_v1_seg_tail_bf16(rows, int(n), _lib.ptr(point_feat), _lib.ptr(gbias),
                                                    _lib.ptr(wstream), consts, int(copy_vecs), _lib.ptr(drop_mask),
                                                    _lib.ptr(logits), _lib.stream_of(point_feat)))
    if obs is not None:
        obs.end(tok)


def v1_obj_chain_bf16(obj2d, m, box, delta, wstream, consts, feat_max, copy_vecs=0):
    """fpn_v1_obj_chain_bf16: as v1_obj_chain."""
    rows = obj2d.shape[0]
    macs = 3 * 128 + 128 * 128 + 128 * 256 + (256 * 512 if box else 0)
    obs = launch_observer
    tok = obs.begin("v1_obj_chain_box_bf16" if box else "v1_obj_chain_tnet_bf16", rows, 0, 0, macs=rows * macs) \
        if obs is not None else None
    with torch.cuda.device(obj2d.device):
        _lib.check(_lib.load().fpn_v1_obj_chain_bf16(rows, int(m), int(bool(box)), _lib.ptr(obj2d), _lib.ptr(delta),
                                                     _lib.ptr(wstream), consts, int(copy_vecs), _lib.ptr(feat_max),
                                                     _ld(feat_max),
                                                     _lib.stream_of(obj2d)))
    if obs is not None:
        obs.end(tok)


# How the npick object points are chosen when a frustum has MORE positives than that (the
# reference draws a random subset, models/model_util.py:73-75): "first" = the first npick in
# index order, "spread" = evenly spaced over the positives (unbiased for inputs whose point
# order is not random).  Module default used by the models; per call through `policy`.
OBJECT_POINT_POLICY = "first"
_PICK = {"first": 0, "spread": 1}


def mask_gather(point_cloud, logits, npick, want_mask=True, want_sel=True, mask=None, mean=None, sel=None,
                policy=None):
    """point_cloud (B,N,C>=3), logits (B,N,2) -> mask (B,N), mean (B,3), sel
    (B,npick) int32, obj (B,npick,3) through fpn_mask_gather.  mask / mean / sel: dense
    tensors of those shapes to write into (allocated when None).  policy: "first" |
    "spread" (default: OBJECT_POINT_POLICY)."""
    policy = OBJECT_POINT_POLICY if policy is None else policy
    if policy not in _PICK:
        raise ValueError("object-point policy must be 'first' or 'spread', got %r" % (policy,))
    _lib.require_device_tensor("point_cloud", point_cloud, torch.float32, 3)
    _lib.require_device_tensor("logits", logits, torch.float32, 3, 2)
    b, n, c = point_cloud.shape
    dev = point_cloud.device
    for name, t, dtype, shape in (("mask", mask, torch.float32, (b, n)), ("mean", mean, torch.float32, (b, 3)),
                                  ("sel", sel, torch.int32, (b, npick))):
        if t is not None:
            _lib.require_device_tensor(name, t, dtype, 2)
            if tuple(t.shape) != shape or t.device != dev:
                raise ValueError("%s must be %s on %s, got %s on %s" % (name, shape, dev, tuple(t.shape), t.device))
    if mask is None and want_mask:
        mask = torch.empty((b, n), dtype=torch.float32, device=dev)
    if mean is None:
        mean = torch.empty((b, 3), dtype=torch.float32, device=dev)
    if sel is None and want_sel:
        sel = torch.empty((b, npick), dtype=torch.int32, device=dev)
    obj = torch.empty((b, npick, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fpn_mask_gather(b, n, c, int(npick), _PICK[policy], _lib.ptr(point_cloud), _lib.ptr(logits),
                                               _lib.ptr(mask), _lib.ptr(mean), _lib.ptr(sel), _lib.ptr(obj),
                                               _lib.stream_of(point_cloud)))
    return mask, mean, sel, obj
