"""GPU parity of the Flipout path, the Keras-style variant, the ensemble statistics
and the test.py post-processing.

Flipout: PARITY UNPINNED by the reference (the arithmetic is tensorflow_probability's,
absent here).  Pinned instead: (i) identical noise -> the HIP path (folded single
product for convolutions, two-pass for dense layers) equals the textbook two-product
formula evaluated in float64 within 1e-3 scaled error; (ii) zero perturbation ->
equals the deterministic network; (iii) the output moments over many draws match the
analytic Flipout moments; (iv) a seeded generator reproduces its draws."""
import numpy as np
import pytest
import torch

from oracle import bayes_numpy as BN
from oracle import v1_numpy as V

pytestmark = pytest.mark.gpu


def _bias_generator(st, _made={}):
    """One CPU generator per variable store, seeded by a constant: the bias draws of a test do
    not depend on the process-wide generator or on which tests ran before."""
    g = _made.get(id(st))
    if g is None:
        _made.clear()
        g = _made[id(st)] = torch.Generator()
        g.manual_seed(424242)
    return g


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def close(got, want, rtol=1e-3):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    s = np.sqrt(np.mean(want ** 2)) + 1e-30
    err = np.abs(got - want) / np.maximum(np.abs(want), s)
    assert err.max() <= rtol, "max scaled error %.3e" % err.max()


def test_flipout_layer_fold_and_two_pass(device):
    from fpn import ops
    rng = np.random.RandomState(0)
    for (groups, rpg, cin, cout, pool) in [(3, 128, 64, 96, 0), (2, 256, 20, 130, 128), (37, 1, 259, 64, 0),
                                           (4, 32, 16, 40, 32)]:
        S = 2
        rows = groups * rpg
        x = rng.standard_normal((S * rows, cin)).astype(np.float32)
        w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
        dw = (rng.standard_normal((S, cin, cout)) * 0.05).astype(np.float32)
        si = rng.choice([-1.0, 1.0], (S * groups, cin)).astype(np.float32)
        so = rng.choice([-1.0, 1.0], (S * groups, cout)).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32)
        got = ops.shared_mlp_flipout(dev(x, device), dev(w, device), dev(dw, device), rows, dev(si, device),
                                     dev(so, device), rpg, shift=dev(bias, device), relu=True, pool_rows=pool)
        want = []
        for s in range(S):
            xs = x[s * rows:(s + 1) * rows].astype(np.float64)
            sis = np.repeat(si[s * groups:(s + 1) * groups], rpg, 0)
            sos = np.repeat(so[s * groups:(s + 1) * groups], rpg, 0)
            y = np.maximum(xs @ w + bias + ((xs * sis) @ dw[s].astype(np.float64)) * sos, 0)
            want.append(y.reshape(-1, pool, cout).max(1) if pool else y)
        close(got.cpu().numpy(), np.concatenate(want, 0), 1e-4)


def bayes_setup(device, B=4, N=256, seed=0, chain=True):
    from fpn import bayes, synthetic, variables
    st = variables.VariableStore(device, seed=seed + 1)
    bp = bayes.BayesPlan(st, 3, chain=chain)
    assert bp.chain == chain
    for k in list(st.keys()):
        if k.endswith("/biases"):
            st[k] = torch.randn(st[k].shape, generator=_bias_generator(st)) * 0.1
    pc, oh = synthetic.make_frustums(B, N, 3, seed=seed)
    return st, bp, pc, oh


@pytest.mark.parametrize("chain,B,N", [(True, 4, 256), (False, 4, 256), (True, 13, 96), (True, 16, 1024)])
def test_bayes_forward_matches_textbook_flipout(device, chain, B, N):
    """chain=True: register-chain kernels on per-frustum effective weights (13 x 96: frustum
    boundaries inside a workgroup and a workgroup count that is not a multiple of 8, 16 x
    1024: the XCD-grouped block order); chain=False: one Flipout layer per launch."""
    st, bp, pc, oh = bayes_setup(device, B=B, N=N, chain=chain)
    g = torch.Generator(device=device)
    g.manual_seed(5)
    S = 2
    noise = bp.sample_noise(S, pc.shape[0], g)
    ep = bp.forward(dev(pc, device), dev(oh, device), S, noise=noise)
    params = st.numpy()
    for s in range(S):
        nz = {k: ((torch.nn.functional.softplus(st[k + "/rho"]) + np.finfo(np.float32).eps) * v[0][s]).cpu().numpy()
              for k, v in noise.items()}
        nz = {k: (nz[k], noise[k][1][s].cpu().numpy(), noise[k][2][s].cpu().numpy()) for k in nz}
        want = BN.forward_sample(pc, oh, params, nz)
        close(ep["mask_logits"][s].cpu().numpy(), want["mask_logits"])
        mask = ep["mask"][s].cpu().numpy()
        if not np.array_equal(mask, want["mask"]):
            want = BN.forward_sample(pc, oh, params, nz, mask_override=mask)
        np.testing.assert_array_equal(ep["object_point_indices"][s].cpu().numpy(), want["object_point_indices"])
        for k in ("center_delta", "stage1_center", "box_output", "center", "size_residuals"):
            close(ep[k][s].cpu().numpy(), want[k])
    # different samples really differ, and a re-seeded generator reproduces them
    assert not torch.equal(ep["box_output"][0], ep["box_output"][1])
    g.manual_seed(5)
    ep2 = bp.forward(dev(pc, device), dev(oh, device), S, generator=g)
    assert torch.equal(ep["box_output"], ep2["box_output"])


def test_bayes_zero_perturbation_equals_deterministic(device):
    from fpn import plan
    st, bp, pc, oh = bayes_setup(device, seed=3)
    noise = bp.sample_noise(1, pc.shape[0])
    noise = {k: (torch.zeros_like(v[0]), v[1], v[2]) for k, v in noise.items()}
    ep = bp.forward(dev(pc, device), dev(oh, device), 1, noise=noise)
    det = plan.V1Plan(st, 3, bn=False, chain=False).forward(dev(pc, device), dev(oh, device))
    close(ep["mask_logits"][0].cpu().numpy(), det["mask_logits"].cpu().numpy(), 1e-4)
    want = V.get_model(pc, oh, st.numpy(), bn=False)
    close(det["mask_logits"].cpu().numpy(), want["mask_logits"])


def test_flipout_moments(device):
    """E[y] = x.W + b and Var[y_j] = sum_k x_k^2 sigma_kj^2 for independent draws."""
    from fpn import ops
    rng = np.random.RandomState(2)
    cin, cout, rows, S = 24, 16, 128, 4000
    x1 = rng.standard_normal((1, cin)).astype(np.float32)
    x = np.repeat(x1, rows, 0)
    w = rng.standard_normal((cin, cout)).astype(np.float32)
    sigma = rng.uniform(0.05, 0.2, (cin, cout)).astype(np.float32)
    g = torch.Generator(device=device)
    g.manual_seed(9)
    eps = torch.randn((S, cin, cout), device=device, generator=g)
    si = torch.randint(0, 2, (S, cin), device=device, generator=g).float() * 2 - 1
    so = torch.randint(0, 2, (S, cout), device=device, generator=g).float() * 2 - 1
    xs = dev(x, device).repeat(S, 1)
    y = ops.shared_mlp_flipout(xs, dev(w, device), (dev(sigma, device) * eps).contiguous(), rows, si, so, rows,
                               relu=False).reshape(S, rows, cout)[:, 0]
    mean = y.mean(0).cpu().numpy()
    var = y.var(0).cpu().numpy()
    want_mean = x1[0].astype(np.float64) @ w
    want_var = (x1[0].astype(np.float64) ** 2) @ (sigma.astype(np.float64) ** 2)
    assert np.abs(mean - want_mean).max() < 5 * np.sqrt(want_var.max() / S)
    np.testing.assert_allclose(var, want_var, rtol=0.15)


@pytest.mark.parametrize("n", [1, 3, 31, 32, 33, 4096, 100003])
def test_flipout_signs_are_the_given_bits(device, n):
    """fpn_flipout_signs: sign i = +1 where bit i % 32 of word i // 32 is set, else -1 - every
    length (whole 16-byte groups and ragged ends), bit for bit against the definition."""
    from fpn import ops
    rng = np.random.RandomState(n)
    words = rng.randint(0, 1 << 32, (n + 31) // 32, dtype=np.uint64).astype(np.uint32)
    got = ops.flipout_signs(dev(words.view(np.int32), device), n).cpu().numpy()
    i = np.arange(n)
    want = np.where((words[i // 32] >> (i % 32).astype(np.uint32)) & 1, 1.0, -1.0).astype(np.float32)
    np.testing.assert_array_equal(got, want)
    with pytest.raises(ValueError):
        ops.flipout_signs(dev(words.view(np.int32), device), 32 * words.size + 1)


def test_sample_noise_layout_and_fairness(device):
    """BayesPlan.sample_noise: every layer's sign matrices are +-1 with the documented shapes, dense for
    the fully connected layers (their kernel takes them as they are), rows on 16-byte boundaries for the
    others; signs and dropout bits are fair (|mean| within 5 standard errors) and a seeded generator
    reproduces the draw."""
    from fpn import bayes, variables
    st = variables.VariableStore(device, seed=3)
    bp = bayes.BayesPlan(st, 4, mc_dropout=True)
    S, B, N = 3, 5, 64
    g = torch.Generator(device=device)
    g.manual_seed(77)
    noise = bp.sample_noise(S, B, g, N)
    table = {scope: (cin, cout) for scope, cin, cout, _, _ in bp.table}
    total = 0.0
    count = 0
    for scope in bp.flip_scopes:
        eps, si, so = noise[scope]
        cin, cout = table[scope]
        assert tuple(eps.shape) == (S, cin, cout) and tuple(si.shape) == (S, B, cin) and tuple(so.shape) == (S, B, cout)
        for t in (si, so):
            assert bool((t.abs() == 1).all()), scope
            assert t.stride(2) == 1 and t.stride(0) == B * t.stride(1)
            if scope.startswith("fc"):
                assert t.is_contiguous(), scope
            else:
                assert t.stride(1) % 4 == 0 and t.data_ptr() % 16 == 0, scope
            total += float(t.sum())
            count += t.numel()
    assert abs(total / count) < 5 / np.sqrt(count)
    keep = bayes.unpack_keep_bits(noise["dropout"])
    assert tuple(noise["dropout"].shape) == (S, B, N, 4) and noise["dropout"].dtype == torch.int32
    per_bit = keep.reshape(-1, 128).mean(0).cpu().numpy()          # every bit position of every word
    assert np.abs(per_bit - 0.5).max() < 5 * 0.5 / np.sqrt(S * B * N)
    g.manual_seed(77)
    again = bp.sample_noise(S, B, g, N)
    for scope in bp.flip_scopes:
        for a, b in zip(noise[scope], again[scope]):
            assert torch.equal(a, b), scope
    assert torch.equal(noise["dropout"], again["dropout"])


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("dropout", [False, True])
def test_flipout_streams_from_sign_bits_are_the_same_bits(device, monkeypatch, dropout, precision):
    """The per-copy weights built from one bit per sign (fpn_flipout_stream_bits_bf16 for the bf16
    streams, fpn_flipout_weights_bits_f32 for the fp32 chains - what a plan does with noise it drew
    itself) against those built from the +-1 floats spread from the same bits: every end point of the whole Flipout pass the same bits.  Three samples of five frustums,
    so the copies of a pass start at a row other than 0 in the second block."""
    from fpn import bayes, synthetic, variables
    st = variables.VariableStore(device, seed=41)
    bp = bayes.BayesPlan(st, 4, mc_dropout=dropout, precision=precision)
    bp.CHAIN_COPIES = 10                       # two samples a pass: the second pass starts at row 10
    S, B, N = 4, 5, 128
    pc, oh = synthetic.make_frustums(B, N, 4, seed=42)
    g = torch.Generator(device=device)
    g.manual_seed(43)
    noise = bp.sample_noise(S, B, g, N)
    assert noise.sign_bits is not None and set(noise.sign_pos) == set(bp.flip_scopes)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FPN_FLIPOUT_BITS", flag)
        ep = bp.forward(dev(pc, device), dev(oh, device), S, noise=noise)
        got[flag] = {k: ep[k].clone() for k in ("mask_logits", "box_output", "stage1_center", "center")}
    assert float(got["1"]["mask_logits"].abs().max()) > 0
    for k in got["1"]:
        assert torch.equal(got["1"][k], got["0"][k]), k


def test_mc_statistics(device):
    from fpn import uncertainty
    rng = np.random.RandomState(4)
    S, B = 32, 50
    box = rng.standard_normal((S, B, 59)).astype(np.float32)
    box[..., :3] += 30.0
    c1 = (rng.standard_normal((S, B, 3)) * 0.01 + 20).astype(np.float32)
    stats = uncertainty.mc_statistics(dev(box, device), dev(c1, device))
    got = {k: v.cpu().numpy() for k, v in uncertainty.unpack(stats).items()}
    want = BN.mc_statistics(box, c1)
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=2e-5, atol=2e-6, err_msg=k)
    # identical samples: zero variance, entropy of a single softmax
    same = np.repeat(box[:1], S, 0)
    st2 = uncertainty.unpack(uncertainty.mc_statistics(dev(same, device), dev(np.repeat(c1[:1], S, 0), device)))
    assert float(st2["total_variance"].abs().max()) < 1e-6


def test_keras_and_bayes_model_surfaces(device):
    from fpn import inference, synthetic, variables
    from fpn.models import FPointNet_Bayes, FPointNet_Keras, frustum_pointnet
    pc, oh = synthetic.make_frustums(4, 1024, 3, seed=1)
    st = variables.VariableStore(device, seed=2)
    with variables.use_store(st):
        train, det = FPointNet_Keras.Frustum_Pointnet_Model(batch_size=4)
        outs = det.predict({"frustum_point_cloud": pc, "one_hot_vec": oh})
        ep = FPointNet_Keras.get_model(dev(pc, device), dev(oh, device), False)
    assert [o.shape for o in outs] == [(4, 1024, 2), (4, 3), (4, 12), (4, 12), (4, 8), (4, 8, 3)]
    want = V.get_model(pc, oh, st.numpy(), bn=False)
    close(outs[0], want["mask_logits"])
    if np.array_equal(ep["mask"].cpu().numpy(), want["mask"]):
        close(outs[1], want["center"])
        close(outs[5], want["size_residuals"])
    with pytest.raises(NotImplementedError):
        train.fit(None)
    # test.py:inference post-processing on the det outputs
    got = inference.postprocess(*outs)
    ref = V.inference_postprocess(*outs)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    assert got[0].shape == (4, 1024) and got[6].shape == (4,)

    st2 = variables.VariableStore(device, seed=3)
    with variables.use_store(st2):
        _, bdet = FPointNet_Bayes.Frustum_Pointnet_Model(batch_size=4)
        o1 = bdet.predict([pc, oh])
        o2 = bdet.predict([pc, oh])
        ep = FPointNet_Bayes.get_model(dev(pc, device), dev(oh, device), False, nsamples=3)
        _, d11 = frustum_pointnet.frustum_pointnet(point_cloud_shape=(1024, 3))
        o11 = d11.predict([pc, oh])
    assert len(o1) == 6 and not np.array_equal(o1[0], o2[0])       # every predict draws fresh weights
    assert ep["box_output"].shape == (3, 4, 59) and ep["mc_statistics"].shape == (4, 30)
    assert len(o11) == 11 and o11[0].shape == (4, 1024) and o11[1].shape == (4, 1024, 2)


@pytest.mark.parametrize("chain,N", [(True, 160), (False, 128)])
def test_bayes_mc_dropout(device, chain, N):
    """Monte-Carlo dropout before the 2-channel head (FPointNet_Bayes.py:92 kept active):
    the keep bits are part of the noise, both paths apply them (the chain path inside the
    tail kernel) and match the oracle given the same bits."""
    from fpn import bayes, synthetic, variables
    st = variables.VariableStore(device, seed=2)
    bp = bayes.BayesPlan(st, 3, mc_dropout=True, chain=chain)
    for k in list(st.keys()):
        if k.endswith("/biases"):
            st[k] = torch.randn(st[k].shape, generator=_bias_generator(st)) * 0.1
    B, S = 5, 2       # (the per-layer kernel pools groups of 32, 64 or a multiple of 128 rows)
    pc, oh = synthetic.make_frustums(B, N, 3, seed=4)
    g = torch.Generator(device=device)
    g.manual_seed(9)
    noise = bp.sample_noise(S, B, g, N)
    bits = noise["dropout"].cpu().numpy()
    assert bits.shape == (S, B, N, 4) and bits.dtype == np.int32
    keep = ((bits[..., None] >> np.arange(32)) & 1).reshape(S, B, N, 128).astype(bool)
    assert 0.45 < keep.mean() < 0.55
    ep = bp.forward(dev(pc, device), dev(oh, device), S, noise=noise)
    params = st.numpy()
    for s in range(S):
        nz = {k: ((torch.nn.functional.softplus(st[k + "/rho"]) + np.finfo(np.float32).eps) * v[0][s]).cpu().numpy()
              for k, v in noise.items() if k != "dropout"}
        nz = {k: (nz[k], noise[k][1][s].cpu().numpy(), noise[k][2][s].cpu().numpy()) for k in nz}
        want = BN.forward_sample(pc, oh, params, nz, dropout_keep=keep[s])
        close(ep["mask_logits"][s].cpu().numpy(), want["mask_logits"])
        plain = BN.forward_sample(pc, oh, params, nz)
        assert np.abs(plain["mask_logits"] - want["mask_logits"]).max() > 1e-3


def test_flipout_effective_weights_layouts(device):
    """fpn_flipout_weights_f32 against W_mean + diag(s_in) dW diag(s_out) in every layout
    (row-major, chunked output-tile major, chunked input-tile major), several samples per
    call and a column slice as sign_in."""
    from fpn import ops
    rng = np.random.RandomState(3)
    for cin, cout, layouts in ((3, 64, (0,)), (64, 128, (0, 1, 2)), (128, 2, (0,)), (96, 64, (1, 2))):
        S, per = 3, 4
        w = rng.standard_normal((cin, cout)).astype(np.float32)
        dw = rng.standard_normal((S, cin, cout)).astype(np.float32)
        si_wide = rng.choice([-1.0, 1.0], (S * per, cin + 32)).astype(np.float32)
        so = rng.choice([-1.0, 1.0], (S * per, cout)).astype(np.float32)
        si = si_wide[:, :cin]
        want = w[None] + si[:, :, None] * np.repeat(dw, per, 0) * so[:, None, :]      # (copies, cin, cout)
        si_dev = dev(si_wide, device)[:, :cin]
        for layout in layouts:
            got = ops.flipout_weights(dev(w, device), dev(dw, device), si_dev, dev(so, device), layout)
            if layout == 0:
                ref = want.reshape(S * per, -1)
            else:
                ref = np.stack([ops.tile_weights_reg(dev(m, device), layout == 2).cpu().numpy() for m in want])
            np.testing.assert_array_equal(got.cpu().numpy(), ref.astype(np.float32), err_msg=str((cin, cout, layout)))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_bayes_frustum_blocks_and_single_sample(device, precision):
    """Frustum blocks smaller than the batch (sign matrices sliced per block, ragged last
    block, several samples per pass) give exactly the results of one block; one frustum
    with one sample runs."""
    from fpn import bayes, synthetic, variables
    st = variables.VariableStore(device, seed=6)
    whole = bayes.BayesPlan(st, 3, precision=precision)
    for k in list(st.keys()):
        if k.endswith("/biases"):
            st[k] = torch.randn(st[k].shape, generator=_bias_generator(st)) * 0.1
    B, N, S = 10, 128, 3
    pc, oh = synthetic.make_frustums(B, N, 3, seed=1)
    g = torch.Generator(device=device)
    g.manual_seed(3)
    noise = whole.sample_noise(S, B, g)
    a = whole.forward(dev(pc, device), dev(oh, device), S, noise=noise)
    for chunk in (4, 16):       # 4: blocks of 4, 4, 2 frustums, one sample per pass; 16: one block, one sample per pass
        blocks = bayes.BayesPlan(st, 3, precision=precision, chunk_frustums=chunk)
        b = blocks.forward(dev(pc, device), dev(oh, device), S, noise=noise)
        for k in a:
            assert torch.equal(a[k], b[k]), (chunk, k)
    one = whole.forward(dev(pc[:1], device), dev(oh[:1], device), 1, generator=g)
    assert one["box_output"].shape == (1, 1, 59) and torch.isfinite(one["box_output"]).all()


def test_flipout_layer_random_shapes(device):
    """Twenty seeded random Flipout layers: sign groups of 1, 7, 32, 128 or 256 rows (the
    last two take the folded-weight mode), 1-3 samples per launch, odd channel counts,
    optional per-group bias and fused pooling."""
    from fpn import ops
    rng = np.random.RandomState(31)
    for trial in range(20):
        rpg = int(rng.choice([1, 7, 32, 128, 256]))
        groups = int(rng.randint(1, 6))
        S = int(rng.choice([1, 2, 3]))
        rows = groups * rpg
        cin = int(rng.choice([3, 20, 64, 131, 259]))
        cout = int(rng.choice([2, 40, 64, 130, 256]))
        pool = rpg if (rpg in (32, 128, 256) and rng.randint(2)) else 0
        use_gb = bool(rng.randint(2))
        x = rng.standard_normal((S * rows, cin)).astype(np.float32)
        w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
        dw = (rng.standard_normal((S, cin, cout)) * 0.05).astype(np.float32)
        si = rng.choice([-1.0, 1.0], (S * groups, cin)).astype(np.float32)
        so = rng.choice([-1.0, 1.0], (S * groups, cout)).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32)
        gb = rng.standard_normal((S * groups, cout)).astype(np.float32) if use_gb else None
        relu = bool(rng.randint(2))
        got = ops.shared_mlp_flipout(dev(x, device), dev(w, device), dev(dw, device), rows, dev(si, device),
                                     dev(so, device), rpg, shift=dev(bias, device), relu=relu, pool_rows=pool,
                                     group_bias=None if gb is None else dev(gb, device))
        want = []
        for s in range(S):
            xs = x[s * rows:(s + 1) * rows].astype(np.float64)
            sis = np.repeat(si[s * groups:(s + 1) * groups], rpg, 0)
            sos = np.repeat(so[s * groups:(s + 1) * groups], rpg, 0)
            y = xs @ w + ((xs * sis) @ dw[s].astype(np.float64)) * sos
            if use_gb:
                y = y + np.repeat(gb[s * groups:(s + 1) * groups], rpg, 0)
            y = y + bias
            if relu:
                y = np.maximum(y, 0)
            want.append(y.reshape(-1, pool, cout).max(1) if pool else y)
        try:
            close(got.cpu().numpy(), np.concatenate(want, 0), 1e-4)
        except AssertionError as e:
            raise AssertionError("trial %r: %s" % ((trial, rpg, groups, S, cin, cout, pool, use_gb, relu), e))


@pytest.mark.parametrize("chain", [True, False])
def test_flipout_with_batch_norm_and_flipout_outputs(device, chain):
    """The network of models/frustum_pointnet.py:108-295 - Flipout in EVERY layer, the two
    output layers included, batch norm after every layer but the last (conv10 and fc3-stage1
    too) - in fp32 on the chain kernels and on the per-layer kernels against the float64
    estimator with the same noise, at the fp32 bar; the two paths agree with each other."""
    import bf16_bounds as BB
    from fpn import synthetic
    from oracle import bayes_numpy as BN
    B, N, S = 5, 128, 2
    st, bp = BB.bayes_setup(device, 31, False, precision="fp32", full=True, chain=chain)
    assert "fc3/rho" in st and "conv10/bn/gamma" in st and "fc3-stage1/bn/gamma" in st and "fc3/bn/gamma" not in st
    pc, oh = synthetic.make_frustums(B, N, 3, seed=32)
    g = torch.Generator(device=device)
    g.manual_seed(33)
    noise = bp.sample_noise(S, B, g, N)
    assert "fc3" in noise and "fc3-stage1" in noise
    ep = bp.forward(dev(pc, device), dev(oh, device), S, noise=noise)
    params = {k: v.astype(np.float64) for k, v in st.numpy().items()}
    pc64, oh64 = pc.astype(np.float64), oh.astype(np.float64)
    for s in range(S):
        nz = BB.flipout_noise_of_sample(st, noise, s)
        want = BN.forward_sample(pc64, oh64, params, nz, bn=True)
        close(ep["mask_logits"][s].cpu().numpy(), want["mask_logits"], 1e-3)
        mask = ep["mask"][s].cpu().numpy()
        want = BN.forward_sample(pc64, oh64, params, nz, bn=True, mask_override=mask, recentre_box_input=False)
        for k in ("center_delta", "stage1_center", "box_output", "center"):
            close(ep[k][s].cpu().numpy(), want[k], 1e-3)
    assert not torch.equal(ep["box_output"][0], ep["box_output"][1])
    # batch norm matters: the same noise without the statistics gives other logits
    plain = BN.forward_sample(pc64, oh64, params, BB.flipout_noise_of_sample(st, noise, 0), bn=False)
    assert np.abs(plain["mask_logits"] - ep["mask_logits"][0].cpu().numpy()).max() > 1e-2


def test_frustum_pointnet_module_builds_that_network(device):
    """fpn.models.frustum_pointnet: constructor name, (train, detection) pair, eleven outputs, and
    behind them the plan with batch norm and Flipout output layers."""
    from fpn import synthetic, variables
    from fpn.models import frustum_pointnet as FP
    st = variables.VariableStore(device, seed=41)
    with variables.use_store(st):
        train, det = FP.frustum_pointnet()                  # default: 4-channel points, xyz + intensity
        pc, oh = synthetic.make_frustums(4, 128, 4, seed=42)
        labels = [np.zeros((4,) + s, np.float32) for s in ((3, 3), (128,), (3,), (), (), (), (3,))]
        outs = det.predict([pc, oh] + labels, batch_size=4)
        ep = FP.get_model(dev(pc, device), dev(oh, device), False)
    assert len(outs) == len(FP.DETECTION_OUTPUTS) == 11
    # outputs=[mask, logits, center, stage1_center, boxnet_center, ...] (frustum_pointnet.py:293-295)
    assert outs[0].shape == (4, 128) and outs[1].shape == (4, 128, 2) and outs[2].shape == (4, 3)
    np.testing.assert_allclose(outs[2], outs[3] + outs[4], rtol=1e-6, atol=1e-6)     # :253
    assert "conv10/bn/moving_variance" in st and "fc3/rho" in st and st["conv1/weights"].shape == (4, 64)
    assert ep["mask_logits"].shape == (4, 128, 2) and ep["center"].shape == (4, 3)


def test_box_stack_on_unshifted_object_points(device):
    """models/frustum_pointnet.py:214: the box stack's input is the gathered object points
    without the T-net's shift.  Chain and per-layer paths, against the estimator with the
    same option; the option changes the box output and nothing before it."""
    import bf16_bounds as BB
    from fpn import bayes, synthetic
    from oracle import bayes_numpy as BN
    B, N = 4, 128
    pc, oh = synthetic.make_frustums(B, N, 3, seed=52)
    got = {}
    for chain in (True, False):
        st, _ = BB.bayes_setup(device, 51, False, precision="fp32", full=True, chain=chain)
        bp = bayes.BayesPlan(st, 3, chain=chain, bn=True, flipout_outputs=True, recentre_box_input=False)
        g = torch.Generator(device=device)
        g.manual_seed(53)
        noise = bp.sample_noise(1, B, g, N)
        ep = bp.forward(dev(pc, device), dev(oh, device), 1, noise=noise)
        params = {k: v.astype(np.float64) for k, v in st.numpy().items()}
        nz = BB.flipout_noise_of_sample(st, noise, 0)
        mask = ep["mask"][0].cpu().numpy()
        kw = dict(bn=True, mask_override=mask)
        want = BN.forward_sample(pc.astype(np.float64), oh.astype(np.float64), params, nz, recentre_box_input=False, **kw)
        other = BN.forward_sample(pc.astype(np.float64), oh.astype(np.float64), params, nz, **kw)
        close(ep["box_output"][0].cpu().numpy(), want["box_output"], 1e-3)
        close(ep["center"][0].cpu().numpy(), want["center"], 1e-3)
        np.testing.assert_array_equal(want["center_delta"], other["center_delta"])
        assert np.abs(want["box_output"] - other["box_output"]).max() > 1e-3
        got[chain] = ep["box_output"][0].cpu().numpy()
    close(got[True], got[False], 1e-3)
