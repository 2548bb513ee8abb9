"""GPU tests of the bf16 layer-chain kernels (bf16 weights and activations between the
layers, fp32 accumulation, fp32 first layers / shifts / maxima / final layer).

Three kinds of check:
  * EXACT: with weights that only select and scale inputs (one non-zero per column,
    small integers) and small-integer inputs every activation is an integer below 256
    and therefore exact in bf16: the bf16 kernels must then agree with the fp32 kernels
    BIT FOR BIT, which pins every operand layout (a wrong lane or channel order picks a
    different input and shows);
  * ROUNDING MODEL: with random weights the whole networks (v1 with batch norm, Flipout
    with and without Monte-Carlo dropout) are compared with the float64 network that has
    the kernels' roundings written out (oracle/bf16_numpy.py), bound MODEL_TOL = 4 u,
    u = 2^-9 - the parity check of the kernels, independent of the draw;
  * DISTANCE TO THE UNROUNDED NETWORK: the stated looser bound of the bf16 path,
    EXACT_TOL = 16 u of the summed magnitude of an output's last layer, against the
    float64 oracles, v1 / Flipout / v2.
  The bounds, their measure and their derivation are in tests/bf16_bounds.py; both are
  asserted over sweeps of 32 seeded draws per workload inside one test, and the observed
  maxima are printed.  Single modules and layers (pooled ReLU features, one product) are
  held to BF16_RTOL = 4e-2 of max(|want|, rms(want)) against their fp32 twins, over seed
  sweeps as well.
"""
import numpy as np
import pytest
import torch

import bf16_bounds as BB
from oracle import v1_numpy as V

pytestmark = pytest.mark.gpu

BF16_RTOL = 4e-2


def scaled_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    s = np.sqrt(np.mean(want ** 2)) + 1e-30
    return float((np.abs(got - want) / np.maximum(np.abs(want), s)).max())


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def selection_store(device, seed):
    """v1 variables without batch norm whose weight matrices have one non-zero per
    column: the first layers multiply an input by a small integer, every other layer
    copies one of its inputs."""
    from fpn import plan, variables
    st = variables.VariableStore(device, seed=seed)
    plan.init_variables(st, 4, bn=False)
    rng = np.random.RandomState(seed)
    for scope, cin, cout, _, _ in plan.layer_table(4):
        w = np.zeros((cin, cout), np.float32)
        cols = np.arange(cout)
        if scope == "conv1":
            w[cols % 4, cols] = cols // 4 + 1                # x <= 7, factor <= 16: values <= 112
        elif scope in ("conv-reg1-stage1", "conv-reg1"):
            w[cols % 3, cols] = 1 + (cols // 3) % 4
        else:
            w[rng.randint(0, cin, cout), cols] = 1
        st[scope + "/weights"] = w
    return st


def test_bf16_chains_exact_on_integer_data(device):
    from fpn import model_util, ops, plan
    st = selection_store(device, 3)
    f32 = plan.V1Plan(st, 4, False)
    b16 = plan.V1Plan(st, 4, False, precision="bf16")
    rng = np.random.RandomState(0)
    # 256, 1024: a workgroup's 128 rows lie in one frustum (maxima merged in LDS first);
    # 64, 32: two / four frustums per workgroup; 96: frustum boundaries inside a workgroup
    # and a last workgroup with idle waves (direct memory atomics); 7 x 32: a partial last
    # workgroup next to full ones
    for b_, n in ((3, 256), (2, 1024), (5, 96), (4, 64), (8, 32), (7, 32)):
        pc = rng.randint(0, 8, (b_, n, 4)).astype(np.float32)
        one_hot = np.eye(3, dtype=np.float32)[rng.randint(0, 3, b_)]
        a = f32.forward(dev(pc, device), dev(one_hot, device))
        c = b16.forward(dev(pc, device), dev(one_hot, device))
        assert float(a["mask_logits"].abs().max()) > 0
        assert torch.equal(a["mask_logits"], c["mask_logits"]), (b_, n)
        # and against the oracle (exact as well: integers all the way)
        want = V.get_model(pc, one_hot, st.numpy(), bn=False)
        np.testing.assert_array_equal(c["mask_logits"].cpu().numpy(), want["mask_logits"])

    # object-point stacks on integer coordinates, with and without the centre shift
    m = model_util.NUM_OBJECT_POINT
    for box in (False, True):
        width = 512 if box else 256
        for b_ in (1, 4):
            obj = dev(rng.randint(0, 8, (b_ * m, 3)).astype(np.float32), device)
            delta = dev(rng.randint(0, 4, (b_, 3)).astype(np.float32), device) if box else None
            want = torch.zeros((b_, width + 3), device=device)
            got = torch.zeros((b_, width + 3), device=device)
            ops.v1_obj_chain(obj, m, box, delta, f32._box_table if box else f32._tnet_table, want)
            ops.v1_obj_chain_bf16(obj, m, box, delta, b16._box_ws if box else b16._tnet_ws,
                                  b16._box_consts if box else b16._tnet_consts, got)
            assert float(want.abs().max()) > 0
            assert torch.equal(want, got), (box, b_)


def _report(name, model, exact):
    print("%s, errors in units of u = 2^-9 over the summed magnitude: vs rounding model %s; vs unrounded network %s" % (
        name, {k: "%.2f" % (v / BB.U) for k, v in model.items()}, {k: "%.2f" % (v / BB.U) for k, v in exact.items()}))


def _assert_bounds(name, model, exact):
    _report(name, model, exact)
    assert max(model.values()) <= BB.MODEL_TOL, (name, model)
    assert max(exact.values()) <= BB.EXACT_TOL, (name, exact)
    assert exact["mask_logits"] > 1e-5, "suspiciously exact: is the bf16 path running?"


@pytest.mark.parametrize("case", ["v1", "flipout", "flipout_dropout", "flipout_batch_norm"])
def test_bf16_bounds_over_a_sweep_of_draws(device, case):
    """32 draws of weights (and batch-norm statistics / biases / Flipout noise / dropout bits)
    and inputs, 4 frustums x 256 points each: every output of every draw within MODEL_TOL of
    the rounding model and within EXACT_TOL of the unrounded float64 network."""
    model, exact = BB.sweep(device, case, BB.DRAWS)
    _assert_bounds("%s, %d draws" % (case, BB.DRAWS), model, exact)


@pytest.mark.parametrize("B,N,seed", [(8, 1024, 101), (3, 160, 102), (5, 96, 103)])
def test_bf16_v1_bounds_other_shapes(device, B, N, seed):
    """1024-point frustums (a workgroup inside one frustum, maxima merged in LDS), 160 and 96
    points (frustum boundaries inside workgroups, idle waves, direct memory atomics)."""
    _assert_bounds("v1 %dx%d" % (B, N), *BB.v1_draw(device, seed, B, N))


def test_bf16_set_abstraction_within_stated_bound(device):
    """The fused set-abstraction kernel in bf16 against its fp32 twin, every shape the v2
    topology uses (features 0 / 1 / 128 / 320 wide, both concatenation orders), five draws of
    weights, batch-norm statistics, coordinates and features per shape.  The outputs are
    pooled ReLU features (no cancelling last sum), three layers deep: BF16_RTOL of
    max(|want|, rms(want))."""
    from fpn import pointnet_util as U
    from fpn import variables
    b = 2
    worst = {}
    for draw in range(5):
        rng = np.random.RandomState(draw)
        xyz_big = rng.uniform(0, 1, (b, 1024, 3)).astype(np.float32)
        xyz_small = rng.uniform(0, 1, (b, 128, 3)).astype(np.float32)
        cases = [
            ("msg", xyz_big, rng.standard_normal((b, 1024, 1)).astype(np.float32),
             (64, [0.1, 0.2, 0.4], [32, 64, 128], [[32, 32, 64], [64, 64, 128], [64, 96, 128]])),
            ("msg", xyz_small, rng.standard_normal((b, 128, 320)).astype(np.float32),
             (32, [0.4, 0.8, 1.6], [64, 64, 128], [[64, 64, 128], [128, 128, 256], [128, 128, 256]])),
            ("ssg", xyz_big, None, (64, 0.2, 64, [64, 64, 128])),
            ("ssg", xyz_big, None, (64, 0.3, 32, [64, 96, 128])),       # three-tile middle layer, one point group per wave
            ("ssg", xyz_small, rng.standard_normal((b, 128, 128)).astype(np.float32), (32, 0.4, 64, [128, 128, 256])),
            # group-all as one ball per frustum: coordinates in metres, not translated
            ("all", xyz_small[:, :32] * 30 + 5, rng.standard_normal((b, 32, 640)).astype(np.float32), [128, 256, 1024]),
            ("all", xyz_small[:, :32] * 30 - 20, rng.standard_normal((b, 32, 256)).astype(np.float32), [256, 256, 512]),
        ]
        for ci, (kind, xyz, pts, args) in enumerate(cases):
            st = variables.VariableStore(device, seed=50 + draw)
            outs = {}
            for precision in ("fp32", "fp32", "bf16"):     # first pass creates the variables
                U.SA_PRECISION = precision
                try:
                    with variables.use_store(st):
                        p = None if pts is None else dev(pts, device)
                        if kind == "msg":
                            _, feat = U.pointnet_sa_module_msg(dev(xyz, device), p, *args, False, None, scope='s')
                        elif kind == "all":
                            nx, feat, ix = U.pointnet_sa_module(dev(xyz.astype(np.float32), device), p, None, None, None, args,
                                                                None, True, False, None, scope='s')
                            assert nx.shape == (b, 1, 3) and float(nx.abs().max()) == 0.0
                            assert torch.equal(ix[0, 0].cpu(), torch.arange(32, dtype=torch.int32)) and ix.shape == (b, 1, 32)
                        else:
                            _, feat, _ = U.pointnet_sa_module(dev(xyz, device), p, *args, None, False, False, None, scope='s')
                finally:
                    U.SA_PRECISION = "fp32"
                if "init" not in outs:
                    outs["init"] = True
                    st.randomize_batch_norm(70 + draw)
                    continue
                outs[precision] = feat.cpu().numpy()
            assert any(len(k) >= 4 and k[3] == "bf16" for k in st.__dict__["_sa_chain_cache"]), "bf16 kernel not taken"
            err = scaled_err(outs["bf16"], outs["fp32"])
            assert 1e-6 < err <= BF16_RTOL, (draw, kind, args[0], err)
            worst[ci] = max(worst.get(ci, 0.0), err)
    print("set abstraction bf16 vs fp32, worst of 5 draws per shape:", {k: "%.4f" % v for k, v in worst.items()})


def test_bf16_flipout_streams_exact(device):
    """fpn_flipout_stream_bf16: every copy's units equal the packed stream of that copy's
    effective weights W_mean + diag(s_in) dW diag(s_out) (signs are +-1: one rounding),
    several layers back to back, several samples per call, a column slice as sign_in."""
    from fpn import ops
    rng = np.random.RandomState(7)
    S, per = 2, 3
    copies = S * per
    shapes = ((64, 512), (512, 256), (32, 64))
    layers, effs = [], []
    for cin, cout in shapes:
        w = rng.standard_normal((cin, cout)).astype(np.float32)
        dw = rng.standard_normal((S, cin, cout)).astype(np.float32)
        si_wide = rng.choice([-1.0, 1.0], (copies, cin + 5)).astype(np.float32)
        so = rng.choice([-1.0, 1.0], (copies, cout)).astype(np.float32)
        layers.append((dev(w, device), dev(dw, device), dev(si_wide, device)[:, :cin], dev(so, device)))
        effs.append(w[None] + si_wide[:, :cin, None] * np.repeat(dw, per, 0) * so[:, None, :])
    got, copy_vecs = ops.flipout_streams_bf16(copies, layers)
    total = sum(a * b for a, b in shapes)
    assert copy_vecs * 8 == total
    for f in range(copies):
        want = ops.chain_stream_bf16([(dev(e[f], device), None) for e in effs])[:total]
        assert torch.equal(got[f * total:(f + 1) * total].view(torch.int16), want.view(torch.int16)), f


@pytest.mark.parametrize("B,N,dropout,seed", [(13, 96, True, 201), (16, 1024, False, 202), (16, 1024, True, 203)])
def test_bf16_flipout_bounds_other_shapes(device, B, N, dropout, seed):
    """The Flipout network on the bf16 chains (per-copy weight streams) with Monte-Carlo
    dropout in the tail kernel; 13 x 96: frustum boundaries inside workgroups, 16 x 1024:
    XCD-grouped order, the workgroup's LDS ring as the weight feed."""
    _assert_bounds("Flipout %dx%d dropout=%s" % (B, N, dropout), *BB.bayes_draw(device, seed, B, N, dropout))


def test_bf16_v2_bound_over_a_sweep_of_draws(device):
    """The v2 topology with every fused kernel in bf16, 32 draws of weights, batch-norm
    statistics and inputs, ONE frustum of 1024 points each (grids of a few workgroups, no XCD
    regrouping), against the float64 numpy restatement under EXACT_TOL."""
    _, exact = BB.sweep(device, "v2", BB.DRAWS)
    print("v2, %d draws, errors in u over the summed magnitude vs the unrounded network: %s" % (
        BB.DRAWS, {k: "%.2f" % (v / BB.U) for k, v in exact.items()}))
    assert max(exact.values()) <= BB.EXACT_TOL, exact
    assert exact["mask_logits"] > 1e-5, "suspiciously exact: is the bf16 path running?"


@pytest.mark.parametrize("B,N,dropout", [(4, 256, False), (16, 1024, True), (13, 96, True)])
def test_bf16_weight_feeds_agree_bit_for_bit(device, monkeypatch, B, N, dropout):
    """The seg tail takes its weight units either through each wave's own register ring or
    through the workgroup's ring in LDS (full workgroups on one stream; 13 x 96 stays on the
    register ring either way): same arithmetic in the same order, so the same bits - shared
    stream (plain v1) and one stream per frustum copy (Flipout)."""
    from fpn import bayes, synthetic, variables
    st = variables.VariableStore(device, seed=4)
    bp = bayes.BayesPlan(st, 3, mc_dropout=dropout, precision="bf16")
    pc, oh = synthetic.make_frustums(B, N, 3, seed=B)
    pcd, ohd = dev(pc, device), dev(oh, device)
    g = torch.Generator(device=device)
    g.manual_seed(11)
    noise = bp.sample_noise(2, B, g, N)
    got = {}
    for feed in ("ring", "lds"):
        monkeypatch.setenv("FPN_BF16_FEED", feed)
        ep = bp.forward(pcd, ohd, 2, noise=noise)
        got[feed] = {k: ep[k].clone() for k in ("mask_logits", "box_output", "stage1_center")}
    for k in got["ring"]:
        assert torch.equal(got["ring"][k], got["lds"][k]), k


@pytest.mark.parametrize("B,N,dropout", [(4, 256, False), (16, 1024, True), (13, 96, True), (3, 160, False), (8, 128, True)])
def test_bf16_tail_forms_agree_bit_for_bit(device, monkeypatch, B, N, dropout):
    """The seg tail with conv6 interleaved into conv7 (one and two point groups per wave, both
    weight feeds) against the layer-by-layer kernel: every accumulator sees the same MFMAs
    in the same order, so the logits - and everything after them - are the same bits; shared
    stream (plain v1) and one stream per frustum copy (Flipout, with the dropout bits).
    96- and 160-point frustums have no whole 64-point groups: the two-group form falls back
    to one group there, 13 x 96 and 3 x 160 also leave workgroups partly filled.  "walk2": the
    two-group form with workgroups that walk a frustum in rounds of 256 points and frustum
    after frustum (where a frustum holds whole rounds; else it is fused2 / fused1 again)."""
    from fpn import bayes, plan, synthetic, variables
    forms = ("layers", "fused1", "fused2", "walk2")
    monkeypatch.setenv("FPN_BF16_TAIL_GRID", "3")        # walk2: three workgroups take the frustums in turn
    # Flipout: per-copy streams
    st = variables.VariableStore(device, seed=4)
    bp = bayes.BayesPlan(st, 3, mc_dropout=dropout, precision="bf16")
    pc, oh = synthetic.make_frustums(B, N, 3, seed=B)
    pcd, ohd = dev(pc, device), dev(oh, device)
    g = torch.Generator(device=device)
    g.manual_seed(11)
    noise = bp.sample_noise(2, B, g, N)
    got = {}
    for form in forms:
        for feed in ("ring", "lds"):
            monkeypatch.setenv("FPN_BF16_TAIL", form)
            monkeypatch.setenv("FPN_BF16_FEED", feed)
            ep = bp.forward(pcd, ohd, 2, noise=noise)
            got[form, feed] = {k: ep[k].clone() for k in ("mask_logits", "box_output", "stage1_center")}
    assert float(got["layers", "ring"]["mask_logits"].abs().max()) > 0
    for key, res in got.items():
        for k in res:
            assert torch.equal(res[k], got["layers", "ring"][k]), (key, k)
    # the deterministic network: one shared stream
    st = variables.VariableStore(device, seed=6)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(7)
    net = plan.V1Plan(st, 4, True, precision="bf16")
    pc, oh = synthetic.make_frustums(B, N, 4, seed=B + 1)
    pcd, ohd = dev(pc, device), dev(oh, device)
    got = {}
    for form in forms:
        for feed in ("ring", "lds"):
            monkeypatch.setenv("FPN_BF16_TAIL", form)
            monkeypatch.setenv("FPN_BF16_FEED", feed)
            got[form, feed] = net.forward(pcd, ohd)["mask_logits"].clone()
    for key, res in got.items():
        assert torch.equal(res, got["layers", "ring"]), key


@pytest.mark.parametrize("B,N,C", [(5, 256, 4), (3, 1024, 3), (2, 2048, 4), (8, 512, 8)])
def test_bf16_head_forms_agree_bit_for_bit(device, monkeypatch, B, N, C):
    """The seg head with a workgroup per frustum (points in LDS, rounds over a wrapping weight
    ring, maxima merged in LDS) against the kernel whose waves take one round each: the same
    arithmetic per point, so point_feat and the frustum maxima are the same bits; shared
    stream and one stream per frustum copy.  "walk": two workgroups that each take every
    second frustum (what the library does by itself from two frustums per CU on)."""
    from fpn import bayes, ops, plan, synthetic, variables
    st = variables.VariableStore(device, seed=14)
    plan.init_variables(st, C, True)
    st.randomize_batch_norm(15)
    net = plan.V1Plan(st, C, True, precision="bf16")
    pc, _ = synthetic.make_frustums(B, N, C, seed=B + N)
    x = dev(pc, device).reshape(B * N, C)
    got = {}
    for form in ("round", "frustum", "walk"):
        monkeypatch.setenv("FPN_BF16_HEAD", "frustum" if form == "walk" else form)
        if form == "walk":
            monkeypatch.setenv("FPN_BF16_HEAD_GRID", "2")
        point_feat = torch.zeros((B * N, 64), dtype=torch.bfloat16, device=device)
        gcat = torch.zeros((B, 1024 + 3), device=device)
        ops.v1_seg_head_bf16(x, N, net._head_ws, net._head_consts, point_feat, gcat)
        got[form] = (point_feat, gcat)
    assert float(got["round"][1].abs().max()) > 0 and float(got["round"][0].float().abs().max()) > 0
    for form in ("frustum", "walk"):
        assert torch.equal(got["round"][0], got[form][0]), form
        assert torch.equal(got["round"][1], got[form][1]), form
    if C != 3:
        return
    # (FPN_BF16_HEAD_GRID stays set: the per-copy streams below are walked two workgroups wide as well)
    # Flipout: per-copy streams through the whole plan
    st = variables.VariableStore(device, seed=16)
    bp = bayes.BayesPlan(st, 3, precision="bf16")
    pc, oh = synthetic.make_frustums(B, N, 3, seed=17)
    g = torch.Generator(device=device)
    g.manual_seed(18)
    noise = bp.sample_noise(2, B, g, N)
    res = {}
    for form in ("round", "frustum"):
        monkeypatch.setenv("FPN_BF16_HEAD", form)
        ep = bp.forward(dev(pc, device), dev(oh, device), 2, noise=noise)
        res[form] = {k: ep[k].clone() for k in ("mask_logits", "box_output")}
    for k in res["round"]:
        assert torch.equal(res["round"][k], res["frustum"][k]), k


def test_bf16_forms_the_library_picks_at_scale_agree_with_the_round_forms(device, monkeypatch):
    """With two frustums per CU or more the library itself takes the walking forms of the head,
    the tail and the object-point stacks (one workgroup per CU): the whole v1 network on 640
    frustums of 256 points with no switch set, against the same network with every kernel forced
    to its wave-per-round form - all end points the same bits."""
    from fpn import plan, synthetic, variables
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    B, N, C = max(640, 2 * cus + 64), 256, 4
    st = variables.VariableStore(device, seed=31)
    plan.init_variables(st, C, True)
    st.randomize_batch_norm(32)
    net = plan.V1Plan(st, C, True, precision="bf16")
    pc, oh = synthetic.make_frustums(B, N, C, seed=33)
    pcd, ohd = dev(pc, device), dev(oh, device)
    for k in ("FPN_BF16_HEAD", "FPN_BF16_TAIL", "FPN_BF16_OBJ", "FPN_BF16_HEAD_GRID", "FPN_BF16_TAIL_GRID", "FPN_BF16_OBJ_GRID"):
        monkeypatch.delenv(k, raising=False)
    picked = {k: v.clone() for k, v in net.forward(pcd, ohd).items()}
    monkeypatch.setenv("FPN_BF16_HEAD", "round")
    monkeypatch.setenv("FPN_BF16_TAIL", "fused2")
    monkeypatch.setenv("FPN_BF16_OBJ", "round")
    forced = net.forward(pcd, ohd)
    assert float(picked["mask_logits"].abs().max()) > 0
    for k in forced:
        assert torch.equal(picked[k], forced[k]), k


@pytest.mark.parametrize("B,M", [(5, 512), (7, 256), (2, 1024)])
def test_bf16_object_stack_forms_agree_bit_for_bit(device, monkeypatch, B, M):
    """The T-Net and box stacks over the object points with workgroups that walk the frustums
    (first-layer operands, shifts and ring set up once; three workgroups take the frustums in
    turn) against the kernel whose waves take one round each: the same arithmetic per point, so
    the frustum maxima are the same bits - with and without the centre delta, on top of a
    running maximum in the output."""
    from fpn import ops, plan, variables
    st = variables.VariableStore(device, seed=21)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(22)
    net = plan.V1Plan(st, 4, True, precision="bf16")
    rng = np.random.RandomState(B * M)
    obj = dev(rng.standard_normal((B * M, 3)).astype(np.float32), device)
    delta = dev(rng.standard_normal((B, 3)).astype(np.float32), device)
    monkeypatch.setenv("FPN_BF16_OBJ_GRID", "3")
    for box, ws, consts, width in ((False, net._tnet_ws, net._tnet_consts, 256), (True, net._box_ws, net._box_consts, 512)):
        for d in (None, delta):
            got = {}
            for form in ("round", "walk"):
                monkeypatch.setenv("FPN_BF16_OBJ", form)
                out = torch.zeros((B, width + 3), device=device)
                out[0, :width] = 0.25                    # a running maximum that some channels stay below
                ops.v1_obj_chain_bf16(obj, M, box, d, ws, consts, out)
                got[form] = out
            assert float(got["round"][1:, :width].abs().max()) > 0
            assert torch.equal(got["round"], got["walk"]), (box, d is not None)


@pytest.mark.parametrize("rows,cin,cout,pool,gb", [
    (300, 67, 59, 0, 0), (4096, 448, 128, 0, 0), (1024, 643, 200, 32, 0), (2048, 256, 1024, 64, 0),
    (1536, 128, 256, 128, 0), (2048, 640, 128, 0, 256), (4096, 256, 512, 512, 0), (640, 130, 128, 0, 5),
])
def test_bf16_tiled_layer(device, rows, cin, cout, pool, gb):
    """fpn_shared_mlp_bf16 (the per-layer kernel on bf16 operands): on small-integer data -
    exact in bf16, sums exact in fp32 - it must equal the fp32 kernel bit for bit, for ragged
    rows, K and column counts off the tile sizes, every pooling form and the group bias; on
    random data it stays within the bf16 bound of the float64 product of the rounded operands."""
    from fpn import ops
    rng = np.random.RandomState(rows + cin)
    # row stride off and on a 16-byte multiple (scalar and vector loads of x)
    for kind, ldx in (("int", cin + 3), ("real", cin + 3), ("int", (cin + 3) // 4 * 4 + 4)):
        if kind == "int":
            x = rng.randint(-3, 4, (rows, ldx)).astype(np.float32)
            w = rng.randint(-2, 3, (cin, cout)).astype(np.float32)
            scale = rng.randint(1, 3, cout).astype(np.float32)
            shift = rng.randint(-4, 5, cout).astype(np.float32)
        else:
            x = rng.standard_normal((rows, ldx)).astype(np.float32)
            w = (rng.standard_normal((cin, cout)) * cin ** -0.5).astype(np.float32)
            scale = (rng.random_sample(cout) + 0.5).astype(np.float32)
            shift = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        bias = None
        if gb:
            groups = -(-rows // gb)
            bias = (rng.randint(-3, 4, (groups, cout)) if kind == "int" else rng.standard_normal((groups, cout))).astype(np.float32)
        xd, wd = dev(x, device), dev(w, device)
        kw = dict(relu=True, pool_rows=pool, group_bias=None if bias is None else dev(bias, device), rows_per_group=gb)
        want32 = ops.shared_mlp(xd[:, :cin], wd, dev(scale, device), dev(shift, device), **kw)
        got = ops.shared_mlp(xd[:, :cin], wd, dev(scale, device), dev(shift, device), wt_bf16=ops.pack_mlp_bf16(wd), **kw)
        assert got.shape == want32.shape
        if kind == "int":
            assert torch.equal(got, want32)
        else:
            xb = torch.from_numpy(x[:, :cin]).bfloat16().double().numpy()
            wb = torch.from_numpy(w).bfloat16().double().numpy()
            acc = xb @ wb
            if bias is not None:
                acc = acc + np.repeat(bias.astype(np.float64), gb, axis=0)[:rows]
            v = np.maximum(acc * scale + shift, 0)
            if pool:
                v = v.reshape(rows // pool, pool, cout).max(axis=1)
            assert scaled_err(got.cpu().numpy(), v) <= 1e-4          # fp32 accumulation of exact bf16 products
            assert scaled_err(got.cpu().numpy(), want32.cpu().numpy()) <= BF16_RTOL


@pytest.mark.parametrize("rows,ldx,cin,cout,gb,relu", [
    (131072, 128, 128, 128, 0, True),        # whole blocks, two K tiles
    (131072 + 77, 324, 320, 320, 0, False),  # three column tiles (the last one half full), ragged tail of rows
    (131072 + 128, 320, 320, 128, 1, True),  # a bias row per row
    (140000, 136, 132, 200, 128, True),      # cin off the K tile, a bias row per 128 rows, ragged tail
    (131072 + 96, 64, 64, 64, 3, True),      # any grouping (131072 is not a multiple of 3: the per-tile kernel takes it all)
    (131072 + 96, 68, 68, 72, 2, True),      # any grouping with a tail that starts on a group boundary
])
def test_bf16_streamed_layer_bit_for_bit(device, monkeypatch, rows, ldx, cin, cout, gb, relu):
    """Many rows, few input channels: fpn_shared_mlp_bf16 runs the persistent kernel (the weights resident in LDS,
    one pipeline of x tiles across row blocks) on the whole 128-row blocks and the per-tile kernel on a ragged tail.
    Same operand slots, k order and epilogue arithmetic: it must equal the per-tile kernel (FPN_BF16_MLP=tile) bit
    for bit, and on small-integer data the fp32 kernel too."""
    from fpn import ops
    rng = np.random.RandomState(rows % 1000 + cin)
    for kind in ("int", "real"):
        if kind == "int":
            x = rng.randint(-3, 4, (rows, ldx)).astype(np.float32)
            w = rng.randint(-2, 3, (cin, cout)).astype(np.float32)
            scale = rng.randint(1, 3, cout).astype(np.float32)
            shift = rng.randint(-4, 5, cout).astype(np.float32)
        else:
            x = rng.standard_normal((rows, ldx)).astype(np.float32)
            w = (rng.standard_normal((cin, cout)) * cin ** -0.5).astype(np.float32)
            scale = (rng.random_sample(cout) + 0.5).astype(np.float32)
            shift = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        if ldx > cin:
            x[:, cin:] = np.nan                  # columns past cin are not the layer's: they must not be read into it
        bias = None
        if gb:
            groups = -(-rows // gb)
            bias = (rng.randint(-3, 4, (groups, cout)) if kind == "int" else rng.standard_normal((groups, cout))).astype(np.float32)
        xd, wd = dev(x, device), dev(w, device)
        wt = ops.pack_mlp_bf16(wd)
        kw = dict(relu=relu, group_bias=None if bias is None else dev(bias, device), rows_per_group=gb)
        monkeypatch.setenv("FPN_BF16_MLP", "tile")
        want = ops.shared_mlp(xd[:, :cin], wd, dev(scale, device), dev(shift, device), wt_bf16=wt, **kw)
        monkeypatch.delenv("FPN_BF16_MLP", raising=False)
        got = ops.shared_mlp(xd[:, :cin], wd, dev(scale, device), dev(shift, device), wt_bf16=wt, **kw)
        assert torch.isfinite(got).all()
        assert torch.equal(got, want)
        if kind == "int":
            assert torch.equal(got, ops.shared_mlp(xd[:, :cin], wd, dev(scale, device), dev(shift, device), **kw))


def test_bf16_interpolation_output_feeds_the_head_bit_for_bit(device):
    """Under bf16 the last feature propagation hands its interpolated features to the fused
    head kernel as bf16 (fpn_three_nn_interpolate_bf16 -> fpn_fp_head_chain_bf16x) instead of
    float rows that the head rounds itself: the same logits, bit for bit; and the bf16
    interpolation equals the float one rounded to nearest even."""
    from fpn import pointnet_util as U
    from fpn import synthetic, variables
    from fpn.models import frustum_pointnets_v2 as M2
    from fpn.tf_ops.tf_interpolate import three_nn_interpolate
    rng = np.random.RandomState(3)
    xyz1, xyz2 = rng.random_sample((3, 700, 3)).astype(np.float32), rng.random_sample((3, 50, 3)).astype(np.float32)
    for c in (128, 6):
        pts = rng.standard_normal((3, 50, c)).astype(np.float32)
        f32 = three_nn_interpolate(dev(xyz1, device), dev(xyz2, device), dev(pts, device))
        b16 = three_nn_interpolate(dev(xyz1, device), dev(xyz2, device), dev(pts, device), out_dtype=torch.bfloat16)
        assert torch.equal(b16, f32.bfloat16())
    pc, oh = synthetic.make_frustums(3, 2048, 4, seed=5)
    pc[..., :3] *= 0.05
    st = variables.VariableStore(device, seed=2)
    got = {}
    for flag in (True, False, True):
        # (the head that takes interpolated FEATURES: not the form that interpolates first-layer products)
        U.SA_PRECISION, U.INTERP_BF16, U.HOIST_FEATURES = "bf16", flag, False
        try:
            with variables.use_store(st):
                got[flag] = M2.get_model(dev(pc, device), dev(oh, device), False)["mask_logits"].clone()
        finally:
            U.SA_PRECISION, U.INTERP_BF16, U.HOIST_FEATURES = "fp32", True, True
    assert torch.equal(got[True], got[False])


def test_bf16_packed_features_bit_for_bit(device):
    """Set abstraction in bf16 over wide point features: gathered as float and converted per
    row in the kernel, or converted once into the operand order and gathered as bf16
    (fpn_sa_pack_features_bf16 -> fpn_sa_chain_bf16p): the same bits, for the multi-scale
    module (one table for all scales) and the single-scale one."""
    from fpn import pointnet_util as U
    from fpn import variables
    rng = np.random.RandomState(8)
    b = 3
    xyz = rng.uniform(0, 1, (b, 128, 3)).astype(np.float32)
    cases = [("msg", rng.standard_normal((b, 128, 320)).astype(np.float32),
              (32, [0.4, 0.8, 1.6], [64, 64, 128], [[64, 64, 128], [128, 128, 256], [128, 128, 256]])),
             ("ssg", rng.standard_normal((b, 128, 128)).astype(np.float32), (32, 0.4, 64, [128, 128, 256])),
             # one point group per wave (32 samples), and workgroups that end inside the last ball's rows
             ("ssg", rng.standard_normal((b, 128, 320)).astype(np.float32), (7, 0.5, 32, [128, 128, 256])),
             ("ssg", rng.standard_normal((b, 128, 128)).astype(np.float32), (5, 0.6, 32, [128, 128, 256]))]
    for kind, pts, args in cases:
        st = variables.VariableStore(device, seed=6)
        got = {}
        for flag in (True, False, True):
            # (the kernels that gather the features themselves: not the form that gathers first-layer products)
            U.SA_PRECISION, U.PACK_FEATURES, U.HOIST_FEATURES = "bf16", flag, False
            try:
                with variables.use_store(st):
                    if kind == "msg":
                        _, feat = U.pointnet_sa_module_msg(dev(xyz, device), dev(pts, device), *args, False, None, scope='s')
                    else:
                        _, feat, _ = U.pointnet_sa_module(dev(xyz, device), dev(pts, device), *args, None, False, False, None,
                                                          scope='s')
            finally:
                U.SA_PRECISION, U.PACK_FEATURES, U.HOIST_FEATURES = "fp32", True, True
            got[flag] = feat.clone()
        assert torch.equal(got[True], got[False]), kind
        assert got[True].abs().max() > 0


def test_bf16_persistent_set_abstraction_bit_for_bit(device, monkeypatch):
    """Set abstraction in bf16 when the first layer reads coordinates: the persistent kernel
    (weights resident in LDS, a wave walks whole balls) and the kernel launched per point
    group (FPN_BF16_SA=launch) give the same bits - three stack shapes, one and two point
    groups per wave, one to four rounds per ball, 0 / 1 / 4 feature channels beside the
    coordinates, ball counts that leave waves and whole workgroups without work; and with
    wide features, whose first-layer products both kernels gather."""
    from fpn import pointnet_util as U
    from fpn import variables
    monkeypatch.setattr(U, "POISON_UNCLEARED", True)      # outputs the persistent kernels are trusted to fill start as NaN
    rng = np.random.RandomState(9)
    cases = [("msg", 3, 1024, 1, (64, [0.1, 0.2, 0.4], [32, 64, 128], [[32, 32, 64], [64, 64, 128], [64, 96, 128]])),
             ("ssg", 2, 512, 0, (5, 0.3, 32, [64, 96, 128])),
             ("ssg", 1, 512, 4, (3, 0.3, 128, [64, 128, 128])),
             ("ssg", 5, 700, 1, (129, 0.25, 256, [64, 64, 128])),
             ("ssg", 40, 256, 0, (128, 0.3, 64, [64, 96, 128])),
             # wide features: the kernels gather first-layer products made once per point
             ("msg", 3, 128, 320, (32, [0.4, 0.8, 1.6], [64, 64, 128], [[64, 64, 128], [128, 128, 256], [128, 128, 256]])),
             ("ssg", 2, 128, 128, (7, 0.5, 32, [128, 128, 256])),
             ("ssg", 9, 96, 64, (33, 0.4, 64, [64, 64, 128]))]
    for kind, b, n, cf, args in cases:
        xyz = dev(rng.uniform(0, 1, (b, n, 3)).astype(np.float32), device)
        pts = dev(rng.standard_normal((b, n, cf)).astype(np.float32), device) if cf else None
        st = variables.VariableStore(device, seed=6)
        got = {}
        for mode in ("persistent", "launch", "persistent"):
            if mode == "launch":
                monkeypatch.setenv("FPN_BF16_SA", "launch")
            else:
                monkeypatch.delenv("FPN_BF16_SA", raising=False)
            U.SA_PRECISION = "bf16"
            try:
                with variables.use_store(st):
                    if kind == "msg":
                        _, feat = U.pointnet_sa_module_msg(xyz, pts, *args, False, None, scope='s')
                    else:
                        _, feat, _ = U.pointnet_sa_module(xyz, pts, *args, None, False, False, None, scope='s')
            finally:
                U.SA_PRECISION = "fp32"
            got[mode] = feat.clone()
        assert any(len(k) >= 4 and k[3] == "bf16" for k in st.__dict__["_sa_chain_cache"]), "bf16 kernel not taken"
        assert torch.equal(got["persistent"], got["launch"]), (kind, b, args[0])
        assert got["persistent"].abs().max() > 0


@pytest.mark.parametrize("b,n,cb", [(3, 224, 4), (1, 96, 0), (5, 2048, 7), (2, 160, 8), (300, 64, 1)])
def test_bf16_persistent_head_bit_for_bit(device, monkeypatch, b, n, cb):
    """Feature propagation + per-point head in bf16: the persistent kernel (stream in LDS, a
    wave walks consecutive point groups with the next group's inputs in flight) and the kernel
    launched per group (FPN_BF16_SA=launch) give the same logits bit for bit - row counts that
    are odd multiples of 32 (one group per wave), that leave waves without work, that give
    the waves unequal shares; 0 to 8 own channels."""
    from fpn import pointnet_util as U
    from fpn import variables
    rng = np.random.RandomState(b + cb)
    m = 48
    xyz1 = dev(rng.uniform(0, 1, (b, n, 3)).astype(np.float32), device)
    xyz2 = xyz1[:, :m].contiguous()
    p1 = dev(rng.standard_normal((b, n, cb)).astype(np.float32), device) if cb else None
    p2 = dev(rng.standard_normal((b, m, 128)).astype(np.float32), device)
    st = variables.VariableStore(device, seed=4)
    got = {}
    for mode in ("persistent", "launch", "persistent"):
        if mode == "launch":
            monkeypatch.setenv("FPN_BF16_SA", "launch")
        else:
            monkeypatch.delenv("FPN_BF16_SA", raising=False)
        U.SA_PRECISION, U.HOIST_FEATURES = "bf16", False     # (the head that takes interpolated features)
        try:
            with variables.use_store(st):
                got[mode] = U.pointnet_fp_module_with_head(xyz1, xyz2, p1, p2, [128, 128], 128, 2, False, None, scope='fa',
                                                           head_scope='h1', out_scope='h2').clone()
        finally:
            U.SA_PRECISION, U.HOIST_FEATURES = "fp32", True
    assert len(st.__dict__.get("_fp_head_cache", {})) == 1, "fused kernel not taken"
    assert torch.equal(got["persistent"], got["launch"])
    assert got["persistent"].abs().max() > 0
