"""GPU parity of the shared-MLP layer and of the whole v1 network against the numpy
oracle.

Tolerance (floating point): the fp32 MFMA path and the numpy fp32 GEMM differ only
by summation order; the bar is |got - want| <= 1e-3 * max(|want|, s) with s the
tensor's RMS (the north-star 1e-3 relative bound, made robust to outputs that pass
through zero).  In practice the error is ~1e-6 relative; a float64 evaluation of the
oracle is used to show both fp32 results are equally close to the exact value.
"""
import numpy as np
import pytest
import torch

from oracle import v1_numpy as V

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def close(got, want, rtol=RTOL):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    s = np.sqrt(np.mean(want ** 2)) + 1e-30
    err = np.abs(got - want) / np.maximum(np.abs(want), s)
    assert err.max() <= rtol, "max scaled error %.3e" % err.max()
    return err.max()


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("rows,cin,cout", [(256, 4, 64), (1000, 64, 64), (384, 64, 128), (512, 128, 1024),
                                           (130, 1091, 512), (64, 515, 512), (777, 3, 128), (256, 128, 2),
                                           (33, 256, 59), (128, 512, 256), (5, 259, 256)])
@pytest.mark.parametrize("relu", [True, False])
def test_layer_matches_numpy(device, rows, cin, cout, relu):
    from fpn import ops
    rng = np.random.RandomState(rows + cin + cout)
    x = rng.standard_normal((rows, cin)).astype(np.float32)
    w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32) * rng.choice([-1, 1], cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    want = (x.astype(np.float64) @ w.astype(np.float64)) * scale + shift
    if relu:
        want = np.maximum(want, 0)
    got = ops.shared_mlp(dev(x, device), dev(w, device), dev(scale, device), dev(shift, device), relu=relu)
    close(got.cpu().numpy(), want, 1e-4)


def test_few_row_kernel_against_tiled_kernel(device):
    """Launches with few rows split K over the four waves of a 32 x 64 tile (partial sums
    added in wave order), larger ones run the LDS-tiled workgroups: the two differ by the
    summation order only (1e-5 of the output's scale), the few-row result of a row does not
    depend on the batch it arrives in, and repeating a launch reproduces it bit for bit
    (plain layer with a per-group bias, a narrow output, and the two-product Flipout layer
    with one sign group per row)."""
    from fpn import ops
    rng = np.random.RandomState(12)
    rows, few, cin, cout = 8192, 77, 515, 512          # 8192 x 512: 2048 tiles of 32 x 64 = tiled path
    x = dev(rng.standard_normal((rows, cin)).astype(np.float32), device)
    w = dev((rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32), device)
    scale = dev(rng.uniform(0.5, 1.5, cout).astype(np.float32), device)
    shift = dev(rng.standard_normal(cout).astype(np.float32), device)
    gb = dev(rng.standard_normal((rows, cout)).astype(np.float32), device)

    def check(big, small, fewer):
        assert float(big.abs().max()) > 0
        close(small.cpu().numpy(), big[:few].cpu().numpy(), 1e-5)
        assert torch.equal(small[:50], fewer)
    run = lambda r: ops.shared_mlp(x[:r].contiguous(), w, scale, shift, relu=True, group_bias=gb[:r].contiguous(),     # noqa: E731
                                   rows_per_group=1)
    small = run(few)
    check(run(rows), small, run(50))
    assert torch.equal(small, run(few))
    # a narrow output (fc3's 59 columns)
    w59 = w[:, :59].contiguous()
    run = lambda xx: ops.shared_mlp(xx, w59, None, shift[:59].contiguous(), relu=False)      # noqa: E731
    check(run(torch.cat([x] * 8)), run(x[:few].contiguous()), run(x[:50].contiguous()))
    dw = dev((rng.standard_normal((1, cin, cout)) * 0.05).astype(np.float32), device)
    si = dev(rng.choice([-1.0, 1.0], (rows, cin)).astype(np.float32), device)
    so = dev(rng.choice([-1.0, 1.0], (rows, cout)).astype(np.float32), device)
    run = lambda r: ops.shared_mlp_flipout(x[:r].contiguous(), w, dw, r, si[:r].contiguous(), so[:r].contiguous(), 1,     # noqa: E731
                                           scale=None, shift=shift, relu=True)
    small = run(few)
    check(run(rows), small, run(50))
    want = x[:few].double() @ w.double() + ((x[:few] * si[:few]).double() @ dw[0].double()) * so[:few].double() + shift.double()
    close(small.cpu().numpy(), np.maximum(want.cpu().numpy(), 0), 1e-4)


@pytest.mark.parametrize("pool", [32, 64, 128, 256, 1024])
def test_layer_pool_and_group_bias(device, pool):
    from fpn import ops
    rng = np.random.RandomState(pool)
    groups, cin, cout = 5, 64, 200
    rows = groups * pool
    x = rng.standard_normal((rows, cin)).astype(np.float32)
    w = (rng.standard_normal((cin, cout)) / 8).astype(np.float32)
    gb = rng.standard_normal((groups, cout)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    full = (x.astype(np.float64) @ w + np.repeat(gb, pool, 0)) * scale + shift
    want = full.reshape(groups, pool, cout).max(1)
    # written into a slice of a wider buffer (the concat-buffer use of the models)
    wide = torch.zeros((groups, cout + 3), device=device)
    ops.shared_mlp(dev(x, device), dev(w, device), dev(scale, device), dev(shift, device), relu=False,
                   pool_rows=pool, group_bias=dev(gb, device), rows_per_group=pool, out=wide[:, :cout])
    close(wide[:, :cout].cpu().numpy(), want, 1e-4)
    assert (wide[:, cout:] == 0).all()
    got = ops.max_pool_rows(dev(full.astype(np.float32), device), pool)
    np.testing.assert_array_equal(got.cpu().numpy(), full.astype(np.float32).reshape(groups, pool, cout).max(1))


def make_store(device, pc_channels=4, bn=True, seed=1):
    from fpn import plan, variables
    st = variables.VariableStore(device, seed=seed)
    plan.init_variables(st, pc_channels, bn)
    st.randomize_batch_norm(seed + 1)
    return st


def compare_end_points(ep, want, pc, one_hot, params, bn=True):
    """Staged parity: logits; mask (flips allowed only where the two logits agree
    to within the tolerance band); everything after the mask against the oracle
    re-run on the GPU's mask, so a borderline point cannot hide a real error."""
    logits = ep["mask_logits"].cpu().numpy()
    close(logits, want["mask_logits"])
    mask = ep["mask"].cpu().numpy()
    flips = mask != want["mask"]
    if flips.any():
        wl = want["mask_logits"]
        gap = np.abs(wl[..., 0] - wl[..., 1])[flips]
        assert gap.max() <= RTOL * np.sqrt(np.mean(wl ** 2)), "mask differs away from the decision boundary"
        want = V.get_model(pc, one_hot, params, bn=bn, mask_override=mask)
    np.testing.assert_array_equal(ep["object_point_indices"].cpu().numpy(), want["object_point_indices"])
    for k in ["mask_xyz_mean", "center_delta", "stage1_center", "box_output", "center_boxnet", "heading_scores",
              "heading_residuals_normalized", "heading_residuals", "size_scores",
              "size_residuals_normalized", "size_residuals", "center"]:
        close(ep[k].cpu().numpy(), want[k])
    return int(flips.sum())


@pytest.mark.parametrize("literal", [False, True])
def test_v1_get_model_config0(device, literal):
    """BASELINE config 0: batch 32 x 1024 points, v1, against the CPU oracle."""
    from fpn import synthetic, variables
    from fpn.models import frustum_pointnets_v1 as M
    pc, one_hot = synthetic.make_frustums(32, 1024, 4, seed=0)
    st = make_store(device)
    with variables.use_store(st):
        ep = M.get_model(dev(pc, device), dev(one_hot, device), is_training=False, literal=literal)
    params = st.numpy()
    want = V.get_model(pc, one_hot, params)
    assert ep["mask_logits"].shape == (32, 1024, 2) and ep["size_residuals"].shape == (32, 8, 3)
    compare_end_points(ep, want, pc, one_hot, params)
    # the segmentation must be non-trivial for the test to mean anything
    cnt = want["mask"].sum(1)
    assert (cnt > 0).any() and (cnt < 1024).any()


def test_v1_chunking_and_sizes(device):
    """Chunk boundaries do not change results; N=2048 and a ragged batch work."""
    from fpn import plan, synthetic
    st = make_store(device)
    pc, one_hot = synthetic.make_frustums(7, 2048, 4, seed=3)
    a = plan.V1Plan(st, 4, True, chunk_frustums=3).forward(dev(pc, device), dev(one_hot, device))
    b = plan.V1Plan(st, 4, True, chunk_frustums=7).forward(dev(pc, device), dev(one_hot, device))
    for k in a:
        assert torch.equal(a[k], b[k]), k
    want = V.get_model(pc, one_hot, st.numpy())
    compare_end_points(a, want, pc, one_hot, st.numpy())


@pytest.mark.parametrize("b,n,chain", [(1, 1024, True), (1, 32, True), (2, 4096, True), (1, 96, False), (1, 160, True)])
def test_v1_single_frustums_and_extreme_point_counts(device, b, n, chain):
    """One frustum per call (grids of a handful of workgroups), the smallest point count a
    wave can carry (32), a large one (4096), counts that are no multiple of 128."""
    from fpn import plan, synthetic
    st = make_store(device, seed=4)
    pc, one_hot = synthetic.make_frustums(b, n, 4, seed=n)
    ep = plan.V1Plan(st, 4, True, chain=chain).forward(dev(pc, device), dev(one_hot, device))
    compare_end_points(ep, V.get_model(pc, one_hot, st.numpy()), pc, one_hot, st.numpy())
    with pytest.raises(ValueError):
        plan.V1Plan(st, 4, True).forward(dev(pc[:, :n - 1], device), dev(one_hot, device))


def test_v1_oracle_fp64_error_budget(device):
    """Both fp32 evaluations (GPU, numpy) sit equally close to the float64 value."""
    from fpn import plan, synthetic
    st = make_store(device, seed=5)
    pc, one_hot = synthetic.make_frustums(4, 1024, 4, seed=9)
    ep = plan.V1Plan(st, 4, True).forward(dev(pc, device), dev(one_hot, device))
    exact = V.get_model(pc, one_hot, st.numpy(), dtype=np.float64)
    cpu32 = V.get_model(pc, one_hot, st.numpy(), dtype=np.float32)
    e_gpu = close(ep["mask_logits"].cpu().numpy(), exact["mask_logits"], 1e-3)
    e_cpu = close(cpu32["mask_logits"], exact["mask_logits"], 1e-3)
    assert e_gpu < 1e-4 and e_cpu < 1e-4


def test_chain_kernels_match_layer_path(device):
    """The register-chain kernels against the one-layer-per-launch path (same weights,
    same inputs): logits within the fp32 summation-order band.  96 and 160 points per
    frustum give wave counts that are not a multiple of the workgroup's four waves
    (surplus waves leave early) and frustum boundaries inside a workgroup."""
    from fpn import plan, synthetic
    st = make_store(device, seed=11)
    for b_, n in ((5, 1024), (5, 2048), (5, 128), (3, 96), (5, 160), (1, 32)):
        pc, one_hot = synthetic.make_frustums(b_, n, 4, seed=n)
        a = plan.V1Plan(st, 4, True, chain=True).forward(dev(pc, device), dev(one_hot, device))
        if n % 128 == 0 or n in (32, 64):     # group sizes the per-layer kernel can max-pool in its epilogue
            b = plan.V1Plan(st, 4, True, chain=False).forward(dev(pc, device), dev(one_hot, device))
            close(a["mask_logits"].cpu().numpy(), b["mask_logits"].cpu().numpy(), 1e-4)
        want = V.get_model(pc, one_hot, st.numpy())
        compare_end_points(a, want, pc, one_hot, st.numpy())


def test_hip_graph_replay_matches_eager(device):
    """A v1 forward captured into a HIP graph (fpn.graph.GraphedForward) and replayed on
    new inputs gives the eager results bit for bit."""
    from fpn import graph, plan, synthetic
    st = make_store(device, seed=21)
    net = plan.V1Plan(st, 4, True)
    pc0, oh0 = synthetic.make_frustums(6, 256, 4, seed=1)
    pc1, oh1 = synthetic.make_frustums(6, 256, 4, seed=2)
    keys = ("mask_logits", "box_output", "center")
    g = graph.GraphedForward(lambda a, b: {k: v for k, v in net.forward(a, b).items() if k in keys},
                             [dev(pc0, device), dev(oh0, device)])
    for pc, oh in ((pc1, oh1), (pc0, oh0)):
        got = {k: v.clone() for k, v in g(dev(pc, device), dev(oh, device)).items()}
        want = net.forward(dev(pc, device), dev(oh, device))
        for k in keys:
            assert torch.equal(got[k], want[k]), k


def test_v1_full_size_properties(device):
    """BASELINE config 1 size (1024 frustums x 1024 points): frustums are independent, so
    a frustum's results do not depend on what else is in the batch (bit for bit), and a
    sample of frustums matches the CPU oracle."""
    from fpn import plan, synthetic
    st = make_store(device, seed=31)
    net = plan.V1Plan(st, 4, True)
    pc, one_hot = synthetic.make_frustums(1024, 1024, 4, seed=5)
    full = net.forward(dev(pc, device), dev(one_hot, device))
    sub = [0, 17, 500, 1023]
    alone = net.forward(dev(pc[sub], device), dev(one_hot[sub], device))
    for k in ("mask_logits", "box_output", "center", "stage1_center", "mask"):
        assert torch.equal(full[k][sub], alone[k]), k
    assert torch.isfinite(full["box_output"]).all() and torch.isfinite(full["mask_logits"]).all()
    want = V.get_model(pc[sub], one_hot[sub], st.numpy())
    compare_end_points(alone, want, pc[sub], one_hot[sub], st.numpy())


def test_two_streams_and_two_threads(device):
    """The library keeps no state between calls: forwards enqueued on two HIP streams from
    two Python threads at the same time give the results of the same forwards run alone."""
    import threading
    from fpn import plan, synthetic
    st = make_store(device, seed=2)
    net = plan.V1Plan(st, 4, True)
    inputs = [tuple(dev(a, device) for a in synthetic.make_frustums(8, 512, 4, seed=s)) for s in (1, 2)]
    alone = [net.forward(pc, oh) for pc, oh in inputs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=device) for _ in inputs]
    results, errors = [None, None], []

    def work(i):
        try:
            with torch.cuda.device(device), torch.cuda.stream(streams[i]):
                out = None
                for _ in range(5):
                    out = net.forward(*inputs[i])
                streams[i].synchronize()
                results[i] = out
        except Exception as e:            # surfaced below: an exception in a thread would otherwise vanish
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for want, got in zip(alone, results):
        for k in ("mask_logits", "box_output", "object_point_indices", "center"):
            assert torch.equal(want[k], got[k]), k


def test_layer_random_shapes_and_options(device):
    """Thirty seeded random layers: odd row / channel counts, optional scale, shift, ReLU,
    per-group bias, fused pooling, input and output as column slices of wider buffers."""
    from fpn import ops
    rng = np.random.RandomState(99)
    for trial in range(30):
        pool = int(rng.choice([0, 0, 32, 64, 128, 256]))
        groups = int(rng.randint(1, 12))
        rows = groups * pool if pool else int(rng.choice([1, 5, 31, 32, 33, 100, 129, 1000, 2500]))
        cin = int(rng.choice([1, 3, 4, 31, 64, 100, 259, 515, 1027]))
        cout = int(rng.choice([1, 2, 3, 32, 59, 64, 65, 128, 200, 512, 1024]))
        relu = bool(rng.randint(2))
        x_wide = rng.standard_normal((rows, cin + 3)).astype(np.float32)
        w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, cout).astype(np.float32) if rng.randint(2) else None
        shift = rng.standard_normal(cout).astype(np.float32) if rng.randint(2) else None
        use_gb = bool(rng.randint(2))
        rpg = pool if pool else int(rng.choice([1, rows]))
        ngrp = rows // rpg if use_gb else 0
        gb = rng.standard_normal((ngrp, cout)).astype(np.float32) if use_gb else None
        x_dev = dev(x_wide, device)[:, 1:1 + cin]                      # a column slice: row stride cin + 3
        y = x_wide[:, 1:1 + cin].astype(np.float64) @ w
        if use_gb:
            y = y + np.repeat(gb, rpg, 0)
        y = y * (1.0 if scale is None else scale) + (0.0 if shift is None else shift)
        if relu:
            y = np.maximum(y, 0)
        if pool:
            y = y.reshape(groups, pool, cout).max(1)
        wide = torch.full((y.shape[0], cout + 2), 7.0, device=device)
        ops.shared_mlp(x_dev, dev(w, device), None if scale is None else dev(scale, device),
                       None if shift is None else dev(shift, device), relu=relu, pool_rows=pool,
                       group_bias=None if gb is None else dev(gb, device), rows_per_group=rpg if use_gb else 0,
                       out=wide[:, 1:1 + cout])
        tag = (trial, rows, cin, cout, pool, relu, use_gb)
        got = wide.cpu().numpy()
        assert (got[:, 0] == 7).all() and (got[:, -1] == 7).all(), tag            # nothing written outside the slice
        err = np.abs(got[:, 1:1 + cout] - y) / np.maximum(np.abs(y), np.sqrt(np.mean(y ** 2)) + 1e-30)
        assert err.max() <= 1e-4, (tag, err.max())
