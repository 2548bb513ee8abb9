"""GPU parity of the point-set operators: the HIP path (through the C ABI, via the
fpn.tf_ops wrappers) against the CPU oracle on the same seeded inputs.
Bar: indices and copied values bit-exact; interpolated values bit-exact (same
unfused fp32 operation order on both sides)."""
import numpy as np
import pytest
import torch

from oracle import pointset as O

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def clouds(b, n, seed, dup=0.0, scale=1.0):
    rng = np.random.RandomState(seed)
    x = (rng.random_sample((b, n, 3)) * scale).astype(np.float32)
    ndup = int(dup * n)
    for i in range(b):
        if ndup:
            dst = rng.permutation(n)[:ndup]
            src = rng.randint(0, n, size=ndup)
            x[i, dst] = x[i, src]
    return x


@pytest.mark.parametrize("n,npoint", [(1, 1), (1, 4), (64, 16), (100, 100), (129, 32), (512, 128),
                                      (1000, 64), (1024, 128), (2048, 128), (3000, 40), (4096, 33),
                                      (8192, 20), (9000, 12), (128, 32)])
def test_fps_matches_oracle(device, n, npoint):
    from fpn.tf_ops.tf_sampling import farthest_point_sample
    x = clouds(3, n, seed=n + npoint, dup=0.1)
    got = farthest_point_sample(npoint, dev(x, device)).cpu().numpy()
    want = O.farthest_point_sample(npoint, x)
    np.testing.assert_array_equal(got, want)


def test_fps_ties_lowest_index(device):
    from fpn.tf_ops.tf_sampling import farthest_point_sample
    # a lattice: many exactly equal distances; and an all-identical cloud
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(16), indexing="ij"), -1)
    lattice = g.reshape(1, -1, 3).astype(np.float32)
    same = np.ones((1, 1024, 3), np.float32) * 0.25
    for x in (lattice, same, np.concatenate([lattice, lattice[:, ::-1]], 1)):
        got = farthest_point_sample(64, dev(x, device)).cpu().numpy()
        np.testing.assert_array_equal(got, O.farthest_point_sample(64, x))


def test_fps_full_size_properties(device):
    """BASELINE v2 size (2048 -> 128) on a large batch: first index 0, indices in
    range, no repeats while distinct points remain, and equality with the oracle
    on a sampled subset of frustums."""
    from fpn.tf_ops.tf_sampling import farthest_point_sample
    x = clouds(256, 2048, seed=7)
    got = farthest_point_sample(128, dev(x, device)).cpu().numpy()
    assert (got[:, 0] == 0).all() and got.min() >= 0 and got.max() < 2048
    assert all(len(set(r)) == 128 for r in got)
    sub = [0, 17, 100, 255]
    np.testing.assert_array_equal(got[sub], O.farthest_point_sample(128, x[sub]))


@pytest.mark.parametrize("n,npoint", [(257, 32), (300, 64), (512, 128), (513, 65), (1000, 64), (1024, 128), (1025, 1),
                                      (2047, 63), (2048, 128), (2048, 192), (700, 700)])
def test_fps_wave_per_frustum_kernel(device, monkeypatch, n, npoint):
    """The many-frustum form (one wave per frustum; picks stored 64 at a time, so counts
    around the multiples of 64; point counts off the lanes' 8 / 16 / 32-point blocks) against
    the oracle and bit for bit against the workgroup-per-frustum form."""
    from fpn.tf_ops.tf_sampling import farthest_point_sample
    x = clouds(7, n, seed=3 * n + npoint, dup=0.1)     # 7 frustums: a last workgroup with one wave
    xd = dev(x, device)
    monkeypatch.setenv("FPN_FPS", "wave")
    wave = farthest_point_sample(npoint, xd).cpu().numpy()
    monkeypatch.setenv("FPN_FPS", "block")
    block = farthest_point_sample(npoint, xd).cpu().numpy()
    np.testing.assert_array_equal(wave, block)
    np.testing.assert_array_equal(wave, O.farthest_point_sample(npoint, x))


def test_fps_wave_kernel_ties_and_default_dispatch(device, monkeypatch):
    """Exact ties (a lattice, an all-identical cloud, a mirrored lattice, infinities) through
    the wave form; and a batch large enough to take it by default (1536 frustums of 2048
    points) equals the workgroup form on all of them and the oracle on a sample."""
    from fpn.tf_ops.tf_sampling import farthest_point_sample
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(16), indexing="ij"), -1)
    lattice = g.reshape(1, -1, 3).astype(np.float32)
    same = np.ones((1, 1024, 3), np.float32) * 0.25
    far = clouds(1, 600, seed=9)
    far[0, 100] = np.inf                      # an infinite distance never replaces a minimum
    far[0, 200, 1] = np.nan
    monkeypatch.setenv("FPN_FPS", "wave")
    for x in (lattice, same, np.concatenate([lattice, lattice[:, ::-1]], 1)):
        got = farthest_point_sample(64, dev(x, device)).cpu().numpy()
        np.testing.assert_array_equal(got, O.farthest_point_sample(64, x))
    wave = farthest_point_sample(40, dev(far, device)).cpu().numpy()
    monkeypatch.setenv("FPN_FPS", "block")
    np.testing.assert_array_equal(wave, farthest_point_sample(40, dev(far, device)).cpu().numpy())
    x = clouds(1536, 2048, seed=21, dup=0.1)
    xd = dev(x, device)
    block = farthest_point_sample(128, xd).cpu().numpy()
    monkeypatch.delenv("FPN_FPS")
    default = farthest_point_sample(128, xd).cpu().numpy()
    np.testing.assert_array_equal(default, block)
    sub = [0, 511, 1024, 1535]
    np.testing.assert_array_equal(default[sub], O.farthest_point_sample(128, x[sub]))


def test_gather_point(device):
    from fpn.tf_ops.tf_sampling import gather_point
    x = clouds(5, 777, seed=1)
    idx = np.random.RandomState(2).randint(0, 777, size=(5, 300)).astype(np.int32)
    got = gather_point(dev(x, device), dev(idx, device)).cpu().numpy()
    np.testing.assert_array_equal(got, O.gather_point(x, idx))


@pytest.mark.parametrize("n,m,radius,nsample", [(128, 8, 0.3, 32), (512, 128, 0.1, 64), (2048, 128, 0.2, 32),
                                                (2048, 128, 0.4, 64), (2048, 128, 0.8, 128), (128, 32, 1.6, 128),
                                                (5000, 37, 0.15, 16), (9000, 5, 0.05, 8), (64, 64, 0.01, 4)])
def test_query_ball_point(device, n, m, radius, nsample):
    from fpn.tf_ops.tf_grouping import query_ball_point
    x1 = clouds(3, n, seed=n + m, dup=0.05)
    rng = np.random.RandomState(n * 3 + m)
    # half of the queries are dataset points themselves (as inside an SA module)
    x2 = clouds(3, m, seed=n + m + 1)
    pick = rng.randint(0, n, size=(3, m))
    own = rng.random_sample((3, m)) < 0.5
    x2[own] = np.take_along_axis(x1, pick[..., None].repeat(3, 2), 1)[own]
    idx, cnt = query_ball_point(radius, nsample, dev(x1, device), dev(x2, device))
    widx, wcnt = O.query_ball_point(radius, nsample, x1, x2)
    np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), widx)


def test_query_ball_radius_boundary(device):
    """Points at distances straddling the radius by single ulps: the sqrt-free
    accept test must agree with sqrtf(d2) < r everywhere."""
    from fpn.tf_ops.tf_grouping import query_ball_point
    for r in (0.1, 0.2, 0.4, 0.8, 1.6, 0.3, 1e-3, 7.3, 123.456):
        r32 = np.float32(r)
        d = r32 * (1 + np.arange(-600, 600, dtype=np.float32) * np.float32(2.0 ** -24))
        x1 = np.zeros((1, d.size, 3), np.float32)
        x1[0, :, 0] = d
        x1[0, 1::2, 0] *= -1
        x2 = np.zeros((1, 1, 3), np.float32)
        idx, cnt = query_ball_point(float(r32), 2000, dev(x1, device), dev(x2, device))
        widx, wcnt = O.query_ball_point(float(r32), 2000, x1, x2)
        np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
        np.testing.assert_array_equal(idx.cpu().numpy(), widx)
        assert 0 < wcnt[0, 0] < d.size


def test_query_ball_empty(device):
    from fpn.tf_ops.tf_grouping import query_ball_point
    x1 = clouds(2, 100, seed=3)
    x2 = clouds(2, 7, seed=4) + 10.0
    idx, cnt = query_ball_point(0.5, 16, dev(x1, device), dev(x2, device))
    assert (cnt.cpu().numpy() == 0).all() and (idx.cpu().numpy() == 0).all()


@pytest.mark.parametrize("c", [1, 3, 4, 16, 64, 67, 130])
def test_group_point(device, c):
    from fpn.tf_ops.tf_grouping import group_point
    rng = np.random.RandomState(c)
    pts = rng.standard_normal((3, 200, c)).astype(np.float32)
    idx = rng.randint(0, 200, size=(3, 17, 9)).astype(np.int32)
    got = group_point(dev(pts, device), dev(idx, device)).cpu().numpy()
    np.testing.assert_array_equal(got, O.group_point(pts, idx))


@pytest.mark.parametrize("n,m", [(32, 1), (32, 2), (128, 32), (2048, 128), (300, 2500), (1, 3), (257, 5000)])
def test_three_nn(device, n, m):
    from fpn.tf_ops.tf_interpolate import three_nn
    x1 = clouds(2, n, seed=n)
    x2 = clouds(2, m, seed=m + 1, dup=0.2 if m > 10 else 0.0)
    d, i = three_nn(dev(x1, device), dev(x2, device))
    wd, wi = O.three_nn(x1, x2)
    np.testing.assert_array_equal(i.cpu().numpy(), wi)
    np.testing.assert_array_equal(d.cpu().numpy(), wd)


@pytest.mark.parametrize("c", [1, 3, 64, 128, 259])
def test_three_interpolate_and_fused(device, c):
    from fpn.tf_ops.tf_interpolate import three_interpolate, three_nn, three_nn_interpolate
    n, m = 700, 96
    x1 = clouds(2, n, seed=c)
    x2 = clouds(2, m, seed=c + 1)
    p2 = np.random.RandomState(c + 2).standard_normal((2, m, c)).astype(np.float32)
    wd, wi = O.three_nn(x1, x2)
    ww = O.fp_weights(wd)
    want = O.three_interpolate(p2, wi, ww)
    got = three_interpolate(dev(p2, device), dev(wi, device), dev(ww, device)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    out, d, i = three_nn_interpolate(dev(x1, device), dev(x2, device), dev(p2, device), return_neighbours=True)
    np.testing.assert_array_equal(i.cpu().numpy(), wi)
    np.testing.assert_array_equal(d.cpu().numpy(), wd)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    out2 = three_nn_interpolate(dev(x1, device), dev(x2, device), dev(p2, device))
    np.testing.assert_array_equal(out2.cpu().numpy(), want)


def test_fused_interpolate_exact_hit_and_few_known(device):
    """A query that coincides with a known point (distance 0 -> floor 1e-10) and
    m < 3 (+inf distances -> weight 0)."""
    from fpn.tf_ops.tf_interpolate import three_nn_interpolate
    x2 = clouds(1, 2, seed=5)
    x1 = np.concatenate([x2, clouds(1, 30, seed=6)], 1)
    p2 = np.random.RandomState(7).standard_normal((1, 2, 8)).astype(np.float32)
    wd, wi = O.three_nn(x1, x2)
    want = O.three_interpolate(p2, wi, O.fp_weights(wd))
    got = three_nn_interpolate(dev(x1, device), dev(x2, device), dev(p2, device)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert np.isfinite(got).all()


def test_select_top_k(device):
    from fpn.tf_ops.tf_grouping import knn_point, select_top_k
    # the reference's own known-answer case (selection_sort.cpp:65-94)
    dist = (10 - np.arange(16, dtype=np.float32)).reshape(2, 2, 4)
    oi, ov = select_top_k(3, dev(dist, device))
    wi, wv = O.select_top_k(3, dist)
    np.testing.assert_array_equal(oi.cpu().numpy(), wi)
    np.testing.assert_array_equal(ov.cpu().numpy(), wv)
    assert (wi[..., :4] == np.array([3, 2, 1, 0])).all()
    rng = np.random.RandomState(0)
    for (b, m, n, k) in [(2, 5, 100, 8), (1, 3, 1000, 128), (2, 2, 64, 64), (1, 4, 70, 1)]:
        dist = rng.randint(0, 50, size=(b, m, n)).astype(np.float32)   # many ties
        oi, ov = select_top_k(k, dev(dist, device))
        wi, wv = O.select_top_k(k, dist)
        np.testing.assert_array_equal(oi.cpu().numpy(), wi)
        np.testing.assert_array_equal(ov.cpu().numpy(), wv)
    x1 = clouds(2, 50, seed=11)
    x2 = clouds(2, 6, seed=12)
    val, idx = knn_point(4, dev(x1, device), dev(x2, device))
    assert val.shape == (2, 6, 4) and idx.shape == (2, 6, 4)
    d = ((x1[:, None] - x2[:, :, None]) ** 2).sum(-1)
    np.testing.assert_array_equal(idx.cpu().numpy(), np.argsort(d, axis=-1, kind="stable")[..., :4])


def test_mask_gather(device):
    from fpn import ops
    from oracle import v1_numpy as V
    rng = np.random.RandomState(5)
    for (b, n, frac) in [(4, 1024, 0.3), (3, 2048, 0.6), (2, 96, 0.5), (2, 1024, 0.0), (2, 1024, 1.0)]:
        pc = rng.standard_normal((b, n, 4)).astype(np.float32) * 10
        logits = rng.standard_normal((b, n, 2)).astype(np.float32)
        if frac == 0.0:
            logits[..., 1] = logits[..., 0] - 1
        elif frac == 1.0:
            logits[..., 1] = logits[..., 0] + 1
        else:
            logits[..., 1] += np.float32(np.log(frac / (1 - frac)))
        logits[0, :5, 1] = logits[0, :5, 0]      # exact ties are NOT object points (strict <)
        mask, mean, sel, obj = ops.mask_gather(dev(pc, device), dev(logits, device), 512)
        wobj, wmean, wmask, widx = V.point_cloud_masking(pc, logits)
        np.testing.assert_array_equal(mask.cpu().numpy(), wmask)
        np.testing.assert_array_equal(sel.cpu().numpy(), widx)
        np.testing.assert_allclose(mean.cpu().numpy(), wmean, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(obj.cpu().numpy(), wobj, rtol=1e-5, atol=1e-4)
        # the evenly spaced pick: same mask and centroid, another subset when count > 512
        mask2, mean2, sel2, obj2 = ops.mask_gather(dev(pc, device), dev(logits, device), 512, policy="spread")
        wobj, wmean, wmask, widx = V.point_cloud_masking(pc, logits, policy="spread")
        assert torch.equal(mask2, mask) and torch.equal(mean2, mean)
        np.testing.assert_array_equal(sel2.cpu().numpy(), widx)
        np.testing.assert_allclose(obj2.cpu().numpy(), wobj, rtol=1e-5, atol=1e-4)
        many = wmask.sum(1) > 512
        assert np.array_equal(sel2.cpu().numpy()[~many], sel.cpu().numpy()[~many])
        if many.any():
            s2 = sel2.cpu().numpy()[many]
            assert (np.diff(s2, axis=1) > 0).all() and not np.array_equal(s2, sel.cpu().numpy()[many])
            assert (s2[:, -1] > sel.cpu().numpy()[many][:, -1]).all()       # reaches beyond the first 512 positives
    with pytest.raises(ValueError):
        ops.mask_gather(dev(pc, device), dev(logits, device), 512, policy="random")


def test_argument_errors(device):
    from fpn.tf_ops import tf_grouping, tf_interpolate, tf_sampling
    x = dev(clouds(1, 8, seed=0), device)
    with pytest.raises(ValueError):
        tf_sampling.farthest_point_sample(0, x)
    with pytest.raises(ValueError):
        tf_grouping.query_ball_point(-1.0, 4, x, x)
    with pytest.raises(ValueError):
        tf_grouping.query_ball_point(0.1, 0, x, x)
    with pytest.raises(ValueError):
        tf_interpolate.three_nn(x, x[..., :2].contiguous())
    with pytest.raises(ValueError):
        tf_sampling.farthest_point_sample(4, x.cpu())
    with pytest.raises(ValueError):
        tf_sampling.gather_point(x, torch.zeros((1, 2), dtype=torch.int64, device=device))


def test_query_group_interpolate_full_size_properties(device):
    """BASELINE v2 size (256 frustums x 2048 points, 128 centroids, 64 samples) through
    properties that need no oracle run: every returned neighbour lies in the ball, the
    slots are in index order and padded with the first hit, the counts are right, grouping
    is a pure gather, the three neighbours are the three nearest in ascending order, and
    interpolation weights sum to one; plus oracle equality on a few frustums."""
    from fpn.tf_ops.tf_grouping import group_point, query_ball_point
    from fpn.tf_ops.tf_interpolate import three_nn, three_nn_interpolate
    B, N, M, NS, C, r = 256, 2048, 128, 64, 16, 0.2
    x = clouds(B, N, seed=11, dup=0.1)
    xyz = dev(x, device)
    new_xyz = xyz[:, :M].contiguous()
    feat = dev(np.random.RandomState(12).standard_normal((B, N, C)).astype(np.float32), device)
    idx, cnt = query_ball_point(r, NS, xyz, new_xyz)
    assert idx.dtype == torch.int32 and int(idx.min()) >= 0 and int(idx.max()) < N
    li = idx.long()
    nb = torch.gather(xyz[:, None].expand(B, M, N, 3), 2, li[..., None].expand(B, M, NS, 3))
    d = (nb - new_xyz[:, :, None]).pow(2).sum(-1).sqrt()
    slot = torch.arange(NS, device=device)[None, None]
    valid = slot < cnt[..., None]
    assert bool((d[valid] < r * (1 + 1e-5)).all())
    assert bool(((li[..., 1:] > li[..., :-1]) | ~valid[..., 1:]).all())        # index order
    assert bool((li == torch.where(valid, li, li[..., :1])).all())               # padding = first hit
    dist_all = (new_xyz[:, :, None] - xyz[:, None]).pow(2).sum(-1).sqrt()       # (B,M,N), explicit differences
    lo = (dist_all < r * (1 - 1e-5)).sum(-1).clamp(max=NS)
    hi = (dist_all < r * (1 + 1e-5)).sum(-1).clamp(max=NS)
    assert bool(((cnt >= lo) & (cnt <= hi)).all()) and int(cnt.min()) >= 1     # a centroid is in its own ball
    grouped = group_point(feat, idx)
    assert torch.equal(grouped, torch.gather(feat[:, None].expand(B, M, N, C), 2, li[..., None].expand(B, M, NS, C)))

    dist, nn = three_nn(xyz, new_xyz)                                            # unknown: all points, known: centroids
    assert bool((dist[..., 1:] >= dist[..., :-1]).all()) and int(nn.min()) >= 0 and int(nn.max()) < M
    known = torch.gather(new_xyz[:, None].expand(B, N, M, 3), 2, nn.long()[..., None].expand(B, N, 3, 3))
    assert torch.allclose((known - xyz[:, :, None]).pow(2).sum(-1), dist, rtol=1e-5, atol=1e-9)
    third = dist_all.transpose(1, 2).pow(2).kthvalue(3, dim=-1).values
    assert torch.allclose(third, dist[..., 2], rtol=1e-4, atol=1e-7)
    const = torch.full((B, M, 4), 3.25, device=device)
    assert torch.allclose(three_nn_interpolate(xyz, new_xyz, const), torch.full((B, N, 4), 3.25, device=device), rtol=1e-6)

    sub = [0, 101, 255]
    oi, oc = O.query_ball_point(r, NS, x[sub], x[sub][:, :M])
    np.testing.assert_array_equal(idx[sub].cpu().numpy(), oi)
    np.testing.assert_array_equal(cnt[sub].cpu().numpy(), oc)
    od, on = O.three_nn(x[sub], x[sub][:, :M])
    np.testing.assert_array_equal(nn[sub].cpu().numpy(), on)
    np.testing.assert_array_equal(dist[sub].cpu().numpy(), od)


def test_zero_sized_inputs(device):
    """Empty batches and empty query / sample sets: every operator returns an empty result
    of the right shape without launching anything."""
    from fpn import ops
    from fpn.tf_ops.tf_grouping import group_point, query_ball_point
    from fpn.tf_ops.tf_interpolate import three_interpolate, three_nn, three_nn_interpolate
    from fpn.tf_ops.tf_sampling import farthest_point_sample, gather_point
    f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=device)
    i = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=device)
    assert farthest_point_sample(4, f(0, 16, 3)).shape == (0, 4)
    assert gather_point(f(0, 16, 3), i(0, 4)).shape == (0, 4, 3)
    assert gather_point(f(2, 16, 3), i(2, 0)).shape == (2, 0, 3)
    idx, cnt = query_ball_point(0.5, 8, f(2, 16, 3), f(2, 0, 3))
    assert idx.shape == (2, 0, 8) and cnt.shape == (2, 0)
    idx, cnt = query_ball_point(0.5, 8, f(0, 16, 3), f(0, 5, 3))
    assert idx.shape == (0, 5, 8) and cnt.shape == (0, 5)
    assert group_point(f(2, 16, 7), i(2, 0, 8)).shape == (2, 0, 8, 7)
    d, ix = three_nn(f(2, 0, 3), f(2, 5, 3))
    assert d.shape == (2, 0, 3) and ix.shape == (2, 0, 3)
    assert three_interpolate(f(2, 5, 7), i(2, 0, 3), f(2, 0, 3)).shape == (2, 0, 7)
    assert three_nn_interpolate(f(0, 9, 3), f(0, 5, 3), f(0, 5, 7)).shape == (0, 9, 7)
    assert ops.shared_mlp(f(0, 16), f(16, 32), None, None, relu=True).shape == (0, 32)
    torch.cuda.synchronize()


def test_zero_frustum_batch_through_the_models(device):
    """A batch of zero frustums goes through the v1 plan (every kernel entry point sees an
    empty problem) and comes back as empty end points."""
    from fpn import plan, variables
    st = variables.VariableStore(device, seed=1)
    plan.init_variables(st, 4, True)
    net = plan.V1Plan(st, 4, True)
    ep = net.forward(torch.zeros((0, 1024, 4), device=device), torch.zeros((0, 3), device=device))
    assert ep["mask_logits"].shape == (0, 1024, 2) and ep["box_output"].shape == (0, 59) and ep["center"].shape == (0, 3)


@pytest.mark.parametrize("kind", ["v1_bf16", "v2", "bayes"])
def test_zero_frustum_batch_other_models(device, kind):
    from fpn import bayes, plan, variables
    st = variables.VariableStore(device, seed=1)
    pc, oh = torch.zeros((0, 256, 4), device=device), torch.zeros((0, 3), device=device)
    if kind == "v1_bf16":
        plan.init_variables(st, 4, True)
        ep = plan.V1Plan(st, 4, True, precision="bf16").forward(pc, oh)
        assert ep["mask_logits"].shape == (0, 256, 2) and ep["box_output"].shape == (0, 59)
    elif kind == "v2":
        from fpn.models import frustum_pointnets_v2 as M2
        with variables.use_store(st):
            M2.get_model(dev(np.random.RandomState(13).uniform(0, 1, (2, 256, 4)).astype(np.float32), device),
                         torch.eye(3, device=device)[:2], False)          # creates the variables
            ep = M2.get_model(pc, oh, False)
        assert ep["mask_logits"].shape == (0, 256, 2) and ep["box_output"].shape == (0, 59)
    else:
        bp = bayes.BayesPlan(st, 3)
        ep = bp.forward(pc[:, :, :3].contiguous(), oh, 2)
        assert ep["mask_logits"].shape == (2, 0, 256, 2) and ep["box_output"].shape == (2, 0, 59)


def test_pointset_random_shapes(device):
    """Forty seeded random problems (batch, point counts, radius, samples per ball, channel
    counts, duplicate fractions drawn at random, including single points and more samples
    than points): every operator bit-exact against the oracle."""
    from fpn.tf_ops.tf_grouping import group_point, knn_point, query_ball_point
    from fpn.tf_ops.tf_interpolate import three_interpolate, three_nn, three_nn_interpolate
    from fpn.tf_ops.tf_sampling import farthest_point_sample, gather_point
    rng = np.random.RandomState(2024)
    for trial in range(40):
        b = int(rng.choice([1, 2, 3, 5, 8, 9]))
        n = int(rng.choice([1, 2, 7, 33, 64, 100, 255, 256, 257, 700, 1500]))
        m = int(rng.randint(1, min(n, 200) + 1))
        ns = int(rng.choice([1, 3, 8, 32, 50, 64, 130]))
        c = int(rng.choice([1, 2, 4, 5, 12, 32, 60, 64, 100]))
        r = float(rng.choice([0.05, 0.2, 0.5, 2.0]))
        x = clouds(b, n, seed=trial, dup=float(rng.choice([0.0, 0.3])))
        tag = (trial, b, n, m, ns, c, r)
        idx = farthest_point_sample(m, dev(x, device))
        widx = O.farthest_point_sample(m, x)
        np.testing.assert_array_equal(idx.cpu().numpy(), widx, err_msg=str(tag))
        q = gather_point(dev(x, device), idx)
        wq = O.gather_point(x, widx)
        np.testing.assert_array_equal(q.cpu().numpy(), wq, err_msg=str(tag))
        gi, cnt = query_ball_point(r, ns, dev(x, device), q)
        wgi, wcnt = O.query_ball_point(r, ns, x, wq)
        np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt, err_msg=str(tag))
        np.testing.assert_array_equal(gi.cpu().numpy(), wgi, err_msg=str(tag))
        feat = rng.standard_normal((b, n, c)).astype(np.float32)
        np.testing.assert_array_equal(group_point(dev(feat, device), gi).cpu().numpy(), O.group_point(feat, wgi),
                                      err_msg=str(tag))
        d, i3 = three_nn(dev(x, device), q)
        wd, wi3 = O.three_nn(x, wq)
        np.testing.assert_array_equal(i3.cpu().numpy(), wi3, err_msg=str(tag))
        np.testing.assert_array_equal(d.cpu().numpy(), wd, err_msg=str(tag))
        f2 = rng.standard_normal((b, m, c)).astype(np.float32)
        ww = O.fp_weights(wd)
        want = O.three_interpolate(f2, wi3, ww)
        np.testing.assert_array_equal(three_interpolate(dev(f2, device), i3, dev(ww, device)).cpu().numpy(), want,
                                      err_msg=str(tag))
        np.testing.assert_array_equal(three_nn_interpolate(dev(x, device), q, dev(f2, device)).cpu().numpy(), want,
                                      err_msg=str(tag))
        k = int(min(n, rng.choice([1, 3, 16])))
        val, ki = knn_point(k, dev(x, device), q)
        assert ki.shape == (b, m, k) and val.shape == (b, m, k), tag
        # the k smallest squared distances, ascending (ties and rounding aside: compared as values)
        d2 = ((wq[:, :, None, :].astype(np.float64) - x[:, None, :, :]) ** 2).sum(-1)
        np.testing.assert_allclose(val.cpu().numpy(), np.sort(d2, -1)[..., :k], rtol=1e-5, atol=1e-6, err_msg=str(tag))


@pytest.mark.parametrize("b,n,m,c,k", [(3, 2048, 37, 3, 32), (2, 1000, 5, 3, 128), (1, 64, 64, 3, 64), (2, 130, 9, 5, 7),
                                       (4, 1, 3, 3, 1), (2, 4096, 2, 4, 3)])
def test_knn_point_fused_kernel(device, b, n, m, c, k):
    """fpn_knn_point against the oracle (float32 matrix + selection sort) and bit for bit
    against the reference's composition on the device (tensor operations + select_top_k);
    duplicated points give exact ties."""
    from fpn.tf_ops import tf_grouping
    rng = np.random.RandomState(n + m + k)
    x1 = rng.random_sample((b, n, c)).astype(np.float32)
    if n > 8:
        x1[:, rng.permutation(n)[:n // 8]] = x1[:, rng.randint(0, n, n // 8)]
    x2 = rng.random_sample((b, m, c)).astype(np.float32)
    x2[:, ::2] = x1[:, rng.randint(0, n, len(range(0, m, 2)))]
    val, idx = tf_grouping.knn_point(k, dev(x1, device), dev(x2, device))
    wval, widx = O.knn_point(k, x1, x2)
    np.testing.assert_array_equal(idx.cpu().numpy(), widx)
    np.testing.assert_array_equal(val.cpu().numpy(), wval)
    tf_grouping.KNN_FUSED = False
    try:
        mval, midx = tf_grouping.knn_point(k, dev(x1, device), dev(x2, device))
    finally:
        tf_grouping.KNN_FUSED = True
    assert torch.equal(midx, idx) and torch.equal(mval, val)


def test_knn_point_argument_checks(device):
    from fpn.tf_ops.tf_grouping import knn_point
    x1 = torch.zeros((2, 10, 3), device=device)
    with pytest.raises(ValueError):
        knn_point(0, x1, x1)
    with pytest.raises(ValueError):
        knn_point(11, x1, x1)
    with pytest.raises(ValueError):
        knn_point(2, x1, torch.zeros((3, 4, 3), device=device))
    with pytest.raises(ValueError):
        knn_point(2, x1, torch.zeros((2, 4, 2), device=device))
    val, idx = knn_point(2, torch.zeros((0, 10, 3), device=device), torch.zeros((0, 4, 3), device=device))
    assert val.shape == (0, 4, 2) and idx.shape == (0, 4, 2)


def test_query_ball_point_multi_equals_single_queries(device):
    """One sweep for several radii (fpn_query_ball_point_multi) gives each radius the index
    table and the counts of its own fpn_query_ball_point call, bit for bit: full balls (early
    stop), sparse and empty balls, point counts off the step and tile sizes, five radii (two
    launches), counts skipped."""
    from fpn.tf_ops.tf_grouping import query_ball_point, query_ball_point_multi
    rng = np.random.RandomState(2)
    for b, n, m in ((3, 2048, 128), (2, 517, 33), (1, 5000, 9), (4, 128, 32)):
        xyz = rng.random_sample((b, n, 3)).astype(np.float32)
        new = xyz[:, rng.permutation(n)[:m]] + (rng.random_sample((b, m, 3)).astype(np.float32) - 0.5) * 0.01
        xyz_d, new_d = torch.from_numpy(xyz).to(device), torch.from_numpy(np.ascontiguousarray(new)).to(device)
        radii, ns = [0.02, 0.1, 0.2, 0.4, 0.9], [8, 32, 64, 128, 16]
        many = query_ball_point_multi(radii, ns, xyz_d, new_d)
        nocnt = query_ball_point_multi(radii[:3], ns[:3], xyz_d, new_d, counts=False)
        for i, (r, k) in enumerate(zip(radii, ns)):
            idx, cnt = query_ball_point(r, k, xyz_d, new_d)
            assert torch.equal(many[i][0], idx), (n, r)
            assert torch.equal(many[i][1], cnt), (n, r)
            if i < 3:
                assert nocnt[i][1] is None and torch.equal(nocnt[i][0], idx)
