"""GPU parity of the PointNet++ modules and of the v2 topology (BASELINE config 2:
2048 points, FPS + ball query + group + three_interpolate) against the numpy oracle.
Indices are compared exactly, features within the 1e-3 scaled bound of
tests/test_gpu_mlp_v1.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import pointnet_numpy as P
from oracle import pointset as O
from oracle import v1_numpy as V

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def close(got, want, rtol=1e-3):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    s = np.sqrt(np.mean(want ** 2)) + 1e-30
    err = np.abs(got - want) / np.maximum(np.abs(want), s)
    assert err.max() <= rtol, "max scaled error %.3e" % err.max()


def fresh_store(device, seed=1):
    from fpn import variables
    return variables.VariableStore(device, seed=seed)


def cloud(b, n, c, seed):
    rng = np.random.RandomState(seed)
    xyz = rng.random_sample((b, n, 3)).astype(np.float32)
    pts = rng.standard_normal((b, n, c)).astype(np.float32) if c else None
    return xyz, pts


def test_group_concat_orders(device):
    from fpn.pointnet_util import group_concat
    xyz, pts = cloud(2, 300, 5, 0)
    new_xyz = xyz[:, :17].copy()
    idx, _ = O.query_ball_point(0.3, 12, xyz, new_xyz)
    gx = O.group_point(xyz, idx) - new_xyz[:, :, None, :]
    gp = O.group_point(pts, idx)
    got = group_concat(dev(xyz, device), dev(new_xyz, device), dev(pts, device), dev(idx, device), True)
    np.testing.assert_array_equal(got.cpu().numpy(), np.concatenate([gx, gp], -1))
    got = group_concat(dev(xyz, device), dev(new_xyz, device), dev(pts, device), dev(idx, device), False)
    np.testing.assert_array_equal(got.cpu().numpy(), np.concatenate([gp, gx], -1))
    got = group_concat(dev(xyz, device), dev(new_xyz, device), None, dev(idx, device))
    np.testing.assert_array_equal(got.cpu().numpy(), gx)


def test_sample_and_group(device):
    from fpn import pointnet_util as U
    xyz, pts = cloud(2, 512, 6, 1)
    new_xyz, new_points, idx, grouped_xyz = U.sample_and_group(64, 0.2, 16, dev(xyz, device), dev(pts, device))
    wx, wp, wi, wg = P.sample_and_group(64, 0.2, 16, xyz, pts)
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), wx)
    np.testing.assert_array_equal(new_points.cpu().numpy(), wp)
    np.testing.assert_array_equal(grouped_xyz.cpu().numpy(), wg)
    a, b, c, d = U.sample_and_group_all(dev(xyz, device), dev(pts, device))
    assert a.shape == (2, 1, 3) and (a == 0).all() and b.shape == (2, 1, 512, 9) and c.shape == (2, 1, 512)
    np.testing.assert_array_equal(b[:, 0, :, :3].cpu().numpy(), xyz)
    # knn grouping goes through select_top_k
    _, kp, kidx, _ = U.sample_and_group(8, None, 4, dev(xyz, device), dev(pts, device), knn=True)
    assert kp.shape == (2, 8, 4, 9) and kidx.shape == (2, 8, 4)


@pytest.mark.parametrize("pooling", ["max", "avg", "max_and_avg", "weighted_avg"])
def test_sa_module(device, pooling):
    from fpn import pointnet_util as U
    from fpn import variables
    xyz, pts = cloud(2, 512, 6, 2)
    st = fresh_store(device)
    with variables.use_store(st):
        new_xyz, feat, idx = U.pointnet_sa_module(dev(xyz, device), dev(pts, device), 64, 0.25, 32, [32, 48], None,
                                                  False, False, None, scope='sa', pooling=pooling)
    st.randomize_batch_norm(3)
    with variables.use_store(st):
        new_xyz, feat, idx = U.pointnet_sa_module(dev(xyz, device), dev(pts, device), 64, 0.25, 32, [32, 48], None,
                                                  False, False, None, scope='sa', pooling=pooling)
    assert feat.shape == (2, 64, 96 if pooling == "max_and_avg" else 48)
    if pooling == "max":
        wx, wf = P.sa_module(xyz, pts, 64, 0.25, 32, [32, 48], False, st.numpy(), "sa")
        np.testing.assert_array_equal(new_xyz.cpu().numpy(), wx)
        close(feat.cpu().numpy(), wf)


def test_sa_msg_and_fp_modules(device):
    from fpn import pointnet_util as U
    from fpn import variables
    xyz, pts = cloud(2, 1024, 1, 4)
    st = fresh_store(device)
    args = (64, [0.1, 0.2, 0.4], [32, 64, 128], [[16, 32], [32, 32], [32, 64]])
    for rnd in (0, 1):
        with variables.use_store(st):
            new_xyz, feat = U.pointnet_sa_module_msg(dev(xyz, device), dev(pts, device), *args, False, None, scope='m')
            up = U.pointnet_fp_module(dev(xyz, device), new_xyz, dev(pts, device), feat, [64, 32], False, None,
                                      scope='fp')
        if rnd == 0:
            st.randomize_batch_norm(5)
    wx, wf = P.sa_module_msg(xyz, pts, *args, st.numpy(), "m")
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), wx)
    close(feat.cpu().numpy(), wf)
    wu = P.fp_module(xyz, wx, pts, wf, [64, 32], st.numpy(), "fp")
    assert up.shape == (2, 1024, 32)
    close(up.cpu().numpy(), wu)


@pytest.mark.parametrize("b,n", [(3, 2048), (1, 2048), (1, 512)])
def test_v2_get_model_config2(device, b, n):
    """BASELINE config 2: v2 topology on 2048-point frustums; a single frustum; fewer points."""
    from fpn import synthetic, variables
    from fpn.models import frustum_pointnets_v2 as M2
    pc, one_hot = synthetic.make_frustums(b, n, 4, seed=2)
    pc[..., :3] *= 0.05      # bring the metric frustum into the radius range of the SA modules
    st = fresh_store(device, seed=7)
    with variables.use_store(st):
        M2.get_model(dev(pc, device), dev(one_hot, device), False)      # creates the variables
    st.randomize_batch_norm(8)
    with variables.use_store(st):
        ep = M2.get_model(dev(pc, device), dev(one_hot, device), False)
    params = st.numpy()
    logits = P.v2_instance_seg(pc, one_hot, params, M2.CONFIG)
    close(ep["mask_logits"].cpu().numpy(), logits)
    mask = ep["mask"].cpu().numpy()
    obj, mean, _, idx = V.point_cloud_masking(pc, logits, mask_override=mask)
    np.testing.assert_array_equal(ep["object_point_indices"].cpu().numpy(), idx)
    delta = V.center_regression_net(obj, one_hot, params)
    close(ep["center_delta"].cpu().numpy(), delta)
    # the box net's own FPS runs on (obj - delta); feed the oracle the GPU's delta so that
    # rounding differences in delta cannot change the sampled set
    gdelta = ep["center_delta"].cpu().numpy()
    out = P.v2_box_estimation(obj - gdelta[:, None, :], one_hot, params, M2.CONFIG)
    close(ep["box_output"].cpu().numpy(), out)
    assert ep["center"].shape == (b, 3) and ep["size_residuals"].shape == (b, 8, 3)


def test_sa_chain_kernels(device):
    """Every shape the fused set-abstraction kernel is built for (grouping + 3-layer MLP +
    max over the samples in one launch) against the numpy oracle and against the
    per-layer path; includes widths that are padded inside (32, 96, 64-wide outputs),
    both concatenation orders and feature widths 0, 1, 128, 320."""
    from fpn import pointnet_util as U
    from fpn import variables
    rng = np.random.RandomState(0)
    b = 2
    xyz_big = rng.uniform(0, 1, (b, 1024, 3)).astype(np.float32)
    xyz_small = rng.uniform(0, 1, (b, 128, 3)).astype(np.float32)
    cases = [
        ("msg", xyz_big, rng.standard_normal((b, 1024, 1)).astype(np.float32),
         (64, [0.1, 0.2, 0.4], [32, 64, 128], [[32, 32, 64], [64, 64, 128], [64, 96, 128]])),
        ("msg", xyz_small, rng.standard_normal((b, 128, 320)).astype(np.float32),
         (32, [0.4, 0.8, 1.6], [64, 64, 128], [[64, 64, 128], [128, 128, 256], [128, 128, 256]])),
        ("ssg", xyz_big, None, (64, 0.2, 64, [64, 64, 128])),
        ("ssg", xyz_small, rng.standard_normal((b, 128, 128)).astype(np.float32), (32, 0.4, 64, [128, 128, 256])),
    ]
    for kind, xyz, pts, args in cases:
        st = fresh_store(device, seed=len(args[-1]))
        outs = {}
        for fuse in (True, True, False):            # first pass creates the variables
            U.FUSE_SA = fuse
            try:
                with variables.use_store(st):
                    p = None if pts is None else dev(pts, device)
                    if kind == "msg":
                        new_xyz, feat = U.pointnet_sa_module_msg(dev(xyz, device), p, *args, False, None, scope='s')
                    else:
                        new_xyz, feat, _ = U.pointnet_sa_module(dev(xyz, device), p, *args, None, False, False, None,
                                                                scope='s')
            finally:
                U.FUSE_SA = True
            if "cache" not in outs:
                assert "_sa_chain_cache" in st.__dict__ and len(st.__dict__["_sa_chain_cache"]) >= 1, "fused path not taken"
                outs["cache"] = True
                st.randomize_batch_norm(7)
                continue
            outs[fuse] = feat.cpu().numpy()
        if kind == "msg":
            wx, wf = P.sa_module_msg(xyz, pts, *args, st.numpy(), "s")
        else:
            wx, wf = P.sa_module(xyz, pts, *args, False, st.numpy(), "s")
        np.testing.assert_array_equal(new_xyz.cpu().numpy(), wx)
        assert outs[True].shape == wf.shape
        close(outs[True], wf)
        close(outs[True], outs[False], 1e-4)
        if pts is not None and pts.shape[2] >= 32:
            # wide features: multiplied once per point and gathered as products (the default), or gathered and
            # multiplied per row inside the kernel - the same layer up to fp32 rounding
            assert any(k[-1] is True for k in st.__dict__["_sa_chain_cache"]), "hoisted form not taken"
            U.HOIST_FEATURES = False
            try:
                with variables.use_store(st):
                    if kind == "msg":
                        _, feat = U.pointnet_sa_module_msg(dev(xyz, device), dev(pts, device), *args, False, None, scope='s')
                    else:
                        _, feat, _ = U.pointnet_sa_module(dev(xyz, device), dev(pts, device), *args, None, False, False, None,
                                                          scope='s')
            finally:
                U.HOIST_FEATURES = True
            assert any(k[-1] is False and k[1] >= 32 for k in st.__dict__["_sa_chain_cache"]), "in-kernel form not taken"
            close(outs[True], feat.cpu().numpy(), 1e-5)


@pytest.mark.parametrize("cb", [4, 0, 32])
def test_fp_head_chain(device, cb):
    """Feature propagation + per-point head as one kernel (fp32 and bf16) against the same
    layers run one call each; cb own channels per point (0: no skip features)."""
    from fpn import pointnet_util as U
    from fpn import variables
    rng = np.random.RandomState(cb)
    b, n, m = 3, 224, 48          # 672 rows: a last workgroup with one idle wave
    xyz1 = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
    xyz2 = np.ascontiguousarray(xyz1[:, :m])
    p1 = rng.standard_normal((b, n, cb)).astype(np.float32) if cb else None
    p2 = rng.standard_normal((b, m, 128)).astype(np.float32)
    st = fresh_store(device, seed=4)
    outs = {}
    for mode in ("create", "fused", "layers", "bf16"):
        U.FUSE_SA = mode != "layers"
        U.SA_PRECISION = "bf16" if mode == "bf16" else "fp32"
        try:
            with variables.use_store(st):
                y = U.pointnet_fp_module_with_head(dev(xyz1, device), dev(xyz2, device),
                                                   None if p1 is None else dev(p1, device), dev(p2, device),
                                                   [128, 128], 128, 2, False, None, scope='fa', head_scope='h1',
                                                   out_scope='h2')
        finally:
            U.FUSE_SA = True
            U.SA_PRECISION = "fp32"
        if mode == "create":
            st.randomize_batch_norm(6)
            continue
        assert y.shape == (b, n, 2)
        outs[mode] = y.cpu().numpy()
    assert len(st.__dict__.get("_fp_head_cache", {})) == 2, "fused kernels not taken"
    close(outs["fused"], outs["layers"], 1e-4)
    if cb <= 8:
        # the default took the form that interpolates first-layer products; the form that interpolates the features:
        assert all("pre" in v for v in st.__dict__["_fp_head_cache"].values())
        for precision in ("fp32", "bf16"):
            U.HOIST_FEATURES, U.SA_PRECISION = False, precision
            try:
                with variables.use_store(st):
                    y = U.pointnet_fp_module_with_head(dev(xyz1, device), dev(xyz2, device),
                                                       None if p1 is None else dev(p1, device), dev(p2, device),
                                                       [128, 128], 128, 2, False, None, scope='fa', head_scope='h1',
                                                       out_scope='h2')
            finally:
                U.HOIST_FEATURES, U.SA_PRECISION = True, "fp32"
            if precision == "fp32":
                close(outs["fused"], y.cpu().numpy(), 1e-5)
            else:
                close(y.cpu().numpy(), outs["layers"], 4e-2)
    got, want = outs["bf16"].astype(np.float64), outs["layers"].astype(np.float64)
    s = np.sqrt(np.mean(want ** 2))
    err = (np.abs(got - want) / np.maximum(np.abs(want), s)).max()
    assert 1e-6 < err <= 4e-2, err


def test_sa_and_fp_modules_random_configurations(device):
    """Twelve seeded random module configurations - sample counts that are no multiple of
    32, one- to three-layer MLPs of odd widths, with and without features, group_all -
    so that both the fused kernels and the per-layer fall-backs (and their un-fused
    pooling) run; each against the numpy restatement."""
    from fpn import pointnet_util as U
    from fpn import variables
    rng = np.random.RandomState(77)
    for trial in range(12):
        b = int(rng.choice([1, 2, 3]))
        n = int(rng.choice([64, 200, 512]))
        cf = int(rng.choice([0, 1, 5, 32]))
        npoint = int(rng.choice([8, 16, 32]))
        nsample = int(rng.choice([8, 20, 32, 64]))
        radius = float(rng.choice([0.15, 0.3, 0.6]))
        mlp = [int(w) for w in rng.choice([16, 24, 32, 64, 96, 128], size=int(rng.choice([1, 2, 3])))]
        group_all = bool(trial % 4 == 3)
        xyz, pts = cloud(b, n, max(cf, 1), 100 + trial)
        pts = pts if cf else None
        st = fresh_store(device, seed=trial)
        scope = "sa%d" % trial
        args = (None, None, None) if group_all else (npoint, radius, nsample)
        for rnd in (0, 1):
            with variables.use_store(st):
                new_xyz, feat, idx = U.pointnet_sa_module(dev(xyz, device), None if pts is None else dev(pts, device),
                                                          *args, mlp, None, group_all, False, None, scope=scope)
            if rnd == 0:
                st.randomize_batch_norm(trial)
        wx, wf = P.sa_module(xyz, pts, args[0], args[1], args[2], mlp, group_all, st.numpy(), scope)
        tag = (trial, b, n, cf, npoint, nsample, radius, mlp, group_all)
        np.testing.assert_array_equal(new_xyz.cpu().numpy(), wx, err_msg=str(tag))
        close(feat.cpu().numpy(), wf)
        # propagate back to the dense points through an FP module of random widths
        fp_mlp = [int(w) for w in rng.choice([16, 40, 128], size=int(rng.choice([1, 2])))]
        with variables.use_store(st):
            up = U.pointnet_fp_module(dev(xyz, device), new_xyz, None if pts is None else dev(pts, device), feat,
                                      fp_mlp, False, None, scope=scope + "fp")
        wu = P.fp_module(xyz, wx, pts, wf, fp_mlp, st.numpy(), scope + "fp")
        assert up.shape == (b, n, fp_mlp[-1]), tag
        close(up.cpu().numpy(), wu)


def test_sa_msg_random_configurations(device):
    """Multi-scale grouping with random branches: fused and per-layer branches side by side
    in one output buffer (column slices of different widths)."""
    from fpn import pointnet_util as U
    from fpn import variables
    rng = np.random.RandomState(5)
    for trial in range(8):
        b, n = int(rng.choice([1, 2])), int(rng.choice([128, 300, 1024]))
        cf = int(rng.choice([0, 1, 3, 64]))
        npoint = int(rng.choice([16, 32]))
        nb = int(rng.choice([1, 2, 3]))
        radii = [float(r) for r in rng.choice([0.1, 0.2, 0.4, 0.8], size=nb)]
        nsamples = [int(s) for s in rng.choice([16, 32, 40, 64, 128], size=nb)]
        mlps = [[int(w) for w in rng.choice([32, 48, 64, 96, 128], size=3)] for _ in range(nb)]
        xyz, pts = cloud(b, n, max(cf, 1), 200 + trial)
        pts = pts if cf else None
        st = fresh_store(device, seed=10 + trial)
        scope = "msg%d" % trial
        for rnd in (0, 1):
            with variables.use_store(st):
                new_xyz, feat = U.pointnet_sa_module_msg(dev(xyz, device), None if pts is None else dev(pts, device),
                                                         npoint, radii, nsamples, mlps, False, None, scope=scope)
            if rnd == 0:
                st.randomize_batch_norm(trial)
        wx, wf = P.sa_module_msg(xyz, pts, npoint, radii, nsamples, mlps, st.numpy(), scope)
        np.testing.assert_array_equal(new_xyz.cpu().numpy(), wx, err_msg=str((trial, radii, nsamples, mlps)))
        assert feat.shape == (b, npoint, sum(m[-1] for m in mlps))
        close(feat.cpu().numpy(), wf)


@pytest.mark.parametrize("c1,c2,mlp", [(40, 67, [64, 32]), (640, 1027, [128, 128]), (5, 3, [16])])
def test_fp_module_single_known_point(device, c1, c2, mlp):
    """Feature propagation from ONE known point (the level above was a group-all set
    abstraction; models/pointnet_util.py:198-228 with m = 1): the interpolation is an exact
    broadcast, run here as a per-frustum bias of the first layer.  Against the numpy oracle
    (which interpolates and concatenates) and against this build's own interpolate-and-
    concatenate path."""
    from fpn import pointnet_util as U
    from fpn import variables
    b, n = 3, 96
    rng = np.random.RandomState(c1)
    xyz1 = rng.random_sample((b, n, 3)).astype(np.float32)
    xyz2 = rng.random_sample((b, 1, 3)).astype(np.float32)
    p1 = rng.standard_normal((b, n, c1)).astype(np.float32)
    p2 = rng.standard_normal((b, 1, c2)).astype(np.float32)
    st = fresh_store(device, seed=9)
    got = {}
    for fused in (True, False):
        old = U.FUSE_SA
        U.FUSE_SA = fused
        try:
            for rnd in (0, 1):
                with variables.use_store(st):
                    got[fused] = U.pointnet_fp_module(dev(xyz1, device), dev(xyz2, device), dev(p1, device), dev(p2, device),
                                                      mlp, False, None, scope='fp1')
                if fused and rnd == 0:
                    st.randomize_batch_norm(3)
        finally:
            U.FUSE_SA = old
    want = P.fp_module(xyz1, xyz2, p1, p2, mlp, st.numpy(), "fp1")
    assert got[True].shape == (b, n, mlp[-1])
    close(got[True].cpu().numpy(), want)
    close(got[True].cpu().numpy(), got[False].cpu().numpy(), rtol=1e-5)


def test_persistent_set_abstraction_bit_for_bit(device, monkeypatch):
    """fp32 set abstraction over coordinates (64-64-128, 64-96-128 with both layers in LDS,
    128-128-256 with the second): the
    persistent kernel and the kernel launched per 32 samples (FPN_F32_SA=launch) give the
    same bits - one to eight rounds per ball, 0 / 1 / 4 direct feature channels, ball counts
    that leave waves and workgroups without work, and wide features entering as first-layer
    products."""
    from fpn import pointnet_util as U
    from fpn import variables
    monkeypatch.setattr(U, "POISON_UNCLEARED", True)      # outputs the persistent kernels are trusted to fill start as NaN
    rng = np.random.RandomState(11)
    cases = [("msg", 3, 1024, 1, (64, [0.1, 0.2, 0.4], [32, 64, 128], [[32, 32, 64], [64, 64, 128], [64, 96, 128]])),
             ("ssg", 2, 512, 0, (5, 0.3, 32, [64, 96, 128])),
             ("ssg", 1, 512, 4, (3, 0.3, 128, [64, 64, 128])),
             ("ssg", 5, 700, 1, (129, 0.25, 256, [64, 64, 128])),
             ("ssg", 40, 256, 0, (128, 0.3, 64, [64, 96, 128])),
             ("msg", 3, 128, 320, (32, [0.4, 0.8], [64, 32], [[64, 64, 128], [64, 96, 128]])),
             # 128-128-256: the second layer from LDS, the third from memory
             ("msg", 3, 128, 320, (32, [0.4, 0.8, 1.6], [64, 64, 128], [[64, 64, 128], [128, 128, 256], [128, 128, 256]])),
             ("ssg", 2, 128, 128, (7, 0.5, 32, [128, 128, 256]))]
    for kind, b, n, cf, args in cases:
        xyz = dev(rng.uniform(0, 1, (b, n, 3)).astype(np.float32), device)
        pts = dev(rng.standard_normal((b, n, cf)).astype(np.float32), device) if cf else None
        st = fresh_store(device, seed=6)
        got = {}
        for mode in ("persistent", "launch", "persistent"):
            if mode == "launch":
                monkeypatch.setenv("FPN_F32_SA", "launch")
            else:
                monkeypatch.delenv("FPN_F32_SA", raising=False)
            with variables.use_store(st):
                if kind == "msg":
                    _, feat = U.pointnet_sa_module_msg(xyz, pts, *args, False, None, scope='s')
                else:
                    _, feat, _ = U.pointnet_sa_module(xyz, pts, *args, None, False, False, None, scope='s')
            got[mode] = feat.clone()
        assert len(st.__dict__.get("_sa_chain_cache", {})) >= 1, "fused kernel not taken"
        assert torch.equal(got["persistent"], got["launch"]), (kind, b, args[0])
        assert got["persistent"].abs().max() > 0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_v2_outputs_do_not_depend_on_uncleared_buffers(device, monkeypatch, precision):
    """Set-abstraction outputs are cleared only for the kernels that merge maxima into them
    (fpn_sa_chain_overwrites / fpn_sa_chain_bf16_overwrites say which): with every uncleared
    output filled with NaN first, the v2 net gives the same bits."""
    from fpn import pointnet_util as U
    from fpn import synthetic, variables
    from fpn.models import frustum_pointnets_v2 as M2
    monkeypatch.setattr(U, "SA_PRECISION", precision)
    pc, one_hot = synthetic.make_frustums(3, 2048, 4, seed=4)
    pc[..., :3] *= 0.05
    st = fresh_store(device, seed=9)
    with variables.use_store(st):
        M2.get_model(dev(pc, device), dev(one_hot, device), False)
    st.randomize_batch_norm(3)
    got = {}
    for poison in (False, True):
        monkeypatch.setattr(U, "POISON_UNCLEARED", poison)
        with variables.use_store(st):
            ep = M2.get_model(dev(pc, device), dev(one_hot, device), False)
        got[poison] = {k: ep[k].clone() for k in ("mask_logits", "center_delta", "box_output")}
    for k, v in got[True].items():
        assert torch.isfinite(v).all(), k
        assert torch.equal(v, got[False][k]), k


@pytest.mark.parametrize("cf,ns,m", [(0, 64, 33), (4, 96, 7), (1, 32, 130)])
def test_narrow_set_abstraction_kernel(device, cf, ns, m):
    """The 32-32-64 stack at its own widths (persistent kernel, weights in registers) against the
    per-layer path and the float64 restatement: several groups of 32 samples per ball (maxima
    merged across rounds and waves), 0 / 1 / 4 direct feature channels, ball counts that leave
    waves without work."""
    from fpn import pointnet_util as U
    from fpn import variables
    rng = np.random.RandomState(ns + cf)
    b, n = 3, 600
    xyz = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
    pts = rng.standard_normal((b, n, cf)).astype(np.float32) if cf else None
    st = fresh_store(device, seed=3)
    outs = {}
    for fuse in (True, True, False):
        U.FUSE_SA = fuse
        try:
            with variables.use_store(st):
                new_xyz, feat, _ = U.pointnet_sa_module(dev(xyz, device), None if pts is None else dev(pts, device), m, 0.3, ns,
                                                        [32, 32, 64], None, False, False, None, scope='s')
        finally:
            U.FUSE_SA = True
        if "init" not in outs:
            outs["init"] = True
            st.randomize_batch_norm(5)
            continue
        outs[fuse] = feat.cpu().numpy()
    assert len(st.__dict__.get("_sa_chain_cache", {})) == 1 and list(st.__dict__["_sa_chain_cache"].values())[0]["pad"] == (32, 32, 64)
    wx, wf = P.sa_module(xyz, pts, m, 0.3, ns, [32, 32, 64], False, st.numpy(), "s")
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), wx)
    close(outs[True], wf)
    close(outs[True], outs[False], 1e-4)
