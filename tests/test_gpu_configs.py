"""The two workload configurations no other GPU test runs at their stated sizes:

  * the Bayesian detector (models/FPointNet_Bayes.py:54-68,92,140-223) with THIRTY-TWO
    Monte-Carlo samples per frustum, Flipout noise and Monte-Carlo dropout, 1024-point
    frustums, fp32 and bf16: samples spread over the passes against the float64 textbook
    Flipout oracle, then the ensemble statistics of that real output;
  * the v2 topology (models/pointnet_util.py:86-228 composed as fpn/models/v2_config.json
    says) in bf16 on 2048-point frustums against the float64 numpy oracle under the
    stated bf16 bound, and a chunk of 1024 such frustums through the independence
    property (a frustum's results do not depend on the rest of the batch).

Flipout parity is unpinned by the reference (the arithmetic lives in
tensorflow_probability, absent); the oracle is the published estimator with the same
noise."""
import numpy as np
import pytest
import torch

import bf16_bounds as BB
from oracle import bayes_numpy as BN
from oracle import pointnet_numpy as P
from oracle import v1_numpy as V

pytestmark = pytest.mark.gpu

# bf16: the stated bounds of tests/bf16_bounds.py (MODEL_TOL against the rounding model,
# EXACT_TOL against the unrounded network, both over the summed magnitude of the last layer)
FP32_RTOL = 1e-3


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def scaled_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    s = np.sqrt(np.mean(want ** 2)) + 1e-30
    return float((np.abs(got - want) / np.maximum(np.abs(want), s)).max())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_bayes_thirty_two_samples(device, precision):
    from fpn import synthetic, uncertainty
    B, N, S = 6, 1024, 32
    st, bp = BB.bayes_setup(device, 8, True, precision)
    pc, oh = synthetic.make_frustums(B, N, 3, seed=21)
    g = torch.Generator(device=device)
    g.manual_seed(17)
    noise = bp.sample_noise(S, B, g, N)
    ep = bp.forward(dev(pc, device), dev(oh, device), S, noise=noise)
    assert ep["mask_logits"].shape == (S, B, N, 2) and ep["box_output"].shape == (S, B, 59)
    for k in ("mask_logits", "box_output", "stage1_center"):
        assert torch.isfinite(ep[k]).all(), k
    picked = (0, 5, 13, 22, 31)       # first, last and three in between
    if precision == "bf16":
        model, exact = {}, {}
        BB.bayes_compare(st, ep, noise, pc, oh, picked, model, exact)
        print("Bayes S=32 bf16, errors in u over the summed magnitude: vs rounding model %s; vs unrounded %s" % (
            {k: "%.2f" % (v / BB.U) for k, v in model.items()}, {k: "%.2f" % (v / BB.U) for k, v in exact.items()}))
        assert max(model.values()) <= BB.MODEL_TOL, model
        assert max(exact.values()) <= BB.EXACT_TOL, exact
    else:
        params = {k: v.astype(np.float64) for k, v in st.numpy().items()}
        pc64, oh64 = pc.astype(np.float64), oh.astype(np.float64)
        worst = {}
        for s in picked:
            keep = BB.keep_mask(noise, s)
            nz = BB.flipout_noise_of_sample(st, noise, s)
            want = BN.forward_sample(pc64, oh64, params, nz, dropout_keep=keep)
            worst["mask_logits"] = max(worst.get("mask_logits", 0),
                                       scaled_err(ep["mask_logits"][s].cpu().numpy(), want["mask_logits"]))
            mask = ep["mask"][s].cpu().numpy()
            # a point may flip only where the two logits agree within the bound
            flips = mask != want["mask"]
            if flips.any():
                gap = np.abs(want["mask_logits"][..., 0] - want["mask_logits"][..., 1])[flips]
                assert gap.max() <= 2 * FP32_RTOL * np.sqrt(np.mean(want["mask_logits"] ** 2)), (s, gap.max())
            want = BN.forward_sample(pc64, oh64, params, nz, mask_override=mask, dropout_keep=keep)
            np.testing.assert_array_equal(ep["object_point_indices"][s].cpu().numpy(), want["object_point_indices"])
            for k in ("center_delta", "stage1_center", "box_output", "center"):
                worst[k] = max(worst.get(k, 0), scaled_err(ep[k][s].cpu().numpy(), want[k]))
        print("Bayes S=32 fp32 scaled errors:", {k: "%.2e" % v for k, v in worst.items()})
        assert max(worst.values()) <= FP32_RTOL, worst
    # all 32 samples differ (fresh weights and dropout bits per sample)
    flat = ep["box_output"].reshape(S, -1)
    assert len({tuple(r.tolist()) for r in flat.cpu()}) == S
    # ensemble statistics of the network's own 32 samples
    stats = uncertainty.mc_statistics(ep["box_output"].contiguous(), ep["stage1_center"].contiguous())
    got = {k: v.cpu().numpy() for k, v in uncertainty.unpack(stats).items()}
    want = BN.mc_statistics(ep["box_output"].cpu().numpy(), ep["stage1_center"].cpu().numpy())
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=2e-5, atol=2e-5, err_msg=k)
    assert (got["total_variance"] > 0).all()


def _v2_store(device, pc, one_hot, seed=7):
    from fpn import variables
    from fpn.models import frustum_pointnets_v2 as M2
    st = variables.VariableStore(device, seed=seed)
    with variables.use_store(st):
        M2.get_model(dev(pc[:1], device), dev(one_hot[:1], device), False)      # creates the variables
    st.randomize_batch_norm(seed + 1)
    return st


def _v2_forward(device, st, pc, one_hot, precision):
    from fpn import pointnet_util, variables
    from fpn.models import frustum_pointnets_v2 as M2
    old = pointnet_util.SA_PRECISION
    pointnet_util.SA_PRECISION = precision
    try:
        with variables.use_store(st):
            return M2.get_model(dev(pc, device), dev(one_hot, device), False)
    finally:
        pointnet_util.SA_PRECISION = old


def test_v2_bf16_2048_points_against_float64_oracle(device):
    from fpn import synthetic
    b, n = 4, 2048
    pc, one_hot = synthetic.make_frustums(b, n, 4, seed=12)
    pc[..., :3] *= 0.05      # bring the metric frustum into the radius range of the SA modules
    st = _v2_store(device, pc, one_hot)
    ep = _v2_forward(device, st, pc, one_hot, "bf16")
    assert any(len(k) >= 4 and k[3] == "bf16" for k in st.__dict__["_sa_chain_cache"]), "bf16 kernels not taken"
    exact = BB.v2_exact_errors(ep, st, pc, one_hot)
    print("v2 bf16 4 x 2048, errors in u over the summed magnitude:", {k: "%.2f" % (v / BB.U) for k, v in exact.items()})
    assert max(exact.values()) <= BB.EXACT_TOL, exact
    assert exact["mask_logits"] > 1e-5, "suspiciously exact: is the bf16 path running?"
    assert ep["center"].shape == (b, 3) and ep["size_residuals"].shape == (b, 8, 3)


def test_v2_bf16_chunk_of_1024_frustums_properties(device):
    """The unit of work of the streamed 200k-frustum scene set: one chunk of 1024 frustums x
    2048 points in bf16.  Frustums are independent: a frustum's results inside the chunk
    equal its results in a batch of four - sampling and grouping indices bit for bit, features
    up to the summation order of the layers whose kernel is chosen by the batch's row count
    (few rows: K split over a workgroup's waves), which the bf16 roundings after them can
    turn into single bf16 steps; everything is finite; and those four match the float64
    oracle under the bf16 bound."""
    from fpn import synthetic
    from fpn.models import frustum_pointnets_v2 as M2
    b, n = 1024, 2048
    pc, one_hot = synthetic.make_frustums(b, n, 4, seed=3)
    pc[..., :3] *= 0.05
    st = _v2_store(device, pc, one_hot, seed=11)
    full = _v2_forward(device, st, pc, one_hot, "bf16")
    sub = [0, 333, 777, 1023]
    alone = _v2_forward(device, st, pc[sub], one_hot[sub], "bf16")
    params = {k: v.astype(np.float64) for k, v in st.numpy().items()}
    mag = {}
    logits = P.v2_instance_seg(pc[sub], one_hot[sub], params, M2.CONFIG, dtype=np.float64, magnitude=mag)
    err = BB.magnitude_error(full["mask_logits"][sub].cpu().numpy(), alone["mask_logits"].cpu().numpy(), mag["mask_logits"])
    assert err <= BB.MODEL_TOL, err
    same_mask = (full["mask"][sub] == alone["mask"]).all(dim=1).cpu().numpy()
    assert same_mask.any()
    for k in ("object_point_indices",):
        assert torch.equal(full[k][sub][same_mask], alone[k][same_mask]), k
    for k in ("center_delta", "box_output", "center"):
        err = scaled_err(full[k][sub][same_mask].cpu().numpy(), alone[k][same_mask].cpu().numpy())
        assert err <= 1e-2, (k, err)
    assert torch.isfinite(full["mask_logits"]).all() and torch.isfinite(full["box_output"]).all()
    assert full["mask_logits"].shape == (b, n, 2) and full["box_output"].shape == (b, 59)
    err = BB.magnitude_error(alone["mask_logits"].cpu().numpy(), logits, mag["mask_logits"])
    assert 1e-5 < err <= BB.EXACT_TOL, err
