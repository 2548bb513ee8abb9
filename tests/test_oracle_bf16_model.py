"""CPU checks of the float64 rounding model of the bf16 path (oracle/bf16_numpy.py) - the
checker the GPU tests hold the bf16 kernels to - so that the checker itself is pinned:
its rounding is bfloat16's round-to-nearest-even, its split operands carry 16 bits, it is
the unrounded network wherever nothing needs rounding, and its distance to the unrounded
network stays within the stated bound (tests/bf16_bounds.py) with no GPU involved."""
import numpy as np
import torch

import bf16_bounds as BB
from oracle import bf16_numpy as Q
from oracle import v1_numpy as V


def test_rounding_is_bfloat16_round_to_nearest_even():
    rng = np.random.RandomState(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-30, 30, 200000))).astype(np.float32)
    # exact ties: 1 + 2^-8 (-> 1, even), 1 + 3 * 2^-8 (-> 1 + 2^-6 ... the even neighbour), signed zero, a subnormal
    x[:6] = [1.00390625, 1.01171875, 0.0, -0.0, 1e-40, -3.0e38]
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float64).numpy()
    got = Q.rb(x)
    assert np.array_equal(got, want)
    assert got[0] == 1.0 and got[1] == 1.015625
    hi, lo = Q.split(x[6:])
    assert np.array_equal(hi, Q.rb(x[6:]))
    err = np.abs(hi + lo - x[6:].astype(np.float64))
    assert (err <= 2.0 ** -16 * np.abs(x[6:])).all()


def _store(seed, bn=True, channels=4):
    from fpn import plan, variables
    st = variables.VariableStore(torch.device("cpu"), seed=seed)
    plan.init_variables(st, channels, bn)
    if bn:
        st.randomize_batch_norm(seed + 1)
    return st


def test_model_equals_unrounded_network_on_bfloat16_exact_data():
    """Weights that select and scale by small integers, integer inputs: every activation is
    an integer below 256, nothing is changed by rounding, so the rounding model must return
    the unrounded network bit for bit."""
    from fpn import plan
    st = _store(3, bn=False)
    rng = np.random.RandomState(3)
    for scope, cin, cout, _, _ in plan.layer_table(4):
        w = np.zeros((cin, cout), np.float32)
        cols = np.arange(cout)
        if scope == "conv1":
            w[cols % 4, cols] = cols // 4 + 1
        elif scope in ("conv-reg1-stage1", "conv-reg1"):
            w[cols % 3, cols] = 1 + (cols // 3) % 4
        else:
            w[rng.randint(0, cin, cout), cols] = 1
        st[scope + "/weights"] = torch.from_numpy(w)
    pc = rng.randint(0, 8, (3, 64, 4)).astype(np.float32)
    oh = np.eye(3, dtype=np.float32)[rng.randint(0, 3, 3)]
    params = st.numpy()
    want = V.get_model(pc, oh, params, bn=False, dtype=np.float64)
    got = Q.forward(pc, oh, Q.Weights(params, False))
    assert np.abs(want["mask_logits"]).max() > 0
    # (the stacks after the mask read coordinates minus the masked mean - not integers)
    np.testing.assert_array_equal(got["mask_logits"], want["mask_logits"])
    np.testing.assert_array_equal(got["object_point_indices"], want["object_point_indices"])


def test_rounding_model_within_the_stated_distance_of_the_unrounded_network():
    """The bound EXACT_TOL is a statement about the roundings, not about the kernels: the
    model alone must keep it (eight draws here; the GPU tests sweep 32 through the kernels)."""
    from fpn import synthetic
    worst = {}
    for seed in range(8):
        st = _store(100 + seed)
        params = st.numpy()
        pc, oh = synthetic.make_frustums(3, 128, 4, seed=200 + seed)
        model = Q.forward(pc, oh, Q.Weights(params, True))
        exact = V.get_model(pc, oh, params, dtype=np.float64, mask_override=model["mask"])
        for k, m in model["summed_magnitude"].items():
            e = BB.magnitude_error(model[k], exact[k], m)
            worst[k] = max(worst.get(k, 0.0), e)
            # the output's own size can be far below what it sums: that is why the bound is not stated on it
            assert (np.abs(exact[k]) <= m * (1 + 1e-9) + 1e-12).all()
    assert 0.1 * BB.U < max(worst.values()) <= BB.EXACT_TOL, worst


def test_flipout_weights_of_the_model_are_the_textbook_estimator():
    """With all roundings switched off by bfloat16-exact data the Flipout form of the model
    (one effective matrix per frustum) equals the two-product estimator of
    oracle/bayes_numpy.py."""
    from fpn import bayes, variables
    from oracle import bayes_numpy as BN
    st = variables.VariableStore(torch.device("cpu"), seed=5)
    bayes.init_variables(st, 3)
    rng = np.random.RandomState(5)
    params = st.numpy()
    B, N = 2, 32
    noise = {}
    from fpn import plan
    for scope, cin, cout, _, _ in plan.layer_table(3):
        if scope in BN.PLAIN:
            continue
        # weights and perturbations on a coarse binary grid, signs +-1: sums of a few such products stay exact
        params[scope + "/weights"] = (rng.randint(-2, 3, (cin, cout)) * (rng.rand(cin, cout) < 4.0 / cin)).astype(np.float32)
        noise[scope] = ((rng.randint(-1, 2, (cin, cout)) * (rng.rand(cin, cout) < 2.0 / cin)).astype(np.float64),
                        rng.choice([-1.0, 1.0], (B, cin)), rng.choice([-1.0, 1.0], (B, cout)))
    pc = rng.randint(0, 4, (B, N, 3)).astype(np.float32)
    oh = np.eye(3, dtype=np.float32)[[0, 2]]
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    want = BN.forward_sample(pc.astype(np.float64), oh.astype(np.float64), p64, noise)
    got = Q.forward(pc, oh, Q.FlipoutWeights(params, noise))
    # integer arithmetic while values stay below 256; allow the rare larger activation its bf16 step
    np.testing.assert_allclose(got["mask_logits"], want["mask_logits"], rtol=2.0 ** -6, atol=1e-9)


def _full_flipout_store(seed):
    """Variables of the network of models/frustum_pointnet.py (batch norm everywhere but the
    last layer, rho for every layer), batch-norm statistics random."""
    from fpn import bayes, variables
    st = variables.VariableStore(torch.device("cpu"), seed=seed)
    bayes.init_variables(st, 3, bn=True, flipout_outputs=True)
    st.randomize_batch_norm(seed + 1)
    return st


def _full_noise(rng, B, rho_scale, params):
    from fpn import plan
    from oracle import bayes_numpy as BN
    noise = {}
    for scope, cin, cout, _, _ in plan.layer_table(3):
        sigma = np.log1p(np.exp(params[scope + "/rho"].astype(np.float64)))
        noise[scope] = (rho_scale * sigma * rng.standard_normal((cin, cout)),
                        rng.choice([-1.0, 1.0], (B, cin)), rng.choice([-1.0, 1.0], (B, cout)))
    return noise


def test_batch_norm_placement_of_the_flipout_estimator_is_the_v1_networks():
    """With the perturbation at zero the Flipout + batch-norm estimator is a deterministic
    network: the one of oracle/v1_numpy.py (pinned on its own fixtures) with batch norm,
    except that it ALSO normalises conv10 and fc3-stage1 - give those two identity statistics
    and the two must agree to float64 rounding."""
    from oracle import bayes_numpy as BN
    st = _full_flipout_store(11)
    for scope in ("conv10", "fc3-stage1"):
        n = st[scope + "/biases"].shape[0]
        st[scope + "/bn/gamma"] = torch.ones(n)
        st[scope + "/bn/beta"] = torch.zeros(n)
        st[scope + "/bn/moving_mean"] = torch.zeros(n)
        st[scope + "/bn/moving_variance"] = torch.full((n,), 1.0) - V.BN_EPS
    params = {k: v.astype(np.float64) for k, v in st.numpy().items()}
    for scope in ("conv10", "fc3-stage1"):
        params[scope + "/bn/moving_variance"] = np.full_like(params[scope + "/bn/moving_variance"], 1.0 - V.BN_EPS)
    from fpn import synthetic
    pc, oh = synthetic.make_frustums(3, 64, 3, seed=12)
    rng = np.random.RandomState(13)
    noise = _full_noise(rng, 3, 0.0, params)
    got = BN.forward_sample(pc.astype(np.float64), oh.astype(np.float64), params, noise, bn=True)
    want = V.get_model(pc, oh, params, bn=True, dtype=np.float64)
    np.testing.assert_allclose(got["mask_logits"], want["mask_logits"], rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(got["object_point_indices"], want["object_point_indices"])
    np.testing.assert_allclose(got["center"], want["center"], rtol=1e-9, atol=1e-9)
    # and the statistics of those two layers are used when they are not the identity
    params["conv10/bn/gamma"] = params["conv10/bn/gamma"] * 2.0
    twice = BN.forward_sample(pc.astype(np.float64), oh.astype(np.float64), params, noise, bn=True)
    assert np.abs(twice["mask_logits"] - got["mask_logits"]).max() > 1e-3


def test_rounding_model_of_flipout_with_batch_norm_within_the_stated_distance():
    from fpn import synthetic
    from oracle import bayes_numpy as BN
    worst = {}
    for seed in range(3):
        st = _full_flipout_store(300 + seed)
        params = st.numpy()
        p64 = {k: v.astype(np.float64) for k, v in params.items()}
        pc, oh = synthetic.make_frustums(3, 128, 3, seed=400 + seed)
        noise = _full_noise(np.random.RandomState(500 + seed), 3, 1.0, p64)
        model = Q.forward(pc, oh, Q.FlipoutWeights(params, noise, bn=True))
        exact = BN.forward_sample(pc.astype(np.float64), oh.astype(np.float64), p64, noise, bn=True,
                                  mask_override=model["mask"])
        for k, m in model["summed_magnitude"].items():
            worst[k] = max(worst.get(k, 0.0), BB.magnitude_error(model[k], exact[k], m))
    assert 0.1 * BB.U < max(worst.values()) <= BB.EXACT_TOL, worst
