"""Real-data entry into the path: frustums from the pickle-list dataset and from a record
file go through get_model on the GPU and match the CPU oracle on the same arrays."""
import numpy as np
import pytest
import torch

from oracle import v1_numpy as V
from test_frustum_records import frustum_lists

pytestmark = pytest.mark.gpu


def close(got, want, rtol=1e-3):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    s = np.sqrt(np.mean(want ** 2)) + 1e-30
    assert (np.abs(got - want) / np.maximum(np.abs(want), s)).max() <= rtol


def test_dataset_and_record_batches_through_get_model(device, tmp_path):
    from fpn import frustum_records as R
    from fpn import pkl_to_tfrec as P
    from fpn import plan, provider, variables
    ds = provider.FrustumDataset(1024, "val", rotate_to_center=True, one_hot=True, lists=frustum_lists(6, seed=2))
    np.random.seed(3)
    pc, one_hot, rest = ds.batch(range(6), device, channels=4)
    assert pc.shape == (6, 1024, 4) and one_hot.shape == (6, 3) and pc.is_cuda and len(rest) == 6
    st = variables.VariableStore(device, seed=1)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(2)
    ep = plan.V1Plan(st, 4, True).forward(pc, one_hot)
    want = V.get_model(pc.cpu().numpy(), one_hot.cpu().numpy(), st.numpy())
    close(ep["mask_logits"].cpu().numpy(), want["mask_logits"])

    # record file written by the Lyft generator (xyz only), read back in batches of 4
    path = str(tmp_path / "t.tfrec")
    np.random.seed(4)
    P.tfrecGen_test(lists=frustum_lists(6, seed=5, lyft=True, rgb=True)).write_tfrec(path)
    st3 = variables.VariableStore(device, seed=2)
    plan.init_variables(st3, 3, True)
    net3 = plan.V1Plan(st3, 3, True)
    sizes = []
    for pc, one_hot, rest in R.frustum_batches(path, 4, device, test=True):
        sizes.append(pc.shape[0])
        ep = net3.forward(pc, one_hot)
        torch.cuda.synchronize()
        want = V.get_model(pc.cpu().numpy(), one_hot.cpu().numpy(), st3.numpy())
        close(ep["mask_logits"].cpu().numpy(), want["mask_logits"])
    assert sizes == [4, 2]


def test_detection_driver_on_the_device_model(device, tmp_path):
    """scripts/test.py's loop with the real detection model: record file -> batches ->
    det_model.predict -> result lists; the lists equal the oracle's network outputs
    post-processed by the same formulas."""
    from fpn import inference, variables
    from fpn.models import FPointNet_Keras as K
    from test_detection_driver import write_test_records
    path = str(tmp_path / "test.tfrec")
    rows = write_test_records(path, 8, seed=3)
    st = variables.VariableStore(device, seed=5)
    with variables.use_store(st):
        _, det = K.Frustum_Pointnet_Model(batch_size=4)
        lists = inference.test(det, inference.make_test_iterator(path, 4), str(tmp_path / "det.pkl"),
                               result_dir=str(tmp_path / "res"), csv_path=None)
    assert len(lists["ps"]) == 8
    pc = np.stack([r[0] for r in rows])
    oh = np.stack([r[1] for r in rows]).astype(np.float32)
    want = V.get_model(pc, oh, st.numpy(), bn=False)
    seg, c, hc, hr, sc, sr, s = inference.postprocess(want["mask_logits"], want["center"], want["heading_scores"],
                                                      want["heading_residuals"], want["size_scores"],
                                                      want["size_residuals"])
    assert (np.stack(lists["segp"]) != seg).mean() < 0.01          # flips only where the two logits tie
    close(np.stack(lists["center"]), c)
    np.testing.assert_array_equal(np.asarray(lists["heading_cls"]), hc)
    np.testing.assert_array_equal(np.asarray(lists["size_cls"]), sc)
    close(np.asarray(lists["heading_res"]), hr)
    close(np.stack(lists["size_res"]), sr)
    close(np.asarray(lists["score"]), s)


@pytest.mark.parametrize("pinned", [False, True])
def test_streamed_scene_set_equals_one_batch(device, pinned):
    """fpn.dist.run_streamed: chunks with copies overlapped on a second stream give the
    records of the one-batch forward bit for bit (frustums are independent), ragged last
    chunk, more chunks than slots, pinned and pageable sources."""
    from fpn import dist as fdist
    from fpn import plan, synthetic, variables
    st = variables.VariableStore(device, seed=1)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(2)
    net = plan.V1Plan(st, 4, True)
    pc, oh = synthetic.make_frustums(11, 256, 4, seed=8)
    want = fdist.pack_results(net.forward(torch.from_numpy(pc).to(device), torch.from_numpy(oh).to(device)), True)
    src_pc, src_oh = torch.from_numpy(pc), torch.from_numpy(oh)
    if pinned:
        src_pc, src_oh = src_pc.pin_memory(), src_oh.pin_memory()
    for chunk in (3, 4, 16):
        got = fdist.run_streamed(net.forward, src_pc, src_oh, chunk, device, with_logits=True)
        assert torch.equal(got, want), chunk
    with pytest.raises(ValueError):
        fdist.run_streamed(net.forward, torch.from_numpy(pc).to(device), src_oh, 4, device)
