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
