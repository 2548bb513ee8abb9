"""A few launches of the bf16 seg tail (shared stream, 1024 x 1024 points) in one form, for counter passes.
   python tools/tail_one.py <layers|fused1|fused2> <ring|lds>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
os.environ["FPN_BF16_TAIL"], os.environ["FPN_BF16_FEED"] = sys.argv[1], sys.argv[2]
import torch  # noqa: E402

from fpn import ops, plan, variables  # noqa: E402

dev = torch.device("cuda:0")
b, n = 1024, 1024
st = variables.VariableStore(dev, seed=1)
plan.init_variables(st, 4, True)
st.randomize_batch_norm(2)
net = plan.V1Plan(st, 4, True, precision="bf16")
point_feat = (torch.randn((b * n, 64), device=dev) * 0.5).to(torch.bfloat16)
gbias = torch.randn((b, 512), device=dev)
logits = torch.empty((b, n, 2), device=dev)
for _ in range(6):
    ops.v1_seg_tail_bf16(point_feat, n, gbias, net._tail_ws, net._tail_consts, logits)
torch.cuda.synchronize()
