"""A few launches of FPS (2048 -> 128, 4096 frustums) in both forms, for counter passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import synthetic  # noqa: E402
from fpn.tf_ops import tf_sampling  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
xyz = torch.from_numpy(synthetic.unit_cube(b, 2048)).to("cuda:0")
for form in ("block", "wave"):
    os.environ["FPN_FPS"] = form
    for _ in range(4):
        tf_sampling.farthest_point_sample(128, xyz)
torch.cuda.synchronize()
