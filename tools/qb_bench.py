"""Ball-query launch times at the v2 shapes (HIP events over 50 queued launches)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import synthetic  # noqa: E402
from fpn.tf_ops import tf_grouping, tf_sampling  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    for b in (256, 4096):
        for kind in ("metric", "unit"):
            if kind == "metric":
                xyz = torch.from_numpy(synthetic.make_frustums_range(0, b, 2048, 3, seed=0)[0]).to(dev)
            else:
                xyz = torch.from_numpy(synthetic.unit_cube(b, 2048)).to(dev)
            new_xyz = tf_sampling.gather_point(xyz, tf_sampling.farthest_point_sample(128, xyz))
            new2 = tf_sampling.gather_point(new_xyz, tf_sampling.farthest_point_sample(32, new_xyz))
            out = []
            for (r, ns) in ((0.2, 32), (0.4, 64), (0.8, 128)):
                out.append("%.1f" % timed(lambda: tf_grouping.query_ball_point(r, ns, xyz, new_xyz)))
            for (r, ns) in ((0.4, 64), (1.6, 128)):
                out.append("%.1f" % timed(lambda: tf_grouping.query_ball_point(r, ns, new_xyz, new2)))
            print("b=%4d %-6s us per launch: n2048 r0.2/0.4/0.8 | n128 r0.4/1.6:  %s" % (b, kind, "  ".join(out)))


if __name__ == "__main__":
    main()
