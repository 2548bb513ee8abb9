"""Per-shape GPU time of the generic layer launches (fpn_shared_mlp_f32) inside a v2 step:
HIP events around every launch, aggregated by (rows, cin, cout, pool)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import ops, pointnet_util, synthetic, variables  # noqa: E402
from fpn.models import frustum_pointnets_v2 as M2  # noqa: E402


class ShapeTimer:
    def __init__(self):
        self.pending = []

    def begin(self, name, rows, cin, cout, macs=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return ("%s rows=%d cin=%d cout=%d" % (name, rows, cin, cout), 2.0 * (macs or rows * cin * cout), e0, e1)

    def end(self, tok):
        tok[3].record()
        self.pending.append(tok)


def main():
    dev = torch.device("cuda:0")
    if len(sys.argv) > 1 and sys.argv[1] == "bf16":
        pointnet_util.SA_PRECISION = "bf16"
    frustums = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    pc_np, oh_np = synthetic.make_frustums_range(0, frustums, 2048, 4, seed=0)
    pc, oh = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
    st = variables.VariableStore(dev, seed=1)
    with variables.use_store(st):
        M2.get_model(pc[:2], oh[:2], False)
        for _ in range(5):
            M2.get_model(pc, oh, False)
        t = ShapeTimer()
        ops.launch_observer = t
        steps = 5
        for _ in range(steps):
            M2.get_model(pc, oh, False)
        ops.launch_observer = None
    torch.cuda.synchronize()
    agg = {}
    for name, flops, e0, e1 in t.pending:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1)
    for name, (n, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print("%-70s %5.1f us/launch x %.0f per step  %6.1f TFLOP/s" % (name, ms / n * 1e3, n / steps, fl / ms / 1e9))


if __name__ == "__main__":
    main()
