"""Times the v1 forward at both chain precisions and prints per-kernel times (the
benchmark's kernel timer).  Usage: python tools/bf16_time.py [frustums]"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "3d-object-detection_amd"))
sys.path.insert(0, os.path.join(HERE, ".."))
import bench  # noqa: E402
from fpn import ops, plan, synthetic, variables  # noqa: E402


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024   # (FPN_LIB_PATH selects another build of the library)
    dev = torch.device("cuda", 0)
    st = variables.VariableStore(dev, seed=1)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(2)
    pc_np, oh_np = synthetic.make_frustums(b, 1024, 4, seed=0)
    pc, oh = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
    for precision in ("fp32", "bf16"):
        net = plan.V1Plan(st, 4, True, precision=precision)
        for _ in range(3):
            net.forward(pc, oh)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            net.forward(pc, oh)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 10
        timer = bench.KernelTimer()
        ops.launch_observer = timer
        for _ in range(5):
            net.forward(pc, oh)
        torch.cuda.synchronize()
        ops.launch_observer = None
        agg = timer.summary()
        print("%s: %.3f ms/step, %.0f frustums/s" % (precision, ms, b / ms * 1e3), flush=True)
        for k, (n, fl, s) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            print("   %-26s %7.3f ms/launch  %8.1f TFLOP/s" % (k, s / n * 1e3, fl / s / 1e12 if s else 0), flush=True)


if __name__ == "__main__":
    main()
