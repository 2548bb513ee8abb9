"""The three fully connected stacks of the v1 network, one fpn_shared_mlp_f32 call per layer
(HIP events over 50 repetitions; run under tools/fc_trace.sh for GPU-side durations per
launch: events around single small launches measure the host's launch rate)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import ops  # noqa: E402


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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    for name, widths in (("tnet", (259, 256, 128, 3)), ("box", (515, 512, 256, 59)), ("conv6 frustum part", (1027, 512))):
        layers = []
        for cin, cout in zip(widths[:-1], widths[1:]):
            layers.append((torch.randn((cin, cout), device=dev) * cin ** -0.5, torch.rand(cout, device=dev) + 0.5,
                           torch.randn(cout, device=dev) * 0.1, True))
        x = torch.randn((rows, widths[0]), device=dev)

        def per_layer():
            a = x
            for w, s, h, r in layers:
                a = ops.shared_mlp(a, w, s, h, relu=r)
            return a
        macs = rows * sum(a * b for a, b in zip(widths[:-1], widths[1:]))
        t_layers = timed(per_layer)
        print("%-20s rows %5d: %7.1f us for %d launches (%5.1f TFLOP/s)" % (name, rows, t_layers, len(layers), 2 * macs / t_layers / 1e6))


if __name__ == "__main__":
    main()
