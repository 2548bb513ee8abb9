"""Per-layer timing of fpn_shared_mlp_f32 on the v1 layer shapes (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch
from fpn import ops

dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 1024
shapes = [("conv1", F * N, 4, 64, 0), ("conv2", F * N, 64, 64, 0), ("conv4", F * N, 64, 128, 0),
          ("conv5+pool", F * N, 128, 1024, N), ("conv6pt", F * N, 64, 512, 0), ("conv7", F * N, 512, 256, 0),
          ("conv8", F * N, 256, 128, 0), ("conv9", F * N, 128, 128, 0), ("conv10", F * N, 128, 2, 0),
          ("glob", F, 1027, 512, 0), ("t1", F * 512, 3, 128, 0), ("t3+pool", F * 512, 128, 256, 512),
          ("b4+pool", F * 512, 256, 512, 512), ("fc1", F, 515, 512, 0)]
for name, rows, cin, cout, pool in shapes:
    x = torch.randn(rows, cin, device=dev)
    w = torch.randn(cin, cout, device=dev) / cin ** 0.5
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    for _ in range(3):
        ops.shared_mlp(x, w, sc, sh, relu=True, pool_rows=pool)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.shared_mlp(x, w, sc, sh, relu=True, pool_rows=pool)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * rows * cin * cout
    byt = 4.0 * (rows * cin + (0 if pool else rows * cout) + cin * cout)
    print("%-12s rows=%8d %4d->%4d  %8.3f ms  %7.2f TFLOP/s  %7.1f GB/s" % (name, rows, cin, cout, ms, fl / ms / 1e9, byt / ms / 1e6))
