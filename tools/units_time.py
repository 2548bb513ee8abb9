"""Time of the per-copy stream builder (flipout_units_kernel) in one Bayes bf16 step: torch.profiler sum over the
step's launches.  FPN_LIB_PATH selects the build.   python tools/units_time.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402

import bench  # noqa: E402

sys.argv = [sys.argv[0], "--config", "bayes", "--dtype", "bf16", "--no-pointset", "--no-cpu-baseline"]
args = bench.parse_args()
wl = bench.build_bayes(args, torch.device("cuda:0"), 0, 1)
step = wl["step"]
for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(4):
        step()
    torch.cuda.synchronize()
for e in prof.key_averages():
    if "flipout_units" in e.key:
        t = getattr(e, "self_device_time_total", None) or e.self_cuda_time_total
        print("flipout_units_kernel: %.1f us a step in %.0f launches" % (t / 4, e.count / 4))
