"""Time of the fully connected layers (fewrow_mlp_kernel) in one step of a bench.py workload: torch.profiler sum.
python tools/fc_time.py [bench.py flags]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402

import bench  # noqa: E402

sys.argv = [sys.argv[0], "--no-pointset", "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse_args()
wl = {"v1": bench.build_v1, "v2": bench.build_v2, "bayes": bench.build_bayes}[args.config](args, torch.device("cuda:0"), 0, 1)
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
    if "fewrow" in e.key:
        t = getattr(e, "self_device_time_total", None) or e.self_cuda_time_total
        print("%s: %.1f us a step in %.0f launches" % (e.key[:70], t / 4, e.count / 4))
# the launches of the last step, one by one, in launch order
ev = sorted((e for e in prof.events() if "fewrow" in e.name and e.device_type == torch.autograd.DeviceType.CUDA),
            key=lambda e: e.time_range.start)
per = len(ev) // 4
print("last step:", " ".join("%.1f" % (e.time_range.elapsed_us()) for e in ev[-per:]))
