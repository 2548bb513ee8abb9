"""Per-step launch table of one bench.py workload: after warm-up, N steps under
torch.profiler; prints every device kernel / copy with launches per step and microseconds
per step, and the wall time of a step.  Usage:
    python tools/step_trace.py [bench.py flags, e.g. --config v2 --dtype bf16] [--trace-steps 4]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace-steps", type=int, default=4)
    ap.add_argument("--rows", type=int, default=60)
    mine, rest = ap.parse_known_args()
    sys.argv = [sys.argv[0]] + rest
    args = bench.parse_args()
    dev = torch.device("cuda:0")
    wl = {"v1": bench.build_v1, "v2": bench.build_v2, "bayes": bench.build_bayes}[args.config](args, dev, 0, 1)
    step = wl["step"]
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(mine.trace_steps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / mine.trace_steps
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(mine.trace_steps):
            step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        dt = getattr(e, "self_device_time_total", None)
        if dt is None:
            dt = e.self_cuda_time_total
        if dt > 0 and e.device_type == torch.autograd.DeviceType.CUDA:
            rows.append((dt / mine.trace_steps, e.count / mine.trace_steps, e.key))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    launches = sum(r[1] for r in rows)
    print("wall %.1f us/step   device total %.1f us/step   %.1f launches/step" % (wall * 1e6, total, launches))
    for us, n, name in rows[:mine.rows]:
        print("%9.1f us %7.2f x  %s" % (us, n, name[:150]))
    # host-side operators that launch: where the small launches come from
    host = [(e.count / mine.trace_steps, e.key) for e in prof.key_averages()
            if e.device_type == torch.autograd.DeviceType.CPU and e.key.startswith("aten::")]
    host.sort(reverse=True)
    print("aten operators per step:", ", ".join("%s x%.1f" % (k, n) for n, k in host[:25]))


if __name__ == "__main__":
    main()
