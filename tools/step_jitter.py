"""Per-step wall time of the v2 forward with the allocator's device-allocation count and the
host's share (time to enqueue a step without waiting), to tell GPU-bound steps from
host-bound or allocator-bound ones.  Usage: python tools/step_jitter.py [bf16|fp32] [frustums] [observer-name part to bracket with events]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import ops, pointnet_util, synthetic, variables  # noqa: E402
from fpn.models import frustum_pointnets_v2 as M2  # noqa: E402


class Bracket:
    """HIP-event pairs around the launches whose observer name contains `part`."""

    def __init__(self, part):
        self.part, self.pending = part, []

    def begin(self, name, rows, cin, cout, macs=None):
        if self.part not in name:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, e0, e1)

    def end(self, tok):
        if tok is not None:
            tok[2].record()
            self.pending.append(tok)


def main():
    dev = torch.device("cuda:0")
    pointnet_util.SA_PRECISION = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    pc_np, oh_np = synthetic.make_frustums_range(0, b, 2048, 4, seed=0)
    pc, oh = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
    st = variables.VariableStore(dev, seed=1)
    with variables.use_store(st):
        M2.get_model(pc[:2], oh[:2], False)
        for _ in range(5):
            M2.get_model(pc, oh, False)
        torch.cuda.synchronize()
        if len(sys.argv) > 4:       # one step with events around every launch first (bench.py's probe pass)
            probe = Bracket("")
            ops.launch_observer = probe
            M2.get_model(pc, oh, False)
            ops.launch_observer = None
            torch.cuda.synchronize()
            print("probe pass: %d bracketed launches, %.2f ms inside the brackets" % (
                len(probe.pending), sum(e0.elapsed_time(e1) for _, e0, e1 in probe.pending)))
        if len(sys.argv) > 3:
            ops.launch_observer = Bracket(sys.argv[3])
        for i in range(12):
            a0 = torch.cuda.memory_stats()["num_device_alloc"]
            t0 = time.perf_counter()
            M2.get_model(pc, oh, False)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if i == 11:
                # twenty steps queued back to back, one wait at the end (how bench.py times a region)
                t0 = time.perf_counter()
                for _ in range(20):
                    M2.get_model(pc, oh, False)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                print("20 queued steps: enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) * 50, (t2 - t0) * 50))
            print("step %2d: enqueue %.2f ms, total %.2f ms, device allocations %d, reserved %.2f GB" % (
                i, (t1 - t0) * 1e3, (t2 - t0) * 1e3, torch.cuda.memory_stats()["num_device_alloc"] - a0,
                torch.cuda.memory_reserved() / 2 ** 30))


if __name__ == "__main__":
    main()
