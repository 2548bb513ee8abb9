"""Where a wave of the bf16 seg tail spends its cycles: one v1 forward with a library built
with -DFPN_TAIL_STAMPS (tools/build_variant.sh; FPN_LIB_PATH), then the cycle stamps one wave
left in the logits buffer.  FPN_BF16_FEED=ring selects the per-wave register ring."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import plan, synthetic, variables  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    pc_np, oh_np = synthetic.make_frustums_range(0, 1024, 1024, 4, seed=0)
    pc, oh = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
    st = variables.VariableStore(dev, seed=1)
    pl = plan.V1Plan(st, 4, precision="bf16")
    fused = os.environ.get("FPN_BF16_TAIL", "fused2") != "layers"
    names = ["start", "opening (constants, inputs, first units)", "conv6 + conv7, pass 1", "pass 2", "(stamp)", "conv8 (16 units)",
             "conv9 (8)", "dropout + conv10 + store"] if fused else \
        ["start", "inputs + first units", "conv6 (16 units)", "conv7 (64)", "conv8 (16)", "conv9 (8)", "conv10 + store"]
    for _ in range(3):
        ep = pl.forward(pc, oh)
    torch.cuda.synchronize()
    n = len(names)
    t = ep["mask_logits"].reshape(-1)[64:64 + 2 * n].view(torch.int64).cpu().tolist()
    print(os.environ.get("FPN_BF16_TAIL", "fused2"), os.environ.get("FPN_BF16_FEED", "ring"),
          ", ".join("%s +%d" % (names[k], t[k] - t[k - 1]) for k in range(1, n)), "| total", t[n - 1] - t[0])

if __name__ == "__main__":
    main()
