"""Times of the bf16 seg tail in its forms (FPN_BF16_TAIL x FPN_BF16_FEED), shared stream and
one stream per frustum copy: HIP events over queued launches of the tail alone.
   python tools/tail_forms.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import ops, plan, synthetic, variables  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    b, n = 1024, 1024
    st = variables.VariableStore(dev, seed=1)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(2)
    net = plan.V1Plan(st, 4, True, precision="bf16")
    rows = b * n
    point_feat = (torch.randn((rows, 64), device=dev) * 0.5).to(torch.bfloat16)
    gbias = torch.randn((b, 512), device=dev)
    logits = torch.empty((b, n, 2), device=dev)
    # per-copy streams: every frustum its own copy of the shared stream
    shared = net._tail_ws
    per = (shared.numel() - ops._lib.load().fpn_chain_pad_units_bf16() * 2048)
    copies = torch.cat([shared[:per].repeat(b), shared[per:]])
    w10 = net.p["conv10"][0]
    w10c = w10.repeat(b, 1).contiguous()
    consts_pc = ops.pointer_table([net.p[s][2] for s in ("conv7", "conv8", "conv9")] + [w10c, net.p["conv10"][2]])
    flop = rows * 2.0 * (64 * 512 + 512 * 256 + 256 * 128 + 128 * 128 + 128 * 2)
    variants = [("layers", "ring"), ("layers", "lds"), ("fused1", "ring"), ("fused1", "lds"), ("fused2", "ring")]
    for form, feed in variants:
            os.environ["FPN_BF16_TAIL"], os.environ["FPN_BF16_FEED"] = form, feed
            t = timed(lambda: ops.v1_seg_tail_bf16(point_feat, n, gbias, net._tail_ws, net._tail_consts, logits))
            tc = timed(lambda: ops.v1_seg_tail_bf16(point_feat, n, gbias, copies, consts_pc, logits, copy_vecs=per // 8))
            print("%-7s %-4s  shared stream %.3f ms (%.0f TFLOP/s, %.2f of 2516.6)   per-copy streams %.3f ms (%.2f)" % (
                form, feed, t, flop / t / 1e9, flop / t / 1e9 / 2516.6, tc, flop / tc / 1e9 / 2516.6), flush=True)


if __name__ == "__main__":
    main()
