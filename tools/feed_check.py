"""The two weight feeds of the bf16 chains do the same arithmetic in the same order: their
results must agree bit for bit.  Runs the Flipout network and the plain v1 network under
FPN_BF16_FEED=ring and under the default feed and compares."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from fpn import bayes, synthetic, variables
    dev = torch.device("cuda:0")
    bad = 0
    for B, N, dropout in [(4, 256, False), (13, 96, True), (16, 1024, False), (64, 1024, True)]:
        st = variables.VariableStore(dev, seed=4)
        bp = bayes.BayesPlan(st, 3, mc_dropout=dropout, precision="bf16")
        pc, oh = synthetic.make_frustums(B, N, 3, seed=B)
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        noise = bp.sample_noise(2, B, g, N)
        pcd, ohd = torch.from_numpy(pc).to(dev), torch.from_numpy(oh).to(dev)
        res = {}
        for feed in ("ring", "coop", "coop"):
            os.environ["FPN_BF16_FEED"] = feed
            ep = bp.forward(pcd, ohd, 2, noise=noise)
            res.setdefault(feed, []).append({k: ep[k].clone() for k in ("mask_logits", "box_output")})
        for k in ("mask_logits", "box_output"):
            a, b, c = res["ring"][0][k], res["coop"][0][k], res["coop"][1][k]
            d1 = (a != b).sum().item()
            d2 = (b != c).sum().item()
            print(B, N, dropout, k, "ring!=coop:", d1, "coop!=coop:", d2, "max abs", (a - b).abs().max().item(), flush=True)
            bad += d1 + d2
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
