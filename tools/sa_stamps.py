"""Where a wave of the fused set-abstraction kernel (bf16) spends its cycles: run the first
multi-scale module of the v2 config with a library built with -DFPN_SA_STAMPS
(tools/build_variant.sh; FPN_LIB_PATH) and decode the cycle stamps one wave left in the
first output row of each scale."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import pointnet_util, synthetic, variables  # noqa: E402
from fpn.models import frustum_pointnets_v2 as M2  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    pointnet_util.SA_PRECISION = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    pc_np, _ = synthetic.make_frustums_range(0, 256, 2048, 4, seed=0)
    pc = torch.from_numpy(pc_np).to(dev)
    xyz, feat = pc[:, :, :3].contiguous(), pc[:, :, 3:].contiguous()
    s = M2.CONFIG["seg"]["sa1"]
    st = variables.VariableStore(dev, seed=1)
    names = ["start", "gathered", "first layer", "layer 2", "layer 3 + max", "flushed"]
    with variables.use_store(st):
        for i in range(len(s["radius"])):
            for _ in range(3):
                _, out = pointnet_util.pointnet_sa_module_msg(xyz, feat, s["npoint"], [s["radius"][i]], [s["nsample"][i]],
                                                              [s["mlp"][i]], False, None, scope="probe%d" % i)
            torch.cuda.synchronize()
            t = out.reshape(-1)[:12].view(torch.int64).cpu().tolist()
            print("scale %d %s ns=%d:" % (i, s["mlp"][i], s["nsample"][i]),
                  ", ".join("%s +%d" % (names[k], t[k] - t[k - 1]) for k in range(1, 6)), "| total", t[5] - t[0])
        # the second multi-scale module: 128 points with 320 features each, 32 balls
        s2 = M2.CONFIG["seg"]["sa2"]
        gen = torch.Generator(device=dev)
        gen.manual_seed(1)
        xyz2 = xyz[:, :128].contiguous()
        feat2 = torch.randn((xyz.shape[0], 128, 320), device=dev, generator=gen)
        for i in range(len(s2["radius"])):
            for _ in range(3):
                _, out = pointnet_util.pointnet_sa_module_msg(xyz2, feat2, s2["npoint"], [s2["radius"][i]], [s2["nsample"][i]],
                                                              [s2["mlp"][i]], False, None, scope="probe2%d" % i)
            torch.cuda.synchronize()
            t = out.reshape(-1)[:12].view(torch.int64).cpu().tolist()
            print("sa2 scale %d %s ns=%d:" % (i, s2["mlp"][i], s2["nsample"][i]),
                  ", ".join("%s +%d" % (names[k], t[k] - t[k - 1]) for k in range(1, 6)), "| total", t[5] - t[0])


if __name__ == "__main__":
    main()
