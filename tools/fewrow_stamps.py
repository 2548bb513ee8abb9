"""Cycle stamps of one wave of fewrow_mlp_kernel on the 1027 -> 512 layer of a v1 step (a library built with
-DFPN_FEWROW_STAMPS, FPN_VARIANT_UNIT=mlp tools/build_variant.sh; FPN_LIB_PATH).  python tools/fewrow_stamps.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import ops  # noqa: E402

dev = torch.device("cuda:0")
rows, cin, cout = 1024, int(os.environ.get("CIN", 1027)), int(os.environ.get("COUT", 512))
ldx = int(os.environ.get("LDX", cin + (-cin) % 4))       # row stride of x in floats
x = torch.randn((rows, ldx), device=dev)[:, :cin]
w = torch.randn((cin, cout), device=dev)
names = ["ring start", "K loop", "partials to LDS", "barrier + sum", "epilogue"]
for _ in range(4):
    y = ops.shared_mlp(x, w, None, torch.zeros(cout, device=dev), relu=False)
    torch.cuda.synchronize()
    t = y.reshape(-1)[:12].view(torch.int64).cpu().tolist()
    print("total %d  " % (t[5] - t[0]) + "  ".join("%s %d" % (n, t[i + 1] - t[i]) for i, n in enumerate(names)))
