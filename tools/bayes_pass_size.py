"""Flipout forward (32 samples x 128 frustums x 1024 points) against the number of frustum copies per pass of
the chain kernels.   FPN_PRECISION=bf16|fp32 python tools/bayes_pass_size.py [copies ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402

from fpn import bayes, synthetic, variables  # noqa: E402

dev = torch.device("cuda:0")
pc, oh = synthetic.make_frustums(128, 1024, 3, seed=0)
pc, oh = torch.from_numpy(pc).to(dev), torch.from_numpy(oh).to(dev)
for copies in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096]:
    st = variables.VariableStore(dev, seed=1)
    bp = bayes.BayesPlan(st, 3, mc_dropout=True, precision=os.environ.get("FPN_PRECISION", "bf16"), chunk_frustums=copies)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    for _ in range(2):
        bp.forward(pc, oh, 32, generator=gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        bp.forward(pc, oh, 32, generator=gen)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("copies per pass %5d: %.2f ms per forward, %.1f k frustums/s" % (copies, ms, 128 / ms), flush=True)
    del bp, st
    torch.cuda.empty_cache()
