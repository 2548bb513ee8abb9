"""Host-to-device copy rate of this box from pinned memory (what bounds a streamed scene set)."""
import time

import torch

dev = torch.device("cuda:0")
for mb in (32, 128, 512):
    src = torch.empty(mb * (1 << 20) // 4, dtype=torch.float32).pin_memory()
    dst = torch.empty_like(src, device=dev)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%4d MiB: %.2f GB/s" % (mb, mb * (1 << 20) / dt / 1e9), flush=True)
