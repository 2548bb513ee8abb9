"""First contact of the multi-rank code with RCCL on a one-GPU box: a world of ONE rank on
backend "nccl" (RCCL refuses two ranks on one device), driving the same calls the bench
and fpn.dist make on N GPUs: init with device_id, barrier, all_gather_into_tensor of the
result records and of the statistics table, all_reduce(MAX) of the step time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-object-detection_amd")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
    from fpn import dist as fdist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    dist.init_process_group("nccl", device_id=dev)
    print("init %.1f s" % (time.time() - t0), flush=True)
    dist.barrier(device_ids=[0])
    rec = torch.randn((1024, fdist.RECORD), device=dev)
    out = torch.empty_like(rec)
    t0 = time.time()
    dist.all_gather_into_tensor(out, rec)
    torch.cuda.synchronize()
    print("first all_gather %.1f s" % (time.time() - t0), flush=True)
    assert torch.equal(out, rec)
    # the bench's pattern: asynchronous gathers into one buffer beside further work on the
    # compute stream, the last one waited for at the end
    work = None
    for i in range(5):
        rec_i = rec + float(i)
        work = dist.all_gather_into_tensor(out, rec_i, async_op=True)
        rec = rec * 1.0
    work.wait()
    torch.cuda.synchronize()
    assert torch.equal(out, rec + 4.0)
    stats = torch.randn((64, fdist.STATS), device=dev)
    out2 = torch.empty_like(stats)
    dist.all_gather_into_tensor(out2, stats)
    assert torch.equal(out2, stats)
    tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt.item()) == 1.25
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    print("rccl contact ok", flush=True)


if __name__ == "__main__":
    main()
