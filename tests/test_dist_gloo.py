"""The frustum-sharded multi-process path on CPU: two `gloo` ranks shard a batch,
build per-frustum result records, and all-gather them (the same fpn.dist code the
benchmark runs over RCCL with one rank per GPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, out_dir):
    for p in (ROOT, os.path.join(ROOT, "3d-object-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fpn import dist as fdist
    lo, hi = fdist.shard_range(total, rank, world)
    # a deterministic stand-in for the network: record of frustum i is i + column/100
    ids = torch.arange(lo, hi, dtype=torch.float32)
    ep = {"box_output": ids[:, None] + torch.arange(59)[None] / 100.0,
          "stage1_center": ids[:, None].repeat(1, 3) + 0.5,
          "mask_xyz_mean": ids[:, None].repeat(1, 3) + 0.25,
          "mask_logits": ids[:, None, None].repeat(1, 4, 2)}
    rec = fdist.pack_results(ep)
    full = fdist.gather_ragged(rec, total)
    rec_l = fdist.pack_results(ep, with_logits=True)
    full_l = fdist.gather_ragged(rec_l, total)
    if total % world == 0:
        same = fdist.gather_results(rec)
        assert torch.equal(same, full)
    torch.save({"full": full, "full_l": full_l}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _run(total, tmp_path, port):
    world = 2
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    from fpn import dist as fdist
    for o in outs:
        full = o["full"]
        assert full.shape == (total, fdist.RECORD)
        back = fdist.unpack_results(full)
        np.testing.assert_array_equal(back["box_output"][:, 0].numpy(), np.arange(total, dtype=np.float32))
        np.testing.assert_array_equal(back["stage1_center"][:, 1].numpy(), np.arange(total) + 0.5)
        lg = fdist.unpack_results(o["full_l"], 4)["mask_logits"]
        assert lg.shape == (total, 4, 2) and torch.equal(lg[:, 0, 0], torch.arange(total, dtype=torch.float32))
    assert torch.equal(outs[0]["full"], outs[1]["full"])


def test_two_rank_gather_even(tmp_path):
    _run(10, tmp_path, 29511)


def test_two_rank_gather_ragged(tmp_path):
    _run(7, tmp_path, 29512)
