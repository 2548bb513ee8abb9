"""The frustum-sharded multi-process path on CPU: two `gloo` ranks shard a batch,
build per-frustum result records, and all-gather them (the same fpn.dist code the
benchmark runs over RCCL with one rank per GPU)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port nobody listens on right now (two suites on one host do not collide)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


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
        # the asynchronous form (gloo runs its operations on worker threads in no fixed order:
        # one gather into a buffer at a time; RCCL orders them on its stream)
        buf = torch.empty_like(full)
        _, w1 = fdist.gather_results(rec + 1.0, out=buf, async_op=True)
        w1.wait()
        assert torch.equal(buf, full + 1.0)
        out2, w2 = fdist.gather_results(rec, out=buf, async_op=True)
        w2.wait()
        assert out2 is buf and torch.equal(buf, full)
        st_out, w3 = fdist.gather_statistics(ids[:, None] + torch.zeros(fdist.STATS)[None], async_op=True)
        w3.wait()
        assert torch.equal(st_out[:, 0], torch.arange(total, dtype=torch.float32))
    # Monte-Carlo ensembles: the per-frustum statistics tables are what travels
    stats = ids[:, None] * 2 + torch.arange(fdist.STATS)[None] / 10.0
    full_s = fdist.gather_statistics(stats, total)
    if total % world == 0:
        assert torch.equal(fdist.gather_statistics(stats), full_s)
    # finished detections: float64 records (the kernel that fills them needs a GPU; the
    # gather is dtype-agnostic)
    det = ids[:, None].double() * 3 + torch.arange(fdist.DETECTION, dtype=torch.float64)[None] / 7.0
    full_d = fdist.gather_detections(det, total)
    if total % world == 0:
        assert torch.equal(fdist.gather_detections(det), full_d)
        out_d, w4 = fdist.gather_detections(det, async_op=True)
        w4.wait()
        assert torch.equal(out_d, full_d)
    torch.save({"full": full, "full_l": full_l, "full_s": full_s, "full_d": full_d},
               os.path.join(out_dir, "rank%d.pt" % rank))
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
        want_s = torch.arange(total, dtype=torch.float32)[:, None] * 2 + torch.arange(fdist.STATS)[None] / 10.0
        assert torch.equal(o["full_s"], want_s)
        want_d = torch.arange(total, dtype=torch.float64)[:, None] * 3 + \
            torch.arange(fdist.DETECTION, dtype=torch.float64)[None] / 7.0
        assert o["full_d"].dtype == torch.float64 and torch.equal(o["full_d"], want_d)
    assert torch.equal(outs[0]["full"], outs[1]["full"])


def test_two_rank_gather_even(tmp_path):
    _run(10, tmp_path, _free_port())


def test_two_rank_gather_ragged(tmp_path):
    _run(7, tmp_path, _free_port())


def _toy_forward(pc, oh):
    """Per-frustum stand-in for the network: every output row depends on its own frustum only."""
    s = pc.sum(dim=(1, 2))
    return {"box_output": s[:, None] + torch.arange(59, dtype=torch.float32)[None] + oh[:, :1],
            "stage1_center": pc[:, 0, :3] * 2, "mask_xyz_mean": pc.mean(1)[:, :3],
            "mask_logits": pc[:, :, :2].contiguous()}


def _stream_worker(rank, world, port, total, chunk, out_dir):
    for p in (ROOT, os.path.join(ROOT, "3d-object-detection_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fpn import dist as fdist
    rng = np.random.RandomState(0)                      # the same scene set on every rank
    pc = rng.standard_normal((total, 8, 4)).astype(np.float32)
    oh = np.eye(3, dtype=np.float32)[rng.randint(0, 3, total)]
    full = fdist.run_sharded_streamed(_toy_forward, pc, oh, chunk, "cpu", with_logits=True)
    torch.save(full, os.path.join(out_dir, "stream%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_streamed_scene_set(tmp_path):
    """Scene set of 11 frustums, 2 ranks, chunks of 3: every rank streams its block and all
    ranks end with the records of the whole set in frustum order."""
    total, chunk = 11, 3
    mp.spawn(_stream_worker, args=(2, _free_port(), total, chunk, str(tmp_path)), nprocs=2, join=True)
    from fpn import dist as fdist
    rng = np.random.RandomState(0)
    pc = torch.from_numpy(rng.standard_normal((total, 8, 4)).astype(np.float32))
    oh = torch.from_numpy(np.eye(3, dtype=np.float32)[rng.randint(0, 3, total)])
    want = fdist.pack_results(_toy_forward(pc, oh), with_logits=True)
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), "stream%d.pt" % r))
        assert torch.equal(got, want)
    # single process, no process group: the same call degenerates to a plain chunked loop
    assert torch.equal(fdist.run_sharded_streamed(_toy_forward, pc.numpy(), oh.numpy(), 4, "cpu", True), want)
    assert fdist.run_streamed(_toy_forward, pc[:0], oh[:0], 4, "cpu").shape == (0, fdist.RECORD)
