"""Frustum-sharded data parallelism: one process per GPU, no exchange during
compute, one all-gather of the per-frustum result records at the end.

Frustums are independent at inference (batch norm uses stored statistics, every
max-pool is per frustum), so rank r of W simply owns the contiguous block
[r*ceil(B/W), ...) of the batch with the weights replicated.  The only collective
is `all_gather_into_tensor` of a packed (frustums, RECORD) float32 buffer - with
backend "nccl" that is RCCL over xGMI, a fan-in over point-to-point links; there
is no reduction anywhere, so no ring all-reduce.
"""
import torch
import torch.distributed as dist

# box vector (59) | stage1_center (3) | mask_xyz_mean (3)
RECORD = 59 + 3 + 3
# detection record (fpn_detection_records): h w l tx ty tz ry score | classes, residuals,
# centre, positive count, class probabilities - float64
DETECTION = 20


def shard_range(total, rank, world):
    """Contiguous block of `total` items owned by `rank` (last blocks may be short/empty)."""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def pack_results(end_points, with_logits=False):
    """end_points -> (B, RECORD [+ 2N]) float32 record per frustum."""
    parts = [end_points["box_output"], end_points["stage1_center"], end_points["mask_xyz_mean"]]
    if with_logits:
        b = end_points["mask_logits"].shape[0]
        parts.append(end_points["mask_logits"].reshape(b, -1))
    return torch.cat(parts, dim=1).contiguous()


def pack_detections(end_points, rot_angle=None):
    """end_points -> (B, DETECTION) float64 detection records (label fields, score, picks):
    the finished detections, 160 bytes per frustum, computed on the device from the logits
    the raw record would have to carry along (8-16 KB per frustum) to yield a score."""
    from . import inference
    rec, _ = inference.detection_records(end_points["mask_logits"].contiguous(), end_points["box_output"].contiguous(),
                                         end_points["stage1_center"].contiguous(), rot_angle)
    return rec


def gather_detections(records, total=None, group=None, out=None, async_op=False):
    """One all-gather of the ranks' detection records -> (total, DETECTION) float64 on
    every rank in global frustum order.  `total`: size of the global batch when the
    shards are ragged (shard_range blocks)."""
    if records.dim() != 2 or records.shape[1] != DETECTION:
        raise ValueError("expected a (frustums, %d) detection table, got %s" % (DETECTION, tuple(records.shape)))
    if total is None:
        return gather_results(records.contiguous(), group, out, async_op=async_op)
    if async_op:
        raise ValueError("the ragged detection gather has no asynchronous form")
    return gather_ragged(records.contiguous(), total, group)


def unpack_results(records, num_point=None):
    out = {"box_output": records[:, :59], "stage1_center": records[:, 59:62],
           "mask_xyz_mean": records[:, 62:65]}
    if records.shape[1] > RECORD:
        out["mask_logits"] = records[:, RECORD:].reshape(records.shape[0], num_point, 2)
    return out


def gather_results(records, group=None, out=None, async_op=False):
    """All-gather equal-sized per-rank record blocks -> (world*B_local, width) on every
    rank, rank-major (so with shard_range the global frustum order is preserved).
    async_op=True returns (out, work): the collective runs on the backend's own stream and the
    caller's stream does not wait for it - the next batch's kernels overlap the gather; call
    work.wait() (None for a single rank) before reading `out`.  RCCL runs successive gathers
    in issue order on its stream, so they may share one `out`; a backend without that order
    (gloo) needs the previous one waited for first."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return (records, None) if async_op else records
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * records.shape[0], records.shape[1]), dtype=records.dtype,
                          device=records.device)
    work = dist.all_gather_into_tensor(out, records, group=group, async_op=async_op)
    return (out, work) if async_op else out


def gather_ragged(records, total, group=None):
    """All-gather when `total` is not a multiple of the world size: pad every block
    to ceil(total/world) rows, gather, drop the padding."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return records
    world = dist.get_world_size(group)
    per = (total + world - 1) // world
    padded = torch.zeros((per, records.shape[1]), dtype=records.dtype, device=records.device)
    padded[:records.shape[0]] = records
    full = gather_results(padded, group)
    if per * world == total:
        return full
    keep = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        keep.append(full[r * per:r * per + (hi - lo)])
    return torch.cat(keep, dim=0)


STATS = 30      # width of the Monte-Carlo ensemble statistics table (fpn_mc_statistics)


def gather_statistics(stats, total=None, group=None, out=None, async_op=False):
    """Monte-Carlo ensembles: every rank keeps all samples of its own frustums, reduces them
    to the (frustums, STATS) statistics table on its device and only that table travels:
    one all-gather -> (total, STATS) on every rank in global frustum order.  `total`: size
    of the global batch when the shards are ragged (shard_range blocks)."""
    if stats.dim() != 2 or stats.shape[1] != STATS:
        raise ValueError("expected a (frustums, %d) statistics table, got %s" % (STATS, tuple(stats.shape)))
    if total is None:
        return gather_results(stats.contiguous(), group, out, async_op=async_op)
    if async_op:
        raise ValueError("the ragged statistics gather has no asynchronous form")
    return gather_ragged(stats.contiguous(), total, group)


def _as_host_tensor(a):
    if isinstance(a, torch.Tensor):
        if a.device.type != "cpu":
            raise ValueError("run_streamed takes HOST arrays (it owns the copies to the device)")
        return a.contiguous()
    return torch.from_numpy(a).contiguous()       # shares numpy's memory


def run_streamed(forward, point_cloud, one_hot, chunk, device, with_logits=False, detections=False,
                 rot_angle=None):
    """A scene set larger than one batch: walk this rank's frustums in chunks of `chunk`
    with the next chunk's host -> device copy overlapped with the current chunk's compute.

    point_cloud (B,N,C) and one_hot (B,3): float32 HOST arrays or tensors; forward(pc, oh)
    -> end_points on `device`; returns the (B, RECORD [+2N]) result records on `device`, or
    with detections=True the (B, DETECTION) float64 detection records (rot_angle: (B,)
    float32 host array of the frustums' rotation angles, None = unrotated).

    Two device slots and two pinned staging slots; per chunk: the host copies its rows into
    the pinned slot (skipped when the source is pinned already), a COPY stream moves them
    to the device slot, the compute stream waits for that copy only, and the copy stream
    waits for the slot's previous compute before it overwrites the slot.  Compute never
    waits for the host except through a full staging pipeline."""
    pc_h, oh_h = _as_host_tensor(point_cloud), _as_host_tensor(one_hot)
    if pc_h.dtype != torch.float32 or oh_h.dtype != torch.float32 or pc_h.dim() != 3 or oh_h.dim() != 2 \
            or oh_h.shape[0] != pc_h.shape[0]:
        raise ValueError("point_cloud must be (B,N,C) float32 and one_hot (B,K) float32 with the same B")
    if chunk < 1:
        raise ValueError("chunk must be positive")
    total, n, c = pc_h.shape
    device = torch.device(device)
    if detections and with_logits:
        raise ValueError("detection records do not carry logits")
    if detections:
        records = torch.empty((total, DETECTION), dtype=torch.float64, device=device)
        rot_d = None if rot_angle is None else _as_host_tensor(rot_angle).to(device, torch.float32).reshape(total)
        pack = lambda ep, lo, hi: pack_detections(ep, None if rot_d is None else rot_d[lo:hi])      # noqa: E731
    else:
        records = torch.empty((total, RECORD + (2 * n if with_logits else 0)), dtype=torch.float32, device=device)
        pack = lambda ep, lo, hi: pack_results(ep, with_logits)                                       # noqa: E731
    if total == 0:
        return records
    if device.type != "cuda":
        for lo in range(0, total, chunk):
            hi = min(total, lo + chunk)
            records[lo:hi] = pack(forward(pc_h[lo:hi].to(device), oh_h[lo:hi].to(device)), lo, hi)
        return records

    rows = min(chunk, total)
    pinned_src = pc_h.is_pinned() and oh_h.is_pinned()
    with torch.cuda.device(device):
        main = torch.cuda.current_stream()
        copy = torch.cuda.Stream()
        dev_pc = [torch.empty((rows, n, c), dtype=torch.float32, device=device) for _ in range(2)]
        dev_oh = [torch.empty((rows, oh_h.shape[1]), dtype=torch.float32, device=device) for _ in range(2)]
        stage = None if pinned_src else [(torch.empty((rows, n, c), dtype=torch.float32).pin_memory(),
                                          torch.empty((rows, oh_h.shape[1]), dtype=torch.float32).pin_memory())
                                         for _ in range(2)]
        copied = [torch.cuda.Event() for _ in range(2)]      # slot's H2D done (copy stream)
        consumed = [torch.cuda.Event() for _ in range(2)]    # slot's compute done (compute stream)
        used = [False, False]

        def enqueue_copy(i):
            lo = i * chunk
            hi = min(total, lo + chunk)
            slot, k = i % 2, hi - lo
            if used[slot]:
                copy.wait_event(consumed[slot])      # device slot: its previous chunk has been computed
            if stage is not None:
                if used[slot]:
                    copied[slot].synchronize()       # staging slot: its previous chunk has left the host
                stage[slot][0][:k].copy_(pc_h[lo:hi])
                stage[slot][1][:k].copy_(oh_h[lo:hi])
                src_pc, src_oh = stage[slot][0][:k], stage[slot][1][:k]
            else:
                src_pc, src_oh = pc_h[lo:hi], oh_h[lo:hi]
            with torch.cuda.stream(copy):
                dev_pc[slot][:k].copy_(src_pc, non_blocking=True)
                dev_oh[slot][:k].copy_(src_oh, non_blocking=True)
                copied[slot].record(copy)
            used[slot] = True

        nchunks = (total + chunk - 1) // chunk
        enqueue_copy(0)
        for i in range(nchunks):
            if i + 1 < nchunks:
                enqueue_copy(i + 1)
            lo = i * chunk
            hi = min(total, lo + chunk)
            slot, k = i % 2, hi - lo
            main.wait_event(copied[slot])
            records[lo:hi] = pack(forward(dev_pc[slot][:k], dev_oh[slot][:k]), lo, hi)
            consumed[slot].record(main)
        # the slots must outlive the queued work that reads them
        main.synchronize()
    return records


def run_sharded_streamed(forward, point_cloud, one_hot, chunk, device, with_logits=False, group=None,
                         detections=False, rot_angle=None):
    """The whole scene set on every rank's host (or at least this rank's block of it):
    rank r streams frustums shard_range(B, r, W) through `forward` and the records of all
    ranks are gathered once at the end -> (B, RECORD [+2N]) on every rank, or with
    detections=True the (B, DETECTION) detection records."""
    total = point_cloud.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(total, rank, world)
    rec = run_streamed(forward, point_cloud[lo:hi], one_hot[lo:hi], chunk, device, with_logits, detections,
                       None if rot_angle is None else rot_angle[lo:hi])
    return gather_ragged(rec, total, group)
