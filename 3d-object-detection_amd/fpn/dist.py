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


def unpack_results(records, num_point=None):
    out = {"box_output": records[:, :59], "stage1_center": records[:, 59:62],
           "mask_xyz_mean": records[:, 62:65]}
    if records.shape[1] > RECORD:
        out["mask_logits"] = records[:, RECORD:].reshape(records.shape[0], num_point, 2)
    return out


def gather_results(records, group=None, out=None):
    """All-gather equal-sized per-rank record blocks -> (world*B_local, width) on every
    rank, rank-major (so with shard_range the global frustum order is preserved)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return records
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * records.shape[0], records.shape[1]), dtype=records.dtype,
                          device=records.device)
    dist.all_gather_into_tensor(out, records, group=group)
    return out


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
