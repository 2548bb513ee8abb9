"""Monte-Carlo ensemble statistics of the Bayesian detector (fpn_mc_statistics).

The reference has no code for this step: its report (section 3) defines the
quantities and viz_lyft/lyft_util.py:39-41 consumes them as the per-detection
columns `prob, total_variance, entropy`."""
import torch

from . import _lib

STAT_WIDTH = 30
FIELDS = {"mean": (0, 7), "total_variance": (7, 8), "heading_entropy": (8, 9), "size_entropy": (9, 10),
          "heading_prob": (10, 22), "size_prob": (22, 30)}


def mc_statistics(box_output, stage1_center):
    """box_output (S,B,59), stage1_center (S,B,3) -> (B,30) float32 statistics."""
    _lib.require_device_tensor("box_output", box_output, torch.float32, 3, 59)
    _lib.require_device_tensor("stage1_center", stage1_center, torch.float32, 3, 3)
    s, b, _ = box_output.shape
    stats = torch.empty((b, STAT_WIDTH), dtype=torch.float32, device=box_output.device)
    with torch.cuda.device(box_output.device):
        _lib.check(_lib.load().fpn_mc_statistics(s, b, _lib.ptr(box_output), _lib.ptr(stage1_center),
                                                 _lib.ptr(stats), _lib.stream_of(box_output)))
    return stats


def unpack(stats):
    out = {}
    for k, (lo, hi) in FIELDS.items():
        out[k] = stats[:, lo] if hi - lo == 1 else stats[:, lo:hi]
    return out
