"""fpn: MI355X-native Frustum-PointNet point-set hot path.

Python surface (same names as the reference's modules):
  fpn.tf_ops.tf_sampling / tf_grouping / tf_interpolate   point-set operators
  fpn.tf_util                                             fused layer wrappers
  fpn.pointnet_util                                       PointNet++ SA / FP modules
  fpn.model_util, fpn.models.*                            network builders
All compute is done by libfpn_hip.so (hand-written gfx950 kernels) through the C
ABI of include/fpn.h; torch tensors are only the container.
"""
from . import _lib, variables  # noqa: F401

__all__ = ["_lib", "variables"]
