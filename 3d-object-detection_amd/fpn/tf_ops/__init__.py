"""Drop-in counterparts of the reference's models/tf_ops/* Python wrappers."""
from . import tf_grouping, tf_interpolate, tf_sampling  # noqa: F401
