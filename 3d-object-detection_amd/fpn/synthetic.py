"""Synthetic frustums and weights for tests and benchmarks (there is no dataset and
no checkpoint in the reference tree).

Frustum i: depth z ~ U(2,60) m (the clip range of dataset/data_extraction.py:29-30),
x,y ~ N(0,1)*(0.1*z) so the cloud widens with depth like a camera frustum, 30 % of
the points inside a class-sized box (mean sizes of models/model_util.py:38-45) so
the segmentation mask is non-trivial, 10 % exact duplicates (the reference
resamples with replacement, scripts/provider.py:186-187), intensity ~ U(0,1);
one_hot[i] = eye(3)[i % 3].
"""
import numpy as np

_CLASS_SIZE = np.array([[3.88311640418, 1.62856739989, 1.52563191462],   # Car
                        [0.84422524, 0.66068622, 1.76255119],           # Pedestrian
                        [1.76282397, 0.59706367, 1.73698127]])          # Cyclist


def make_frustums(batch, num_point, channels=4, seed=0, first_index=0):
    """-> point_cloud (batch,num_point,channels) float32, one_hot (batch,3) float32.
    first_index: global index of the first frustum (its class is index % 3)."""
    rng = np.random.RandomState(seed)
    z = rng.uniform(2.0, 60.0, size=(batch, num_point)).astype(np.float32)
    xy = rng.standard_normal((batch, num_point, 2)).astype(np.float32) * (0.1 * z)[..., None]
    pc = np.concatenate([xy, z[..., None]], axis=2)
    cls = (np.arange(batch) + first_index) % 3
    n_obj = int(0.3 * num_point)
    centre_z = rng.uniform(5.0, 50.0, size=(batch,)).astype(np.float32)
    centre_xy = rng.standard_normal((batch, 2)).astype(np.float32) * (0.05 * centre_z)[:, None]
    size = _CLASS_SIZE[cls].astype(np.float32)                      # (l, w, h)
    u = rng.uniform(-0.5, 0.5, size=(batch, n_obj, 3)).astype(np.float32)
    obj = np.stack([centre_xy[:, 0:1] + u[..., 0] * size[:, 0:1],
                    centre_xy[:, 1:2] + u[..., 1] * size[:, 2:3],
                    centre_z[:, None] + u[..., 2] * size[:, 1:2]], axis=2)
    where = np.stack([rng.permutation(num_point)[:n_obj] for _ in range(batch)])
    np.put_along_axis(pc, where[..., None].repeat(3, axis=2), obj, axis=1)
    n_dup = int(0.1 * num_point)
    for i in range(batch):
        dst = rng.permutation(num_point)[:n_dup]
        src = rng.randint(0, num_point, size=n_dup)
        pc[i, dst] = pc[i, src]
    if channels > 3:
        extra = rng.uniform(0.0, 1.0, size=(batch, num_point, channels - 3)).astype(np.float32)
        pc = np.concatenate([pc, extra], axis=2)
    one_hot = np.eye(3, dtype=np.float32)[cls]
    return np.ascontiguousarray(pc, dtype=np.float32), one_hot


def make_frustums_range(lo, hi, num_point, channels=4, seed=0):
    """Frustums lo..hi-1 of ONE global seeded set whose frustum i is drawn from its own
    stream (seed, i): any rank can make exactly its shard of the set without making the
    rest, and shards made by different ranks concatenate to the set a single process
    makes.  Same distribution as make_frustums; one_hot[i] = eye(3)[i % 3] of the GLOBAL
    index.  -> point_cloud (hi-lo,num_point,channels), one_hot (hi-lo,3)."""
    pcs = np.empty((max(hi - lo, 0), num_point, channels), np.float32)
    one_hot = np.empty((max(hi - lo, 0), 3), np.float32)
    for i in range(lo, hi):
        pc, oh = make_frustums(1, num_point, channels, seed=[int(seed), i], first_index=i)
        pcs[i - lo], one_hot[i - lo] = pc[0], oh[0]
    return pcs, one_hot


def unit_cube(batch, n, seed=100):
    """U[0,1)^3 clouds, the distribution of the reference's op demos
    (models/tf_ops/grouping/tf_grouping.py:80-83)."""
    rng = np.random.RandomState(seed)
    return rng.random_sample((batch, n, 3)).astype(np.float32)
