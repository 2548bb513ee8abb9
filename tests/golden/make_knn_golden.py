"""Generates tests/golden/knn_golden.npz: knn_point on the inputs the reference itself
chose, and on a tie-heavy set, with the outputs of the reference's own selection sort.

  * "ref_*": the numpy half of models/tf_ops/grouping/test_knn.py:5-18 executed as written -
    np.random.seed(0), a_val = random((2,5,3)), b_val = random((2,2,3)), the brute-force
    squared distances np.sum((t2[i,:]-t1[j,:])**2) in float64 - plus the stable three
    smallest per query (what the script's tf.nn.top_k(-dist, 3) prints; TensorFlow is not
    installed, numpy's stable argsort stands in for it - the values are distinct).
  * "*_idx_ref" / "*_val_ref": the float32 distance matrix of knn_point
    (tf_grouping.py:65-67; oracle.pointset.knn_distances) through the REFERENCE's
    selection_sort_cpu (selection_sort.cpp:20-63, compiled into oracle/_ref), first k columns.
Run in the build container only:   make -C oracle all ref && python tests/golden/make_knn_golden.py
The file holds data only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pointset as O  # noqa: E402


def main():
    assert O.have_ref(), "build oracle/_ref first (needs /root/reference)"
    g = {}
    np.random.seed(0)                                   # test_knn.py:5
    a_val = np.random.random((2, 5, 3))                 # :8
    b_val = np.random.random((2, 2, 3))                 # :9
    d64 = np.zeros((2, 2, 5))
    for b in range(2):                                  # :10-18
        t1, t2 = a_val[b], b_val[b]
        for i in range(2):
            for j in range(5):
                d64[b, i, j] = np.sum((t2[i, :] - t1[j, :]) ** 2)
    g["ref_a_val"], g["ref_b_val"], g["ref_dist_float64"] = a_val, b_val, d64
    g["ref_top3_float64"] = np.argsort(d64, -1, kind="stable")[..., :3].astype(np.int32)
    a32, b32 = a_val.astype(np.float32), b_val.astype(np.float32)
    oi, ov = O.ref_select_top_k(3, O.knn_distances(a32, b32))
    g["ref_idx_ref"], g["ref_val_ref"] = oi[..., :3].copy(), ov[..., :3].copy()
    # lattice coordinates: many exactly equal distances, k deep into the ties; 5 channels
    rng = np.random.RandomState(20240612)
    for tag, (b, n, m, c, k) in {"ties": (2, 200, 7, 3, 16), "wide": (1, 130, 5, 5, 9), "all": (1, 33, 3, 3, 33)}.items():
        x1 = rng.randint(0, 4, (b, n, c)).astype(np.float32)
        x2 = rng.randint(0, 4, (b, m, c)).astype(np.float32)
        oi, ov = O.ref_select_top_k(k, O.knn_distances(x1, x2))
        g[tag + "_xyz1"], g[tag + "_xyz2"], g[tag + "_k"] = x1, x2, np.int32(k)
        g[tag + "_idx_ref"], g[tag + "_val_ref"] = oi[..., :k].copy(), ov[..., :k].copy()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "knn_golden.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
