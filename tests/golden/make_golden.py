"""Generates tests/golden/pointset_golden.npz: small seeded inputs with the outputs of
the REFERENCE's own CPU loops (compiled from /root/reference into oracle/_ref by
oracle/Makefile) where the reference has CPU code (*_ref: ball query, group_point,
three_interpolate, selection sort, and three_nn through the stand-alone threenn_cpu of
interpolate.cpp run on query-translated known sets, see oracle/pointset.ref_three_nn),
and of the restatement in oracle/fpn_oracle.c where it has none (FPS, gather_point:
marked *_restated).  Run in the build container only:

    make -C oracle all ref && python tests/golden/make_golden.py

The file holds data only (inputs and expected outputs).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pointset as O  # noqa: E402


def clouds(rng, b, n, dup=0.1):
    x = rng.random_sample((b, n, 3)).astype(np.float32)
    nd = int(dup * n)
    for i in range(b):
        if nd:
            x[i, rng.permutation(n)[:nd]] = x[i, rng.randint(0, n, nd)]
    return x


def main():
    assert O.have_ref(), "build oracle/_ref first (needs /root/reference)"
    rng = np.random.RandomState(20240607)
    g = {}
    # ball query + group: reference CPU twins (query_ball_point.cpp:19-66)
    for tag, (b, n, m, r, ns, c) in {"a": (2, 64, 8, 0.3, 4, 3), "b": (2, 512, 128, 0.1, 32, 16),
                                     "c": (2, 1100, 8, 0.25, 32, 1), "d": (1, 128, 32, 1.6, 128, 3)}.items():
        x1 = clouds(rng, b, n)
        x2 = clouds(rng, b, m, dup=0.0)
        own = rng.random_sample((b, m)) < 0.5
        pick = rng.randint(0, n, (b, m))
        x2[own] = np.take_along_axis(x1, pick[..., None].repeat(3, 2), 1)[own]
        idx = O.ref_query_ball_point(r, ns, x1, x2)          # rows of empty balls stay -1
        pts = rng.standard_normal((b, n, c)).astype(np.float32)
        safe = np.where(idx < 0, 0, idx)
        g["qb_%s_xyz1" % tag], g["qb_%s_xyz2" % tag] = x1, x2
        g["qb_%s_args" % tag] = np.array([r, ns], np.float64)
        g["qb_%s_idx_ref" % tag] = idx
        g["gp_%s_points" % tag] = pts
        g["gp_%s_out_ref" % tag] = O.ref_group_point(pts, safe)
    # a query set with empty balls (reference leaves those rows untouched: -1 here)
    x1 = clouds(rng, 1, 100)
    x2 = clouds(rng, 1, 6, dup=0.0)
    x2[0, ::2] += 10.0
    g["qb_empty_xyz1"], g["qb_empty_xyz2"] = x1, x2
    g["qb_empty_args"] = np.array([0.4, 8], np.float64)
    g["qb_empty_idx_ref"] = O.ref_query_ball_point(0.4, 8, x1, x2)
    # three_interpolate: reference interpolate_cpu (interpolate.cpp:84-104)
    for tag, (b, n, m, c) in {"a": (2, 128, 32, 16), "b": (1, 300, 5, 3), "c": (2, 64, 2, 7)}.items():
        x1, x2 = clouds(rng, b, n), clouds(rng, b, m)
        p2 = rng.standard_normal((b, m, c)).astype(np.float32)
        d, i = O.ref_three_nn(x1, x2)
        w = O.fp_weights(d)
        g["ti_%s_xyz1" % tag], g["ti_%s_xyz2" % tag], g["ti_%s_points" % tag] = x1, x2, p2
        g["ti_%s_dist_ref" % tag], g["ti_%s_idx_ref" % tag], g["ti_%s_weight" % tag] = d, i, w
        g["ti_%s_out_ref" % tag] = O.ref_three_interpolate(p2, i, w)
    # three_nn: the reference's threenn_cpu (interpolate.cpp:21-64).  "origin*": called as
    # it is, every query at the origin (its distance ignores xyz1); the others: through
    # ref_three_nn (known set translated by each query in fp32).  m < 3, duplicates of
    # known points, lattice coordinates (exact distance ties) and metre-scale frustums.
    for tag, (b, n, m, kind) in {"origin_a": (3, 4, 128, "origin"), "origin_b": (2, 2, 2, "origin"),
                                 "origin_c": (2, 1, 1, "origin"), "origin_ties": (2, 3, 60, "origin_lattice"),
                                 "m1": (2, 40, 1, "cube"), "m2": (2, 40, 2, "cube"), "m5": (2, 70, 5, "cube"),
                                 "m128": (2, 2048, 128, "cube"), "ties": (2, 300, 50, "lattice"),
                                 "subset": (1, 512, 64, "subset"), "metres": (2, 256, 32, "metres")}.items():
        if kind.startswith("origin"):
            x2 = clouds(rng, b, m) - np.float32(0.5) if kind == "origin" else \
                rng.randint(-2, 3, (b, m, 3)).astype(np.float32)
            x1 = np.zeros((b, n, 3), np.float32)
            d, i = O.ref_three_nn_origin(x2, n)
        else:
            if kind == "lattice":
                x1 = rng.randint(0, 4, (b, n, 3)).astype(np.float32)
                x2 = rng.randint(0, 4, (b, m, 3)).astype(np.float32)
            elif kind == "subset":                      # known points are a subset of the queries (FP modules)
                x1 = clouds(rng, b, n)
                x2 = x1[:, rng.permutation(n)[:m]].copy()
            elif kind == "metres":
                x1 = (clouds(rng, b, n) * np.float32([8, 4, 58]) + np.float32([-4, -2, 2])).astype(np.float32)
                x2 = x1[:, :m].copy() + (rng.random_sample((b, m, 3)) < 0.5).astype(np.float32) * np.float32(0.25)
            else:
                x1, x2 = clouds(rng, b, n), clouds(rng, b, m)
            d, i = O.ref_three_nn(x1, x2)
        g["tn_%s_xyz1" % tag], g["tn_%s_xyz2" % tag] = x1, x2
        g["tn_%s_dist_ref" % tag], g["tn_%s_idx_ref" % tag] = d, i
    # selection sort: the reference's known-answer input (selection_sort.cpp:65-94) and
    # a tie-heavy random case, both through the reference's selection_sort_cpu
    kat = (10 - np.arange(16, dtype=np.float32)).reshape(2, 2, 4)
    g["ss_kat_dist"] = kat
    g["ss_kat_idx_ref"], g["ss_kat_val_ref"] = O.ref_select_top_k(3, kat)
    ties = rng.randint(0, 20, (2, 3, 50)).astype(np.float32)
    g["ss_ties_dist"] = ties
    g["ss_ties_idx_ref"], g["ss_ties_val_ref"] = O.ref_select_top_k(7, ties)
    # FPS / gather: restatement only (the reference has GPU code only)
    for tag, (b, n, m) in {"a": (2, 64, 16), "b": (2, 600, 64), "c": (1, 2048, 128)}.items():
        x = clouds(rng, b, n)
        idx = O.farthest_point_sample(m, x)
        g["fps_%s_xyz" % tag] = x
        g["fps_%s_idx_restated" % tag] = idx
        g["fps_%s_idx_lane512_restated" % tag] = O.farthest_point_sample(m, x, lane_emulation=True)
        g["fps_%s_gather_restated" % tag] = O.gather_point(x, idx)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pointset_golden.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
