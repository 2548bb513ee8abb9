// Builds the reference's own stand-alone CPU loops into oracle/_ref/libfpn_ref.so.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes
// the reference sources from where they lie under /root/reference (the include
// directories are given on the compiler command line by oracle/Makefile) and
// re-exports five of their functions with C linkage.  It can only be compiled
// where /root/reference is mounted; the resulting .so is used by the tests to
// pin oracle/fpn_oracle.c and, optionally, as the "reference" CPU baseline.
//
//   query_ball_point.cpp : query_ball_point_cpu (:19-47), group_point_cpu (:52-66)
//   interpolate.cpp      : interpolate_cpu (:84-104); threenn_cpu (:21-64), whose
//                          distance at :34 is x2*x2+y2*y2+z2*z2 (it ignores xyz1:
//                          every query sits at the origin).  Exported under a name
//                          that says so; for a known set translated by the caller
//                          (x2' = x2 - x1 in fp32) its d is bit for bit the
//                          (x2-x1)*(x2-x1)+... of tf_interpolate.cpp:73, and the
//                          double widening, 1e40 start and strict-< cascade (:27-57)
//                          are the text of tf_interpolate.cpp:66-97.
//   selection_sort.cpp   : selection_sort_cpu (:20-63) [debug printf silenced]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

namespace ref_qbp {
#include "query_ball_point.cpp"
}
namespace ref_interp {
#include "interpolate.cpp"
}
#define printf(...) ((void)0)
namespace ref_ssort {
#include "selection_sort.cpp"
}
#undef printf

extern "C" {
__attribute__((visibility("default")))
void ref_query_ball_point_cpu(int b, int n, int m, float radius, int nsample,
                              const float *xyz1, const float *xyz2, int *idx) {
    ref_qbp::query_ball_point_cpu(b, n, m, radius, nsample, xyz1, xyz2, idx);
}
__attribute__((visibility("default")))
void ref_group_point_cpu(int b, int n, int c, int m, int nsample, const float *points,
                         const int *idx, float *out) {
    ref_qbp::group_point_cpu(b, n, c, m, nsample, points, idx, out);
}
__attribute__((visibility("default")))
void ref_interpolate_cpu(int b, int m, int c, int n, const float *points, const int *idx,
                         const float *weight, float *out) {
    ref_interp::interpolate_cpu(b, m, c, n, points, idx, weight, out);
}
__attribute__((visibility("default")))
void ref_threenn_origin_cpu(int b, int n, int m, const float *xyz1, const float *xyz2,
                            float *dist, int *idx) {
    ref_interp::threenn_cpu(b, n, m, xyz1, xyz2, dist, idx);
}
__attribute__((visibility("default")))
void ref_selection_sort_cpu(int b, int n, int m, int k, const float *dist, int *idx,
                            float *val) {
    ref_ssort::selection_sort_cpu(b, n, m, k, dist, idx, val);
}
}
