// 3-D box IoU of the detection evaluation (helpers/mAP_evaluation.py:103-182) in
// float64, one thread per box pair / per prediction.
//
// A box is 10 doubles: translation (x,y,z), size (width,length,height), rotation
// quaternion (w,x,y,z).  The reference builds the ground rectangle from two entries
// of the quaternion's rotation matrix (:117-132), intersects the rectangles with
// shapely and multiplies by the overlap of the height intervals (:156-174).  Here the
// intersection of the two convex quadrilaterals is a Sutherland-Hodgman clip followed
// by the shoelace formula.  This translation unit is compiled with -ffp-contract=off:
// every operation is a separately rounded IEEE double operation, in the order written,
// so that a scalar CPU restatement reproduces the results bit for bit.
#include <cmath>

#include "fpn_common.h"

namespace fpn {
namespace {

struct Ground {
    double x[4], y[4];   // corners 0..3 in the reference's order (:122-132)
    double min_z, max_z, volume;
};

__device__ void ground_of(const double *b, Ground &g)
{
    const double cx = b[0], cy = b[1], cz = b[2];
    const double width = b[3], length = b[4], height = b[5];
    double qw = b[6], qx = b[7], qy = b[8], qz = b[9];
    // pyquaternion normalises before it forms the rotation matrix: skipped when the
    // squared norm is within 1e-14 of one, and when the norm is zero
    const double ss = ((qw * qw + qx * qx) + qy * qy) + qz * qz;
    if (!(fabs(1.0 - ss) < 1e-14)) {
        const double n = sqrt(ss);
        if (n > 0) {
            qw = qw / n; qx = qx / n; qy = qy / n; qz = qz / n;
        }
    }
    // rotation_matrix[0,0] and [0,1] (:119-120) as rows of the 4x4 product
    // Q(q) . Qbar(q)^T:  [x, w, -z, y] . [x, w, -z, y]  and  [x, w, -z, y] . [y, -z, w, x]
    const double c = ((qx * qx + qw * qw) + (-qz) * qz) + qy * (-qy);
    const double s = ((qx * qy + qw * (-qz)) + (-qz) * qw) + qy * qx;
    const double hl = length / 2, hw = width / 2;
    g.x[0] = cx + hl * c + hw * s;  g.y[0] = cy + hl * s - hw * c;
    g.x[1] = cx + hl * c - hw * s;  g.y[1] = cy + hl * s + hw * c;
    g.x[2] = cx - hl * c - hw * s;  g.y[2] = cy - hl * s + hw * c;
    g.x[3] = cx - hl * c + hw * s;  g.y[3] = cy - hl * s - hw * c;
    g.min_z = cz - height / 2;
    g.max_z = cz + height / 2;
    g.volume = (width * length) * height;   // np.prod(size), :79
}

// twice the signed area of a polygon (shoelace)
__device__ double shoelace(const double *x, const double *y, int n)
{
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = i + 1 == n ? 0 : i + 1;
        a = a + (x[i] * y[j] - x[j] * y[i]);
    }
    return a;
}

// area of (quad p) intersect (quad q); both convex
__device__ double quad_intersection_area(const Ground &p, const Ground &q)
{
    double px[8], py[8], rx[8], ry[8];
    int n = 4;
    for (int i = 0; i < 4; ++i) { px[i] = p.x[i]; py[i] = p.y[i]; }
    // the inside of q is to the left of its edges when q is counter-clockwise
    const double orient = shoelace(q.x, q.y, 4) >= 0 ? 1.0 : -1.0;
    for (int e = 0; e < 4 && n > 0; ++e) {
        const int e1 = (e + 1) & 3;
        const double ax = q.x[e], ay = q.y[e];
        const double ex = q.x[e1] - ax, ey = q.y[e1] - ay;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = i + 1 == n ? 0 : i + 1;
            const double dc = orient * (ex * (py[i] - ay) - ey * (px[i] - ax));
            const double dn = orient * (ex * (py[j] - ay) - ey * (px[j] - ax));
            const bool in_c = dc >= 0, in_n = dn >= 0;
            if (in_c && m < 8) { rx[m] = px[i]; ry[m] = py[i]; ++m; }
            if (in_c != in_n && m < 8) {
                const double t = dc / (dc - dn);
                rx[m] = px[i] + t * (px[j] - px[i]);
                ry[m] = py[i] + t * (py[j] - py[i]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) { px[i] = rx[i]; py[i] = ry[i]; }
    }
    if (n < 3) return 0.0;
    return fabs(shoelace(px, py, n)) / 2;
}

__device__ double box_iou(const double *a, const double *b)
{
    Ground ga, gb;
    ground_of(a, ga);
    ground_of(b, gb);
    const double lo = fmax(gb.min_z, ga.min_z), hi = fmin(gb.max_z, ga.max_z);   // :157-160
    const double dz = hi - lo;
    const double height = dz > 0 ? dz : 0.0;
    const double inter = height * quad_intersection_area(ga, gb);                 // :174
    const double uni = (ga.volume + gb.volume) - inter;                           // :178
    const double iou = inter / uni;
    return iou < 0 ? 0.0 : (iou > 1 ? 1.0 : iou);                                 // np.clip, :180
}

__global__ void box_iou_pairs_kernel(long n, const double *__restrict__ a, long stride_a,
                                     const double *__restrict__ b, long stride_b, double *__restrict__ iou)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) iou[i] = box_iou(a + i * stride_a, b + i * stride_b);
}

// the two factors of the intersection volume, separately: area of the overlap of the
// ground rectangles (:162-167) and overlap of the height intervals (:156-160)
__global__ void box_intersection_kernel(long n, const double *__restrict__ a, long stride_a,
                                        const double *__restrict__ b, long stride_b, double *__restrict__ area,
                                        double *__restrict__ height)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ground ga, gb;
    ground_of(a + i * stride_a, ga);
    ground_of(b + i * stride_b, gb);
    const double dz = fmin(gb.max_z, ga.max_z) - fmax(gb.min_z, ga.min_z);
    if (height) height[i] = dz > 0 ? dz : 0.0;
    if (area) area[i] = quad_intersection_area(ga, gb);
}

__global__ void box_best_overlap_kernel(int npred, const double *__restrict__ pred, const double *__restrict__ gt,
                                        const int *__restrict__ gt_begin, const int *__restrict__ gt_count,
                                        double *__restrict__ best_iou, int *__restrict__ best_j)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npred) return;
    double best = -INFINITY;   // :282-283
    int jbest = -1;
    const int g0 = gt_begin[i], cnt = gt_count[i];
    for (int j = 0; j < cnt; ++j) {
        const double v = box_iou(pred + (long)i * 10, gt + (long)(g0 + j) * 10);
        // np.max / np.argmax (:295-297): first maximum wins; a NaN propagates like np.max
        if (jbest < 0 || v > best || (v != v && best == best)) {
            best = v;
            jbest = j;
        }
    }
    best_iou[i] = best;
    best_j[i] = jbest;
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" {

int fpn_box3d_iou_f64(long n, const double *a, long stride_a, const double *b, long stride_b, double *iou,
                      void *stream)
{
    FPN_REQUIRE(n >= 0, "box3d_iou: negative count %ld", n);
    if (n == 0) return FPN_OK;
    FPN_REQUIRE(a && b && iou, "box3d_iou: null pointer");
    FPN_REQUIRE((stride_a == 0 || stride_a >= 10) && (stride_b == 0 || stride_b >= 10),
                "box3d_iou: box strides must be 0 (broadcast) or >= 10 doubles (got %ld, %ld)", stride_a, stride_b);
    hipLaunchKernelGGL(box_iou_pairs_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, as_stream(stream), n, a,
                       stride_a, b, stride_b, iou);
    return check_launch("fpn_box3d_iou_f64");
}

int fpn_box3d_intersection_f64(long n, const double *a, long stride_a, const double *b, long stride_b,
                               double *area, double *height, void *stream)
{
    FPN_REQUIRE(n >= 0, "box3d_intersection: negative count %ld", n);
    if (n == 0) return FPN_OK;
    FPN_REQUIRE(a && b && (area || height), "box3d_intersection: null pointer");
    FPN_REQUIRE((stride_a == 0 || stride_a >= 10) && (stride_b == 0 || stride_b >= 10),
                "box3d_intersection: box strides must be 0 (broadcast) or >= 10 doubles (got %ld, %ld)", stride_a, stride_b);
    hipLaunchKernelGGL(box_intersection_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, as_stream(stream), n, a,
                       stride_a, b, stride_b, area, height);
    return check_launch("fpn_box3d_intersection_f64");
}

int fpn_box3d_best_overlap_f64(int npred, const double *pred, const double *gt, const int *gt_begin,
                               const int *gt_count, double *best_iou, int *best_j, void *stream)
{
    FPN_REQUIRE(npred >= 0, "box3d_best_overlap: negative count %d", npred);
    if (npred == 0) return FPN_OK;
    FPN_REQUIRE(pred && gt_begin && gt_count && best_iou && best_j, "box3d_best_overlap: null pointer");
    hipLaunchKernelGGL(box_best_overlap_kernel, dim3((unsigned)((npred + 63) / 64)), dim3(64), 0, as_stream(stream),
                       npred, pred, gt, gt_begin, gt_count, best_iou, best_j);
    return check_launch("fpn_box3d_best_overlap_f64");
}

}  // extern "C"
