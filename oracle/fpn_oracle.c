/*
 * CPU oracle for the Frustum-PointNet point-set operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library, the
 * fpn package) may link, load or call this file.  It exists so that tests, the
 * smoke check and the benchmark's cpu_baseline leg have a scalar, single-thread
 * statement of what each operator must return.
 *
 * Every function restates the algorithm of one reference routine; the citation
 * above each function names the reference file and lines it follows.  The
 * restatement is pinned two ways (see tests/test_oracle_pinning.py):
 *   - against the reference's own CPU loops compiled from /root/reference into
 *     oracle/_ref (ball query, group, interpolate, selection sort, and three_nn
 *     through the stand-alone threenn_cpu of interpolate.cpp:21-64 run on origin
 *     queries and on query-translated known sets);
 *   - against the one known-answer case the reference carries
 *     (selection_sort.cpp:65-94).
 * FPS and gather_point have no CPU code in the reference (GPU-only kernels):
 * their restatement follows the CUDA text line by line and is otherwise
 * "parity unpinned" by reference tests.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared  (no -ffast-math, no FMA
 * contraction: the distance expression is three rounded products summed left
 * to right, which is what g++ produces for the reference's CPU twins on x86-64).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* squared distance exactly as written in tf_sampling_g.cu:142,
 * tf_grouping_g.cu:24, tf_interpolate.cpp:73: (x2-x1)*(x2-x1)+... */
static inline float sqdist3(const float *p2, const float *p1)
{
    float dx = p2[0] - p1[0];
    float dy = p2[1] - p1[1];
    float dz = p2[2] - p1[2];
    float sx = dx * dx;
    float sy = dy * dy;
    float sz = dz * dz;
    float s = sx + sy;
    s = s + sz;
    return s;
}

/*
 * Farthest point sampling.  Follows farthestpointsamplingKernel,
 * models/tf_ops/sampling/tf_sampling_g.cu:105-170:
 *   :114-116  first picked index is 0
 *   :118      running min-distance starts at 1e38
 *   :142-145  d2 = min(d, temp[k]) is stored back
 *   :146-165  arg-max of d2
 * lane_emulation == 0: ties go to the lowest point index (the rule this build
 *   specifies).
 * lane_emulation == 1: ties resolve as the reference's 512-thread block does:
 *   each thread t scans k = t, t+512, ... keeping the first strict maximum
 *   (:146), then the shared-memory tree keeps the lower slot on ties (:158),
 *   i.e. lowest (k mod 512), then lowest k.  Identical to mode 0 when n <= 512
 *   or when no two candidates tie exactly.
 * tmp must hold n floats.
 */
ORACLE_API void oracle_farthest_point_sample(int b, int n, int m, const float *inp,
                                             float *tmp, int *out, int lane_emulation)
{
    if (m <= 0) return; /* :106 */
    for (int i = 0; i < b; ++i) {
        const float *pts = inp + (size_t)i * n * 3;
        int *o = out + (size_t)i * m;
        int old = 0;
        o[0] = old;
        for (int k = 0; k < n; ++k) tmp[k] = 1e38f;
        for (int j = 1; j < m; ++j) {
            const float *p1 = pts + (size_t)old * 3;
            int besti = 0;
            float best = -1.0f;
            if (!lane_emulation) {
                for (int k = 0; k < n; ++k) {
                    float d = sqdist3(pts + (size_t)k * 3, p1);
                    float d2 = d < tmp[k] ? d : tmp[k];
                    tmp[k] = d2;
                    if (d2 > best) { best = d2; besti = k; }
                }
            } else {
                /* per-lane pass, then the lanes compared in slot order; the
                 * pairwise tree with "lower slot survives ties" returns the
                 * lowest slot holding the maximum, which a linear scan with a
                 * strict compare reproduces. */
                const int lanes = 512;
                for (int t = 0; t < lanes; ++t) {
                    int lbesti = 0;
                    float lbest = -1.0f;
                    for (int k = t; k < n; k += lanes) {
                        float d = sqdist3(pts + (size_t)k * 3, p1);
                        float d2 = d < tmp[k] ? d : tmp[k];
                        tmp[k] = d2;
                        if (d2 > lbest) { lbest = d2; lbesti = k; }
                    }
                    if (t == 0 || lbest > best) { best = lbest; besti = lbesti; }
                }
            }
            old = besti;
            o[j] = old;
        }
    }
}

/* gatherpointKernel, tf_sampling_g.cu:172-181: out[i,j,:] = inp[i,idx[i,j],:] */
ORACLE_API void oracle_gather_point(int b, int n, int m, const float *inp,
                                    const int *idx, float *out)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            int a = idx[(size_t)i * m + j];
            const float *src = inp + ((size_t)i * n + a) * 3;
            float *dst = out + ((size_t)i * m + j) * 3;
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
        }
}

/*
 * Ball query.  Follows query_ball_point_gpu, tf_grouping_g.cu:3-36, and its CPU
 * twin query_ball_point_cpu, query_ball_point.cpp:19-47: scan the dataset in
 * index order, accept while max(sqrtf(d2),1e-20f) < radius, the first hit fills
 * the whole row, stop at nsample hits; pts_cnt = hits (tf_grouping_g.cu:34).
 * The reference leaves a row untouched when no point is in the ball (the output
 * buffer is not zeroed); this build defines that case: the row is all zeros and
 * pts_cnt is 0.
 */
ORACLE_API void oracle_query_ball_point(int b, int n, int m, float radius, int nsample,
                                        const float *xyz1, const float *xyz2,
                                        int *idx, int *pts_cnt)
{
    for (int i = 0; i < b; ++i) {
        const float *data = xyz1 + (size_t)i * n * 3;
        const float *query = xyz2 + (size_t)i * m * 3;
        for (int j = 0; j < m; ++j) {
            int *row = idx + ((size_t)i * m + j) * nsample;
            int cnt = 0;
            for (int l = 0; l < nsample; ++l) row[l] = 0;
            for (int k = 0; k < n && cnt < nsample; ++k) {
                float s = sqdist3(query + (size_t)j * 3, data + (size_t)k * 3);
                float d = sqrtf(s);
                if (d < 1e-20f) d = 1e-20f; /* std::max(d,1e-20f) */
                if (d < radius) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) row[l] = k;
                    row[cnt++] = k;
                }
            }
            if (pts_cnt) pts_cnt[(size_t)i * m + j] = cnt;
        }
    }
}

/* group_point_gpu, tf_grouping_g.cu:40-57 / group_point_cpu,
 * query_ball_point.cpp:52-66: out[i,j,k,:] = points[i,idx[i,j,k],:] */
ORACLE_API void oracle_group_point(int b, int n, int c, int m, int nsample,
                                   const float *points, const int *idx, float *out)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                int ii = idx[((size_t)i * m + j) * nsample + k];
                memcpy(out + (((size_t)i * m + j) * nsample + k) * c,
                       points + ((size_t)i * n + ii) * c, sizeof(float) * (size_t)c);
            }
}

/*
 * Three nearest neighbours on squared distance.  Follows threenn_cpu,
 * models/tf_ops/3d_interpolation/tf_interpolate.cpp:60-103 (NOT
 * interpolate.cpp:21-64, whose distance at :34 ignores xyz1).  The distance is
 * the float expression widened to double (:73); the running bests start at the
 * double 1e40 (:66) and are replaced on strict "<" (:74-89); unfilled slots
 * come out as (float)1e40 = +inf with index 0.
 */
ORACLE_API void oracle_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2,
                                float *dist, int *idx)
{
    for (int i = 0; i < b; ++i) {
        const float *q = xyz1 + (size_t)i * n * 3;
        const float *ref = xyz2 + (size_t)i * m * 3;
        for (int j = 0; j < n; ++j) {
            double b1 = 1e40, b2 = 1e40, b3 = 1e40;
            int i1 = 0, i2 = 0, i3 = 0;
            for (int k = 0; k < m; ++k) {
                double d = (double)sqdist3(ref + (size_t)k * 3, q + (size_t)j * 3);
                if (d < b1) {
                    b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
                } else if (d < b2) {
                    b3 = b2; i3 = i2; b2 = d; i2 = k;
                } else if (d < b3) {
                    b3 = d; i3 = k;
                }
            }
            float *dd = dist + ((size_t)i * n + j) * 3;
            int *ii = idx + ((size_t)i * n + j) * 3;
            dd[0] = (float)b1; dd[1] = (float)b2; dd[2] = (float)b3;
            ii[0] = i1; ii[1] = i2; ii[2] = i3;
        }
    }
}

/* threeinterpolate_cpu, tf_interpolate.cpp:107-127 (= interpolate_cpu,
 * interpolate.cpp:84-104): out = p[i1]*w1 + p[i2]*w2 + p[i3]*w3, left to right */
ORACLE_API void oracle_three_interpolate(int b, int m, int c, int n, const float *points,
                                         const int *idx, const float *weight, float *out)
{
    for (int i = 0; i < b; ++i) {
        const float *p = points + (size_t)i * m * c;
        for (int j = 0; j < n; ++j) {
            const int *ii = idx + ((size_t)i * n + j) * 3;
            const float *w = weight + ((size_t)i * n + j) * 3;
            float *o = out + ((size_t)i * n + j) * c;
            for (int l = 0; l < c; ++l) {
                float t1 = p[(size_t)ii[0] * c + l] * w[0];
                float t2 = p[(size_t)ii[1] * c + l] * w[1];
                float t3 = p[(size_t)ii[2] * c + l] * w[2];
                float s = t1 + t2;
                o[l] = s + t3;
            }
        }
    }
}

/*
 * Inverse-distance weights used by the feature-propagation module,
 * models/pointnet_util.py:211-214: dist = max(dist,1e-10); w = (1/dist)/sum(1/dist).
 * The three reciprocals are summed left to right.
 */
ORACLE_API void oracle_fp_weights(size_t rows, const float *dist, float *weight)
{
    for (size_t r = 0; r < rows; ++r) {
        float inv[3];
        for (int t = 0; t < 3; ++t) {
            float d = dist[r * 3 + t];
            d = d > 1e-10f ? d : 1e-10f;
            inv[t] = 1.0f / d;
        }
        float norm = inv[0] + inv[1];
        norm = norm + inv[2];
        for (int t = 0; t < 3; ++t) weight[r * 3 + t] = inv[t] / norm;
    }
}

/*
 * Partial selection sort, selection_sort_gpu, tf_grouping_g.cu:83-123 /
 * selection_sort_cpu, selection_sort.cpp:20-63: copy dist (b,m,n) to out, fill
 * outi with 0..n-1, then for the first k positions of each row swap in the
 * minimum of the tail (strict "<": the lowest position wins ties).
 */
ORACLE_API void oracle_selection_sort(int b, int n, int m, int k, const float *dist,
                                      int *outi, float *out)
{
    for (size_t r = 0; r < (size_t)b * m; ++r) {
        float *v = out + r * n;
        int *ix = outi + r * n;
        for (int s = 0; s < n; ++s) { v[s] = dist[r * n + s]; ix[s] = s; }
        for (int s = 0; s < k && s < n; ++s) {
            int mn = s;
            for (int t = s + 1; t < n; ++t)
                if (v[t] < v[mn]) mn = t;
            if (mn != s) {
                float tv = v[mn]; v[mn] = v[s]; v[s] = tv;
                int ti = ix[mn]; ix[mn] = ix[s]; ix[s] = ti;
            }
        }
    }
}
