/*
 * fpn.h - C ABI of libfpn_hip.so, the gfx950 (MI355X) implementation of the
 * Frustum-PointNet point-set hot path.
 *
 * Conventions (all entry points):
 *   - plain C: ints, floats, raw device pointers, no C++ or torch types;
 *   - every pointer is DEVICE memory on the current HIP device, dense row-major
 *     (the layout TensorFlow gives the reference's tensors), float = IEEE fp32,
 *     indices = int32;
 *   - the caller owns every buffer (inputs, outputs, workspace): the library
 *     never allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*, NULL = the default stream);
 *   - return value: FPN_OK, or a negative FPN_E_* code; fpn_last_error_string()
 *     then describes the failure (thread-local);
 *   - inference only: no gradient entry points (the reference registers
 *     gradients for GatherPoint/GroupPoint/ThreeInterpolate; they are training
 *     code and not part of this path).
 *
 * Each declaration cites the reference interface it replaces (paths relative to
 * the reference repository root).
 */
#ifndef FPN_H
#define FPN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPN_OK 0
#define FPN_E_INVALID_ARG (-1)   /* bad size, null pointer, unsupported shape  */
#define FPN_E_LAUNCH (-2)        /* the HIP runtime rejected a launch          */
#define FPN_E_UNSUPPORTED (-3)   /* valid request outside the implemented range */

/* ABI version of this header (bumped on any signature change). */
int fpn_abi_version(void);

/* Message for the last non-OK return on the calling thread ("" if none). */
const char *fpn_last_error_string(void);

/*
 * Farthest point sampling: out[i,0]=0, then m-1 greedy picks of the point with
 * the largest running min squared distance, ties to the LOWEST point index.
 * Replaces farthestpointsamplingLauncher(int b,int n,int m,const float* inp,
 * float* temp,int* out), models/tf_ops/sampling/tf_sampling.cpp:94 (kernel
 * tf_sampling_g.cu:105-170).  The (32,n) global scratch `temp` of the reference
 * does not exist here: running distances live in registers / LDS.
 *   inp (b,n,3) f32 -> out (b,m) i32.      Limits: 1 <= n <= 32768.
 */
int fpn_farthest_point_sample(int b, int n, int m, const float *inp, int *out, void *stream);

/*
 * out[i,j,:] = inp[i,idx[i,j],:].  Replaces gatherpointLauncher(int b,int n,
 * int m,const float* inp,const int* idx,float* out), tf_sampling.cpp:125
 * (kernel tf_sampling_g.cu:172-181).   inp (b,n,3), idx (b,m) -> out (b,m,3).
 */
int fpn_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out,
                     void *stream);

/*
 * Ball query: for each query j of xyz2 the first `nsample` dataset indices k (in
 * index order) with max(sqrtf(|xyz2[j]-xyz1[k]|^2),1e-20f) < radius; a short
 * row is padded with its first hit; pts_cnt[j] = hits (<= nsample).  An empty
 * ball gives an all-zero row and pts_cnt 0 (the reference leaves that row
 * unwritten).  Replaces queryBallPointLauncher(int b,int n,int m,float radius,
 * int nsample,const float* xyz1,const float* xyz2,int* idx,int* pts_cnt),
 * models/tf_ops/grouping/tf_grouping.cpp:66 (kernel tf_grouping_g.cu:3-36; CPU
 * twin query_ball_point.cpp:19-47).
 *   xyz1 (b,n,3), xyz2 (b,m,3) -> idx (b,m,nsample) i32, pts_cnt (b,m) i32.
 *   pts_cnt may be NULL.
 */
int fpn_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                         const float *xyz2, int *idx, int *pts_cnt, void *stream);
/*
 * fpn_query_ball_point for the nr (1..4) radii of a multi-scale set abstraction in one
 * launch (models/pointnet_util.py:176-178 queries the same xyz / new_xyz once per
 * radius): radii[r], nsamples[r] (HOST arrays) -> idx[r] (b,m,nsamples[r]) and, where
 * pts_cnt and pts_cnt[r] are not NULL, pts_cnt[r] (b,m) (HOST arrays of device pointers).
 * Every output equals the single-radius call's.
 */
int fpn_query_ball_point_multi(int b, int n, int m, int nr, const float *radii, const int *nsamples,
                               const float *xyz1, const float *xyz2, int *const *idx,
                               int *const *pts_cnt, void *stream);

/*
 * out[i,j,k,:] = points[i,idx[i,j,k],:].  Replaces groupPointLauncher(int b,
 * int n,int c,int m,int nsample,const float* points,const int* idx,float* out),
 * tf_grouping.cpp:142 (kernel tf_grouping_g.cu:40-57; CPU twin
 * query_ball_point.cpp:52-66).  points (b,n,c), idx (b,m,nsample) ->
 * out (b,m,nsample,c).
 */
int fpn_group_point(int b, int n, int c, int m, int nsample, const float *points,
                    const int *idx, float *out, void *stream);

/*
 * Grouping of a set-abstraction module in one pass:
 *   xyz_first != 0: out[i,j,k,:] = [xyz[i,t,:] - new_xyz[i,j,:] | points[i,t,:]]
 *   xyz_first == 0: out[i,j,k,:] = [points[i,t,:] | xyz[i,t,:] - new_xyz[i,j,:]]
 * with t = idx[i,j,k].  Replaces the group_point(xyz) / subtract / group_point(points)
 * / concat sequence of sample_and_group (models/pointnet_util.py:44-49, xyz first)
 * and of pointnet_sa_module_msg (:178-183, features first).  points may be NULL
 * with c == 0 (output is the relative coordinates only, :53 / :185).
 *   xyz (b,n,3), new_xyz (b,m,3), points (b,n,c), idx (b,m,nsample) ->
 *   out (b,m,nsample,3+c).
 */
int fpn_group_concat(int b, int n, int m, int nsample, int c, int xyz_first, const float *xyz,
                     const float *new_xyz, const float *points, const int *idx, float *out,
                     void *stream);

/*
 * Three nearest neighbours by squared distance, ascending, ties to the lowest
 * index; with m < 3 the missing slots are (+inf, 0).  Replaces threenn_cpu(int b,
 * int n,int m,const float* xyz1,const float* xyz2,float* dist,int* idx),
 * models/tf_ops/3d_interpolation/tf_interpolate.cpp:60-103 (the reference runs
 * this op on the host only).  xyz1 (b,n,3), xyz2 (b,m,3) -> dist (b,n,3) f32,
 * idx (b,n,3) i32.
 */
int fpn_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                 int *idx, void *stream);

/*
 * out[i,j,l] = p[i,idx0,l]*w0 + p[i,idx1,l]*w1 + p[i,idx2,l]*w2 (products
 * rounded, summed left to right).  Replaces threeinterpolate_cpu(int b,int m,
 * int c,int n,const float* points,const int* idx,const float* weight,float* out),
 * tf_interpolate.cpp:107-127.  points (b,m,c), idx (b,n,3), weight (b,n,3) ->
 * out (b,n,c).
 */
int fpn_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                          const float *weight, float *out, void *stream);

/*
 * Fused feature propagation front half: three_nn(xyz1,xyz2), the inverse-
 * distance weights of models/pointnet_util.py:211-214 (dist=max(dist,1e-10),
 * w=(1/dist)/sum(1/dist)), and three_interpolate(points2,idx,w) in one kernel;
 * the neighbour table never goes to memory unless asked for.  Same results as
 * the three separate steps.  dist / idx may be NULL (not written).
 *   xyz1 (b,n,3), xyz2 (b,m,3), points2 (b,m,c) -> out (b,n,c),
 *   optionally dist (b,n,3), idx (b,n,3).
 * No single counterpart in the reference: it is pointnet_fp_module's
 * lines :210-215 as one call.
 */
int fpn_three_nn_interpolate(int b, int n, int m, int c, const float *xyz1, const float *xyz2,
                             const float *points2, float *out, float *dist, int *idx,
                             void *stream);
/* fpn_three_nn_interpolate with the result rounded to bf16 (round to nearest even) on the
 * way out: out (b,n,c) bf16; no neighbour table.  For a consumer that reads bf16 operands
 * (fpn_fp_head_chain_bf16x).  Same reference lines as fpn_three_nn_interpolate. */
int fpn_three_nn_interpolate_bf16(int b, int n, int m, int c, const float *xyz1, const float *xyz2,
                                  const float *points2, void *out, void *stream);

/*
 * Partial selection sort of each row of dist (b,m,n): out = dist with the k
 * smallest values of the row moved (by the reference's swap sequence) to the
 * first k columns, outi = the matching permutation of 0..n-1.  Replaces
 * selectionSortLauncher(int b,int n,int m,int k,const float* dist,int* outi,
 * float* out), tf_grouping.cpp:108 (kernel tf_grouping_g.cu:83-123; CPU twin
 * selection_sort.cpp:20-63).
 */
int fpn_select_top_k(int b, int n, int m, int k, const float *dist, int *outi, float *out,
                     void *stream);

/*
 * k nearest dataset points of every query, knn_point(k, xyz1, xyz2) of
 * models/tf_ops/grouping/tf_grouping.py:49-74 as one call: xyz1 (b,n,c) dataset, xyz2
 * (b,m,c) queries -> val (b,m,k) squared distances, idx (b,m,k).  The reference builds
 * the (b,m,n) matrix  reduce_sum((xyz1 - xyz2)**2, -1)  with TensorFlow ops (:65-67), sorts
 * it with selectionSortLauncher (:69, tf_grouping.cpp:108) and slices the first k columns
 * (:70-71).  Here the row of a query is formed in LDS - sum over the channels, left to
 * right, of the rounded squares of the differences - the same k swap steps run on it
 * (strict "<", lowest position among equal values), and only the k columns leave: neither
 * the matrix nor the two (b,m,n) arrays exist.  1 <= k <= n <= 16384.
 */
int fpn_knn_point(int b, int n, int c, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                  void *stream);

/*
 * One shared-MLP layer (1x1 convolution / fully connected) with everything the
 * reference applies after it fused in:
 *     acc  = x[r, 0:cin] . w            (fp32 MFMA, exact fp32 FMA chain)
 *     acc += group_bias[r / rows_per_group, :]        if group_bias != NULL
 *     v    = acc * scale[:] + shift[:]                (NULL scale = 1, NULL shift = 0)
 *     v    = max(v, 0)                                if relu
 *     pool_rows == 0 : y[r, :] = v                    y is (rows, ldy)
 *     pool_rows  > 0 : y[g, :] = max over the pool_rows consecutive rows of
 *                      group g                        y is (rows/pool_rows, ldy)
 * scale/shift carry bias + inference batch norm folded by the caller:
 * scale = gamma/sqrt(var+eps), shift = beta + (bias-mean)*scale.
 * Replaces, per layer, tf_util.conv2d (models/tf_util.py:120-185: conv2d :170,
 * bias_add :176, batch norm :178-181, ReLU :183-184), tf_util.fully_connected
 * (:327-363) and, with pool_rows, the tf_util.max_pool2d([N,1]) that follows
 * (:366-389; models/frustum_pointnets_v1.py:61,124; model_util.py:347) or the
 * reduce_max over nsample of pointnet_util.py:126,192.
 *   x (rows, ldx) with ldx >= cin; w (cin, cout) row-major; y row stride ldy >=
 *   cout (lets a layer write into a slice of a wider concat buffer).
 *   pool_rows must be a multiple of 32 and divide rows.
 *   workspace: needed only when pool_rows > 128; at least
 *   fpn_shared_mlp_workspace_bytes(rows, cout, pool_rows) bytes.
 */
int fpn_shared_mlp_f32(long rows, int cin, int cout, const float *x, int ldx, const float *w,
                       const float *scale, const float *shift, const float *group_bias,
                       int rows_per_group, int relu, int pool_rows, float *y, int ldy,
                       void *workspace, size_t workspace_bytes, void *stream);

size_t fpn_shared_mlp_workspace_bytes(long rows, int cout, int pool_rows);

/*
 * fpn_shared_mlp_f32 on the bf16 matrix instruction (fp32 accumulation, fp32 epilogue):
 * the same layer - tf_util.conv2d / conv1d of models/tf_util.py:120-185 as the
 * PointNet++ modules use it (models/pointnet_util.py:116-126,142-150,222-226) - for wide
 * per-point layers whose inputs are FEATURES (x is rounded to bf16 on the way in:
 * coordinates in metres would lose their decimetres).  Results differ from the fp32
 * entry point by bf16 rounding of both operands (2^-9 relative each).
 *   wt: the weights transposed and converted by fpn_mlp_pack_bf16: (cout, ldw) bf16,
 *   ldw = cin rounded up to a multiple of 64, zeros beyond cin; 16-byte aligned.
 *   Everything else as fpn_shared_mlp_f32.
 * Two kernels, one result: with no pooling, at least 131072 rows, cin a multiple of 4 and at
 * most 448 and x 16-byte aligned with ldx a multiple of 4, persistent workgroups keep their
 * columns of wt in LDS and stream the whole 128-row blocks of x through one pipeline (a
 * ragged tail of rows goes to the per-tile kernel in a second launch - with a group bias
 * only if the tail starts on a group boundary); every other layer runs the per-tile kernel.  Both use the same operand slots, k order and epilogue
 * arithmetic: the choice never changes a bit of y.  FPN_BF16_MLP=tile in the environment
 * keeps the per-tile kernel everywhere (a measurement switch).
 */
int fpn_mlp_pack_bf16(int cin, int cout, const float *w, void *wt, void *stream);
int fpn_shared_mlp_bf16(long rows, int cin, int cout, const float *x, int ldx, const void *wt,
                        const float *scale, const float *shift, const float *group_bias,
                        int rows_per_group, int relu, int pool_rows, float *y, int ldy,
                        void *workspace, size_t workspace_bytes, void *stream);

/*
 * fpn_shared_mlp_f32 for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): same arguments
 * minus the workspace, same kernels, same results bit for bit.  The partial maxima of a
 * layer with pool_rows > 128 go to a buffer of FPN_STATIC_SCRATCH_BYTES that lives in the
 * library's device image (one copy per GPU, held until the process ends) instead of a
 * caller workspace; a call that would need more (fpn_shared_mlp_workspace_bytes(rows,
 * cout, pool_rows) > FPN_STATIC_SCRATCH_BYTES) is refused with FPN_E_INVALID_ARG.  The
 * call allocates and frees nothing.  All calls on one GPU share that GPU's buffer: calls
 * that may overlap on one GPU (different streams or threads) race and must be ordered
 * by the caller.  The buffer's address is looked up on the first call on each device
 * (make that call outside graph capture).
 * No counterpart in the reference (TensorFlow allocates every temporary there).
 */
#define FPN_STATIC_SCRATCH_BYTES (8u << 20)
int fpn_shared_mlp_static_f32(long rows, int cin, int cout, const float *x, int ldx, const float *w,
                              const float *scale, const float *shift, const float *group_bias,
                              int rows_per_group, int relu, int pool_rows, float *y, int ldy,
                              void *stream);

/*
 * y[g,:] = max over the pool_rows consecutive rows of group g of x (rows, ldx):
 * tf_util.max_pool2d with kernel [N,1] (models/tf_util.py:366-389) /
 * tf.reduce_max over the sample axis (models/pointnet_util.py:126,192) as a
 * stand-alone op, for callers that did not fuse it into the layer before it.
 * Any pool_rows >= 1 dividing rows.  y is (rows/pool_rows, ldy).
 */
int fpn_max_pool_rows(long rows, int c, int pool_rows, const float *x, int ldx, float *y, int ldy,
                      void *stream);

/*
 * Flipout form of the same layer (models/FPointNet_Bayes.py:54-68, which calls
 * tfp.layers.Convolution2DFlipout / DenseFlipout):
 *     acc = x . w_mean + ((x * sign_in[g]) . w_delta[s]) * sign_out[g]
 * with g = r / rows_per_group (one sign pair per example) and s =
 * r / rows_per_sample (one weight perturbation per Monte-Carlo sample, so S
 * samples are batched as S*rows_per_sample rows of one call); group_bias
 * (rows/rows_per_group, cout), if not NULL, is added to acc; then the same
 * scale/shift/relu/pool epilogue as fpn_shared_mlp_f32.  When rows_per_group is a
 * multiple of 128 the perturbation is folded into the weight tile
 * (w_mean + sign_in^T * w_delta * sign_out, exact because the signs are +-1) and the
 * layer costs ONE product instead of the two of the textbook form.
 *   w_delta (nsamples, cin, cout); sign_in (rows/rows_per_group, cin) and
 *   sign_out (rows/rows_per_group, cout) hold +1/-1 as f32.
 */
int fpn_shared_mlp_flipout_f32(long rows, int cin, int cout, const float *x, int ldx,
                               const float *w_mean, const float *w_delta, long rows_per_sample,
                               const float *sign_in, const float *sign_out, int rows_per_group,
                               const float *group_bias, const float *scale, const float *shift,
                               int relu, int pool_rows, float *y, int ldy, void *workspace,
                               size_t workspace_bytes, void *stream);

/* Largest squared distance s for which the reference's ball test
 * max(sqrtf(s), 1e-20f) < radius holds (models/tf_ops/grouping/tf_grouping_g.cu:24),
 * or -1 when no distance passes, so that fpn_query_ball_point can compare squared distances and still accept
 * exactly the same points. */
float fpn_ball_threshold(float radius);

/*
 * First half of the v1 instance-segmentation net as ONE kernel: conv1..conv5 of
 * get_instance_seg_v1_net (models/frustum_pointnets_v1.py:41-60) with their
 * bias/batch-norm/ReLU, and the max-pool over the frustum's points at :61.
 * Each wave carries 32 points through all five layers with the activations in
 * registers (see fpn_v1_seg_tail_f32); only the point cloud is read and only
 * conv3's output (`point_feat`, :49, needed again by conv6) and the maxima are
 * written.  The maximum over a frustum's points is accumulated with atomic max
 * operations (ReLU output is >= 0, so integer atomics on the bit patterns order
 * correctly and the result does not depend on the order of arrival).
 *   pc (rows, cpc) with 3 <= cpc <= 8;  rows = frustums * n, n % 32 == 0.
 *   layers: HOST array of 15 device pointers {w,scale,shift} x {conv1..conv5};
 *           conv1's w row-major (cpc,64), the other four in
 *           fpn_tile_weights_reg_f32 layout with jmajor = 0; scale/shift as in
 *           fpn_shared_mlp_f32 (all non-NULL).
 *   point_feat (rows,64).
 *   global_max (rows/n, ldg), ldg >= 1024: columns 0..1023 of row f receive
 *           max(previous content, max over the points of frustum f of conv5's
 *           output) - the caller ZEROES those columns before the call (or passes a
 *           running maximum, e.g. when a frustum's points arrive in several calls).
 *   per_frustum != 0 (this and the two entry points below): every WEIGHT array of
 *           `layers` holds one copy per frustum, back to back (frustum f's copy of a
 *           (cin,cout) layer at float offset f*cin*cout) - the effective weights of a
 *           Flipout pass, see fpn_flipout_weights_f32; scale / shift stay shared.
 */
int fpn_v1_seg_head_f32(long rows, int n, int cpc, const float *pc, const float *const *layers,
                        int per_frustum, float *point_feat, float *global_max, int ldg,
                        void *stream);

/*
 * Flipout (tfp.layers.Convolution2DFlipout / DenseFlipout as used by
 * models/FPointNet_Bayes.py:54-68) for rows that share their sign vectors - all
 * points of a frustum do: y = x.W_mean + ((x*s_in).W_delta)*s_out is the plain
 * product with the frustum's EFFECTIVE weights
 *   W_eff[f] = W_mean + diag(s_in[f]) . W_delta . diag(s_out[f])        (signs are +-1)
 * so one Monte-Carlo pass through the layer-chain kernels is: build W_eff for every
 * frustum and chain layer, then call the fpn_v1_* kernels with per_frustum = 1.
 *   w_mean (cin,cout) row-major; sign_in (b, ld_in >= cin), sign_out (b, ld_out >=
 *   cout), entries +-1.  Several Monte-Carlo samples can share one call, their
 *   frustum copies stacked along b: copy f uses the perturbation matrix
 *   w_delta + (f / frustums_per_delta) * delta_stride (floats; each (cin,cout)
 *   row-major; frustums_per_delta = b and any stride for a single sample).
 *   out (b, cin*cout): frustum f's matrix in `layout` 0 = row-major, 1 = the chunk
 *   layout of fpn_tile_weights_reg_f32 with jmajor = 0, 2 = with jmajor = 1 (cin, cout
 *   multiples of 32 for 1 and 2).  b <= 65535.
 */
int fpn_flipout_weights_f32(int b, int cin, int cout, const float *w_mean, const float *w_delta,
                            int frustums_per_delta, long delta_stride, const float *sign_in,
                            int ld_in, const float *sign_out, int ld_out, int layout, float *out,
                            void *stream);

/*
 * fpn_flipout_weights_f32 for the chunked layouts (1, 2) with the signs as ONE BIT each (set = +1, as
 * fpn_flipout_signs spreads them): frustum f's input sign k is bit in_bit0 + f * ld_in + k of the 32-bit words at
 * `bits`, its output sign c bit out_bit0 + f * ld_out + c (in_bit0 and ld_in multiples of 4).  The same matrices,
 * bit for bit.  No counterpart in the reference (models/FPointNet_Bayes.py:54-68 keeps its signs inside
 * tensorflow_probability's layers).
 */
int fpn_flipout_weights_bits_f32(int b, int cin, int cout, const float *w_mean, const float *w_delta,
                                 int frustums_per_delta, long delta_stride, const void *bits, long in_bit0, int ld_in,
                                 long out_bit0, int ld_out, int layout, float *out, void *stream);

/*
 * Flipout's sign vectors from random bits: signs[i] = +1 where bit (i % 32) of the 32-bit word bits[i / 32] is
 * set, else -1 (i < n; bits holds (n + 31) / 32 words, 4-byte aligned; signs 16-byte aligned).  The layers of
 * models/FPointNet_Bayes.py:54-68 draw theirs inside tensorflow_probability (a Rademacher draw per input and
 * per output channel of every example); here the caller's generator supplies one bit per sign and this call
 * spreads them - any fair bits give the same distribution.
 */
int fpn_flipout_signs(long n, const void *bits, float *signs, void *stream);

/*
 * Weight layout of the fpn_v1_* chain kernels, whose waves keep all activations
 * in registers and feed the matrix cores' weight operand straight from memory:
 * one 4 KiB chunk per (32-row input tile j, 32-column output tile t) with
 *   chunk[(q*64 + l)*4 + i] = w[32j + 8q + 4(l/32) + i][32t + l%32]
 * (l = lane 0..63; q, i = 0..3), chunk (t, j) stored at index t*(cin/32) + j, or,
 * with jmajor != 0, at j*(cout/32) + t.  cin, cout multiples of 32; w_tiled has
 * cin*cout floats and must not alias w.  No counterpart in the reference.
 */
int fpn_tile_weights_reg_f32(int cin, int cout, int jmajor, const float *w, float *w_tiled,
                             void *stream);

/*
 * Second half of the same net as ONE kernel: conv6..conv10
 * (models/frustum_pointnets_v1.py:68-89; dropout :84 is the identity at
 * inference).  conv6's input is [point_feat | global_feat | one_hot] (:64-66);
 * the part of the product that is the same for every point of a frustum comes in
 * folded into conv6's shift:
 *   group_shift[f,:] = ([global_feat | one_hot][f] . W6[64:1091]) * scale6 + shift6
 * (one row per frustum - fpn_shared_mlp_f32 with scale/shift and relu = 0 produces
 * exactly this), the kernel multiplies only point_feat by W6[0:64].  Each wave
 * carries 32 points from conv6 to conv10 with every activation in registers: layers
 * are computed transposed (Y^T = W^T X^T), which leaves a layer's output in the
 * register layout the next layer's MFMA reads its activation operand from, so
 * nothing between point_feat and the logits touches LDS or memory.
 *   point_feat (rows,64); group_shift (rows/n, 512); n % 32 == 0.
 *   layers: HOST array of 13 device pointers {w6[0:64], scale6, w7, scale7, shift7,
 *           w8, scale8, shift8, w9, scale9, shift9, w10, bias10} (all non-NULL);
 *           w6[0:64] and w7 in fpn_tile_weights_reg_f32 layout with jmajor = 0,
 *           w8 and w9 with jmajor = 1, w10 row-major (128,2).
 *   drop_mask: NULL, or (rows,4) 32-bit words, 16-byte aligned: Monte-Carlo dropout before
 *           the 2-channel head (models/FPointNet_Bayes.py:92, layers.Dropout(0.5) kept
 *           active): bit c of word j of a point keeps conv9's channel 32j + c; kept
 *           activations are doubled (1 / (1 - rate)), the others dropped.
 *   logits (rows,2).
 */
int fpn_v1_seg_tail_f32(long rows, int n, const float *point_feat, const float *group_shift,
                        const float *const *layers, int per_frustum, const void *drop_mask,
                        float *logits, void *stream);

/*
 * Convolution stack over the object points as ONE kernel, 32 points per wave,
 * activations in registers:
 *   box == 0: T-Net, conv-reg1-stage1..conv-reg3-stage1 (3->128->128->256) of
 *             get_center_regression_net (models/model_util.py:335-346) and its
 *             max-pool over the object points (:347);
 *   box != 0: box net, conv-reg1..conv-reg4 (3->128->128->256->512) of
 *             get_3d_box_estimation_v1_net (models/frustum_pointnets_v1.py:108-123)
 *             and its max-pool (:124).
 * `delta` (rows/m, 3), if not NULL, is subtracted from every point of the frustum
 * as it is loaded: the translation to the T-Net's centre estimate of
 * frustum_pointnets_v1.py:176-177.
 *   obj (rows,3), rows = frustums * m, m % 32 == 0.
 *   layers: HOST array of 9 (T-Net) or 12 (box) device pointers {w,scale,shift};
 *           the first layer's w row-major (3,128), the others in
 *           fpn_tile_weights_reg_f32 layout with jmajor = 0.
 *   feat_max (rows/m, ldg), ldg >= 256 (T-Net) or 512 (box): running maximum as in
 *           fpn_v1_seg_head_f32 - the caller zeroes columns 0..255 / 0..511 first.
 */
int fpn_v1_obj_chain_f32(long rows, int m, int box, const float *obj, const float *delta,
                         const float *const *layers, int per_frustum, float *feat_max, int ldg,
                         void *stream);

/*
 * Set-abstraction module with max pooling as ONE kernel: the grouping of
 * sample_and_group (models/pointnet_util.py:36-51: group_point of xyz, translation
 * to the centroid, group_point of the features, concat) or of the MSG loop
 * (:176-186), the three conv2d layers of the module's MLP (:116-121 / :187-191)
 * and the max over the nsample axis (:126 / :192).  A wave takes 32 consecutive
 * samples of one ball and keeps every activation in registers (see
 * fpn_v1_seg_tail_f32); the (b,m,nsample,3+cf) grouped tensor is never written.
 *   xyz (b,n,3), new_xyz (b,m,3), feat (b,n,cf) or NULL, idx (b,m,ns) from
 *   fpn_query_ball_point; ns % 32 == 0.
 *   Layer widths c1, c2 (multiples of 64; c2 = 96 is also built for the c1 = 64, c3 = 128,
 *   cf < 32 shape) and c3 (multiple of 128) are the PADDED
 *   widths: a narrower layer is padded by the caller with zero weight columns and
 *   zero scale / shift.  The narrow stack 32,32,64 over cf <= 5 channels has a kernel at its
 *   own widths.  layers: HOST array of 9 device pointers {w,scale,shift} x 3:
 *     cf < 32 : w1 row-major (3+cf, c1), rows ordered [dx,dy,dz, features];
 *     cf >= 32: cf % 32 == 0, w1 = fpn_tile_weights_reg_f32(jmajor = 0) of the
 *               (cf+32, c1) matrix whose rows are [features, dx,dy,dz, 29 zero rows];
 *     w2 (c1,c2), w3 (c2,c3) in fpn_tile_weights_reg_f32 layout, jmajor = 0.
 *   out (b*m, ldo): columns [0, c3_valid) of row (frustum, ball) receive the running
 *   maximum (atomic max; the caller zeroes them first - unless fpn_sa_chain_overwrites says
 *   that this call runs a persistent kernel, which stores whole rows instead).
 *   c3_valid: a multiple of 64, <= c3.
 * fpn_sa_chain_supported tells whether a kernel exists for the padded shape
 * (FPN_E_UNSUPPORTED from fpn_sa_chain_f32 otherwise: use the per-layer path).
 */
int fpn_sa_chain_supported(int cf, int c1, int c2, int c3);
int fpn_sa_chain_f32(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                     const float *new_xyz, const float *feat, const int *idx,
                     const float *const *layers, float *out, int ldo, int c3_valid, void *stream);
/* The same with the wide point features taken out of the kernel.  A set-abstraction layer
 * is linear before its batch norm and ReLU and a point is gathered into many balls
 * (models/pointnet_util.py:43-49, :177-184 group and concatenate, :187-190 multiply), so the
 * product of a point's features with their rows of the first layer is the same in every
 * ball: the caller computes it once per point - pre (b*n, ld_pre >= c1) fp32 = features .
 * W1[feature rows], no scale, no shift, c1 the padded width, 16-byte aligned, ld_pre % 4 == 0
 * (the products of all scales of a multi-scale module can be the column blocks of one
 * matrix, made by one launch) - and the kernel gathers
 * pre[idx] as the first layer's starting value, adds the relative coordinates' (and up to 5
 * direct feature channels') part, then scale, shift and ReLU as before.  cf, feat, layers:
 * as for cf < 32 above (w1 holds the rows of the coordinates and direct channels only).
 * Besides the cf < 32 shapes of fpn_sa_chain_supported, widths 128,128,256. */
int fpn_sa_chain_pre_f32(int b, int n, int m, int ns, int cf, int c1, int c2, int c3,
                         const float *xyz, const float *new_xyz, const float *feat, const float *pre,
                         int ld_pre, const int *idx, const float *const *layers, float *out, int ldo,
                         int c3_valid, void *stream);

/*
 * Per-point head after a feature-propagation step as ONE kernel: the MLP of
 * pointnet_fp_module (models/pointnet_util.py:218-226) on the concatenation
 * [interpolated features | the point's own features] followed by two 1-wide
 * convolutions (a conv1d + batch norm + ReLU layer and the final 2-channel conv1d of a
 * PointNet++ segmentation net), all hidden layers 128 wide:
 *   [xa (ca = 128) | xb (cb <= 32)] -> 128 -> 128 -> 128 -> 2.
 * The concatenated tensor is never written; every activation stays in registers.
 *   xa (rows, 128) 16-byte aligned, xb (rows, cb) or NULL when cb == 0; rows % 32 == 0.
 *   layers: HOST array of 11 device pointers {w1, scale1, shift1, w2, scale2, shift2,
 *   w3, scale3, shift3, w4, bias4}: w1 = fpn_tile_weights_reg_f32(jmajor = 0) of the
 *   (160,128) matrix whose rows are [xa's 128 | xb's cb | zero rows]; w2, w3 (128,128) in
 *   the same layout; w4 (128,2) row-major.  logits (rows,2).
 */
int fpn_fp_head_chain_f32(long rows, int ca, int cb, const float *xa, const float *xb,
                          const float *const *layers, float *logits, void *stream);
/* The same head with the interpolated features' share of the first layer as per-point
 * products: three_interpolate (models/pointnet_util.py:210-215) and the first convolution
 * before its batch norm (:222-226) are both linear, so the caller multiplies the KNOWN points'
 * features with their 128 rows of W1 once (no scale, no shift), interpolates those products to
 * the query points - pre (rows,128), 16-byte aligned - and this kernel starts the first layer
 * from them and adds the own channels' part.  cb <= 8.  layers: 11 device pointers {w1b,
 * scale1, shift1, w2, scale2, shift2, w3, scale3, shift3, w4, bias4}: w1b (cb,128) row-major =
 * the own channels' rows of W1 (ignored when cb == 0), the rest as above. */
int fpn_fp_head_chain_pre_f32(long rows, int cb, const float *pre, const float *xb,
                              const float *const *layers, float *logits, void *stream);

/*
 * The same layer chains on the bf16 matrix instruction (fp32 accumulation): the
 * shared MLPs of models/frustum_pointnets_v1.py:41-88,108-124, models/model_util.py:
 * 335-347 and models/pointnet_util.py:116-126,187-192 with bf16 weights and bf16
 * activations between the layers.  Kept in fp32: the first layer of every stack that
 * reads coordinates from memory, all batch-norm shifts, the maxima over points and
 * the last 128 -> 2 layer of the segmentation net.  Results differ from the fp32
 * entry points by bf16 rounding (2^-9 relative per operand).
 *
 * Weights reach these kernels as one STREAM per kernel: for every layer after the
 * first, fpn_chain_pack_bf16 turns w (cin,cout) fp32 - with the folded batch-norm
 * scale multiplied into its columns (scale may be NULL = ones) - into
 * cin*cout bf16 values in the order the kernel consumes them; the caller places the
 * layers of a kernel back to back (the order is given with each entry point) and
 * leaves fpn_chain_pad_units_bf16() * 4096 bytes of readable padding after the last
 * one (the kernels read ahead).  cin % 32 == 0, cout % 64 == 0.
 *   `consts`: HOST array of device pointers to the fp32 per-channel constants.  The
 * streams and every constant array must be 16-byte aligned (they are read as 16-byte
 * vectors); a stream or a copy's stream must be shorter than 2 GiB.
 */
int fpn_chain_pad_units_bf16(void);
int fpn_chain_pack_bf16(int cin, int cout, const float *w, const float *scale, void *units,
                        void *stream);
/*
 * Flipout (models/FPointNet_Bayes.py:54-68) on these chains: copy_vecs > 0 gives every
 * frustum its OWN stream, copy_vecs 16-byte vectors after the previous frustum's
 * (>= the stream's own length; padding only after the last copy), written layer by
 * layer by fpn_flipout_stream_bf16 - arguments as fpn_flipout_weights_f32, `units` =
 * where the layer starts inside the first copy's stream - and its own fp32 first-layer
 * (and conv10) matrix, (frustum, k, cout) back to back (fpn_flipout_weights_f32,
 * layout 0).  copy_vecs == 0: one stream for all frustums.
 */
int fpn_flipout_stream_bf16(long copies, int cin, int cout, const float *w_mean, const float *w_delta,
                            int frustums_per_delta, long delta_stride, const float *sign_in, int ld_in,
                            const float *sign_out, int ld_out, void *units, long copy_vecs,
                            void *stream);

/*
 * fpn_flipout_stream_bf16 with the signs as ONE BIT each (set = +1, as fpn_flipout_signs spreads them): copy f's
 * input sign k is bit in_bit0 + f * ld_in + k of the 32-bit words at `bits`, its output sign c bit
 * out_bit0 + f * ld_out + c (in_bit0 and ld_in multiples of 4).  Same units, bit for bit, as the float signs
 * give; the signs of a pass are then a few megabytes that stay in the cache instead of 4 bytes a sign read from
 * memory between the kernel's writes.  No counterpart in the reference (its signs live inside
 * tensorflow_probability's layers, models/FPointNet_Bayes.py:54-68).
 */
int fpn_flipout_stream_bits_bf16(long copies, int cin, int cout, const float *w_mean, const float *w_delta,
                                 int frustums_per_delta, long delta_stride, const void *bits, long in_bit0, int ld_in,
                                 long out_bit0, int ld_out, void *units, long copy_vecs, void *stream);
/* stream: conv2, conv3, conv4, conv5; consts {w1 (cpc,64), scale1, shift1, shift2..shift5};
 * point_feat (rows,64) bf16 in an internal operand order, only meaningful to
 * fpn_v1_seg_tail_bf16; global_max as in fpn_v1_seg_head_f32. */
int fpn_v1_seg_head_bf16(long rows, int n, int cpc, const float *pc, const void *wstream,
                         const float *const *consts, long copy_vecs, void *point_feat,
                         float *global_max, int ldg, void *stream);
/* stream: conv6 rows 0..63, conv7, conv8, conv9; consts {shift7, shift8, shift9, w10
 * (128,2), bias10}; group_shift and drop_mask as in fpn_v1_seg_tail_f32. */
int fpn_v1_seg_tail_bf16(long rows, int n, const void *point_feat, const float *group_shift,
                         const void *wstream, const float *const *consts, long copy_vecs,
                         const int *drop_mask, float *logits, void *stream);
/* stream: layer 2, layer 3 (, layer 4 when box != 0); consts {w1 (3,128), scale1,
 * shift1, shift2, shift3 (, shift4)}. */
int fpn_v1_obj_chain_bf16(long rows, int m, int box, const float *obj, const float *delta,
                          const void *wstream, const float *const *consts, long copy_vecs,
                          float *feat_max, int ldg, void *stream);
/* Shapes and padding rules of fpn_sa_chain_f32; c3_valid % 64 == 0.
 *   cf < 32 : stream: layer 2, layer 3; consts {w1 (3+cf,c1) fp32, scale1, shift1,
 *             shift2, shift3}; with c2 = 96 (c1 = 64, c3 = 128) the stream holds layer 2
 *             packed as (64,128) - columns 96..127 zero - and layer 3 as (96,128);
 *   cf >= 32: stream: layer 1 (rows [features, dx,dy,dz, a zero row, dx,dy,dz again, 25
 *             zero rows]: the kernel feeds each coordinate as two bf16 numbers, the
 *             second what the first leaves over), layer 2, layer 3; consts {shift1,
 *             shift2, shift3}.
 * fpn_sa_chain_bf16_supported: the shapes of fpn_sa_chain_supported whose widths suit the
 * stream, plus two group-all stacks (models/pointnet_util.py:62-84, :108-109: one ball of
 * all 32 points per frustum - idx = 0..31, new_xyz = 0 - over 640 features through
 * 128-256-1024 and over 256 features through 256-256-512). */
int fpn_sa_chain_bf16_supported(int cf, int c1, int c2, int c3);
/* The kernels of fpn_sa_chain_f32 / _pre_f32 (fpn_sa_chain_overwrites) and of fpn_sa_chain_bf16 /
 * _pre_bf16 / _bf16p (fpn_sa_chain_bf16_overwrites) come in two kinds: some MERGE their maxima
 * into out (atomic max on values >= 0), so out must start at zero; the persistent ones STORE
 * every ball's pooled row, all c3_valid columns of it, so out need not be initialised.  These
 * queries tell which kind a call with rows = b*m*ns and the same cf and widths runs: 1 = stores
 * whole rows, 0 = merges (or the shape has no kernel).  The answer depends on the process's
 * environment switches: ask at call time.  A zeroed out is always correct.  No counterpart in
 * the reference (tf.reduce_max of models/pointnet_util.py:128, :192 writes a new tensor). */
int fpn_sa_chain_overwrites(long rows, int cf, int c1, int c2, int c3);
int fpn_sa_chain_bf16_overwrites(long rows, int cf, int c1, int c2, int c3);
int fpn_sa_chain_bf16(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                      const float *new_xyz, const float *feat, const int *idx, const void *wstream,
                      const float *const *consts, float *out, int ldo, int c3_valid, void *stream);
/* fpn_sa_chain_pre_f32's form on bf16 operands: pre (b*n, ld_pre) fp32 = features . (W1[feature
 * rows] . diag(scale1)) - scaled like the stream's weights, no shift - gathered as the first
 * layer's starting value; cf <= 1 direct channels; consts and stream as for cf < 32; widths as
 * there, and 128,128,256. */
int fpn_sa_chain_pre_bf16(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                          const float *new_xyz, const float *feat, const float *pre, int ld_pre,
                          const int *idx, const void *wstream, const float *const *consts, float *out,
                          int ldo, int c3_valid, void *stream);
/* The same for cf >= 32 with the point features converted once instead of once per gathered
 * row: fpn_sa_pack_features_bf16 turns feat (points, cf) fp32 (cf % 32 == 0, 16-byte
 * aligned) into `packed` (points * cf bf16, an internal operand order, meaningful only to
 * fpn_sa_chain_bf16p).  Same results bit for bit; half the gathered bytes (the modules of a
 * multi-scale set abstraction gather the same table several times,
 * models/pointnet_util.py:176-183). */
int fpn_sa_pack_features_bf16(long points, int cf, const float *feat, void *packed, void *stream);
int fpn_sa_chain_bf16p(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                       const float *new_xyz, const void *feat_packed, const int *idx,
                       const void *wstream, const float *const *consts, float *out, int ldo,
                       int c3_valid, void *stream);
/* fpn_fp_head_chain_pre_f32 on bf16 operands: pre (rows,128) bf16 = the interpolated products
 * of the known points' features with W1[their rows] . diag(scale1) (fpn_shared_mlp_bf16 then
 * fpn_three_nn_interpolate_bf16); stream: layer 2, layer 3; consts {w1b (cb,128) fp32, scale1,
 * shift1, shift2, shift3, w4 (128,2), bias4}; cb <= 8. */
int fpn_fp_head_chain_pre_bf16(long rows, int cb, const void *pre_bf16, const float *xb,
                               const void *wstream, const float *const *consts, float *logits,
                               void *stream);
/* fpn_fp_head_chain_f32 in bf16: stream: layer 1 (the padded (160,128) matrix), layer 2,
 * layer 3; consts {shift1, shift2, shift3, w4 (128,2) fp32, bias4}. */
int fpn_fp_head_chain_bf16(long rows, int ca, int cb, const float *xa, const float *xb,
                           const void *wstream, const float *const *consts, float *logits,
                           void *stream);
/* The same with the interpolated features already in bf16 (fpn_three_nn_interpolate_bf16's
 * output: (rows, ca) bf16 row-major, 16-byte aligned): half the bytes between the two
 * kernels, the same logits bit for bit (the kernel rounds fp32 rows to bf16 itself). */
int fpn_fp_head_chain_bf16x(long rows, int ca, int cb, const void *xa_bf16, const float *xb,
                            const void *wstream, const float *const *consts, float *logits,
                            void *stream);

/*
 * Mask, masked centroid and object-point gather of point_cloud_masking +
 * tf_gather_object_pc (models/model_util.py:275-319, :55-86), on the device
 * (the reference leaves the graph for numpy through tf.py_func, :84):
 *   mask[i,j]      = logits[i,j,0] < logits[i,j,1]            (as 0/1 f32)
 *   mean[i,:]      = sum_j mask*xyz / max(count,1)
 *   sel[i,0:npick] = indices of the object points.  count <= npick: the positives in
 *                    index order, repeated cyclically; all zero when count == 0.  (The
 *                    reference draws this set with numpy's global RNG; for count <=
 *                    npick the SET is the same, and the nets after it are invariant to
 *                    order and repetition.)  count > npick, where the reference draws a
 *                    random subset: policy FPN_PICK_FIRST takes the first npick
 *                    positives in index order, FPN_PICK_SPREAD positive number
 *                    floor(t * count / npick) for slot t - evenly spaced over the list,
 *                    so the subset covers the whole object whatever order the points
 *                    arrive in (raw sensor order is not random).
 *   obj[i,t,:]     = xyz[i,sel[i,t],:] - mean[i,:]
 *   point_cloud (b,n,cpc) with xyz in channels 0..2; logits (b,n,2).
 *   Outputs: mask (b,n) f32, mean (b,3), sel (b,npick) i32, obj (b,npick,3);
 *   mask and sel may be NULL.  Limits: n <= 16384, npick <= n or any npick when
 *   positives repeat.
 */
#define FPN_PICK_FIRST 0
#define FPN_PICK_SPREAD 1
int fpn_mask_gather(int b, int n, int cpc, int npick, int policy, const float *point_cloud,
                    const float *logits, float *mask, float *mean, int *sel, float *obj,
                    void *stream);

/*
 * 3-D IoU of box pairs as the detection evaluation defines it:
 * Box3D.get_iou (helpers/mAP_evaluation.py:176-182) = clip(I / (Va + Vb - I), 0, 1)
 * with I = overlap of the height intervals (:156-160) x area of the intersection
 * of the two ground rectangles (:162-174), the rectangles built from
 * rotation_matrix[0,0] and [0,1] of the box's quaternion (:117-132) and
 * V = width*length*height (:79).  All arithmetic in float64.
 *   A box is 10 doubles: translation x,y,z; size width,length,height; rotation
 *   quaternion w,x,y,z (the field order of the evaluation's dictionaries, :10-16).
 *   iou[i] = IoU(a + i*stride_a, b + i*stride_b), i < n; a stride of 0 compares
 *   every box of the other array with one box (get_ious, :259-260), otherwise
 *   strides are in doubles and >= 10.
 */
int fpn_box3d_iou_f64(long n, const double *a, long stride_a, const double *b, long stride_b,
                      double *iou, void *stream);

/*
 * The two factors of a box pair's intersection volume, for callers of the evaluation's
 * Box3D.get_area_intersection (helpers/mAP_evaluation.py:162-167: area of the overlap of
 * the two ground rectangles) and get_height_intersection / get_intersection (:156-160,
 * :169-174: overlap of the height intervals; their product is the volume).  Boxes, strides
 * and arithmetic as fpn_box3d_iou_f64; area (n) and height (n), either may be NULL.
 */
int fpn_box3d_intersection_f64(long n, const double *a, long stride_a, const double *b, long stride_b,
                               double *area, double *height, void *stream);

/*
 * The inner search of recall_precision (helpers/mAP_evaluation.py:282-297) for all
 * predictions at once: prediction i is compared with the gt_count[i] ground-truth
 * boxes gt[gt_begin[i] ...] of its own sample; best_iou[i] = the largest IoU,
 * best_j[i] = its index within that range (the first one on ties, as np.argmax);
 * gt_count[i] == 0 gives -inf and -1 (:282-283).  pred (npred,10), gt (*,10).
 */
int fpn_box3d_best_overlap_f64(int npred, const double *pred, const double *gt,
                               const int *gt_begin, const int *gt_count, double *best_iou,
                               int *best_j, void *stream);

/*
 * Monte-Carlo ensemble statistics over S sampled forward passes of the Bayesian
 * detector (the reference produces the samples with repeated `predict` calls of
 * models/FPointNet_Bayes.py's det_model and holds no code for the statistics; the
 * definitions are those of its project report, section 3: Shannon entropy of the
 * sample-mean class probabilities and total variance = trace of
 * C = 1/S sum v v^T - mean mean^T, eq. 8).
 *   box_output (s,b,59), stage1_center (s,b,3) -> stats (b,30):
 *     [0:7]   mean detection v = (cx,cy,cz, l,w,h, heading); per sample the centre is
 *             center_boxnet + stage1_center, the size is the mean size of the arg-max
 *             size cluster plus its residual, the heading is the centre of the
 *             arg-max heading bin plus its residual
 *     [7]     total variance of v
 *     [8],[9] entropy of the mean heading-bin / size-cluster probabilities (nats)
 *     [10:22] mean heading-bin probabilities, [22:30] mean size-cluster probabilities
 */
int fpn_mc_statistics(int s, int b, const float *box_output, const float *stage1_center,
                      float *stats, void *stream);

/*
 * Detection records: the host arithmetic that follows `model.predict` in the reference's
 * detection driver (scripts/test.py:145-160: soft-max scores, arg-max classes, residual
 * picks) and in its label writer (scripts/provider.py:366-373 from_prediction_to_label_format
 * = class2size :101-104, class2angle :73-82, rotate_pc_along_y :35-48, ty += h/2), as one
 * kernel over the network's outputs.
 *   mask_logits (b,n,2) f32, box_output (b,59) f32 (the box net's raw vector, sliced as
 *   models/model_util.py:211-243), stage1_center (b,3) f32, rot_angle (b) f32 or NULL (= 0:
 *   frustums that were not rotated to the centre view).
 *   records (b, FPN_DETECTION_RECORD) f64:
 *     [0:7]   h, w, l, tx, ty, tz, ry  - the label fields, float64 arithmetic on the fp32
 *             outputs as the reference's numpy code performs it; angle > pi wraps by -2pi
 *             before rot_angle is added; cos/sin of the fp32 angle are fp32 (correctly
 *             rounded here)
 *     [7]     score = (mean soft-max probability of the positive points + max heading
 *             probability + max size probability) / 3, soft-max in fp32, the mean in
 *             float64; NaN for a frustum without positive points (test.py:148 divides 0/0)
 *     [8],[9] heading_cls, size_cls (first maximum, as np.argmax)
 *     [10]    heading_res, [11:14] size_res (l,w,h)  (fp32 products of the model graph)
 *     [14:17] center = center_boxnet + stage1_center (frustum frame, fp32 sum)
 *     [17]    number of positive points, [18] max heading probability, [19] max size probability
 *   seg_bits (b, ceil(n/32)) u32 or NULL: bit (j mod 32) of word j/32 = 1 iff point j is an
 *   object point (np.argmax(logits, 2), test.py:162: logit 1 strictly above logit 0).
 * One wave per frustum; reads every input once.
 */
#define FPN_DETECTION_RECORD 20
int fpn_detection_records(int b, int n, const float *mask_logits, const float *box_output,
                          const float *stage1_center, const float *rot_angle, double *records,
                          unsigned *seg_bits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FPN_H */
