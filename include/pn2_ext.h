/*
 * pn2_ext.h -- MI355X-side extensions of libpn2_hip.so that have NO counterpart in the
 * reference's native module: work the reference does in Python/torch around the operator
 * stack, moved onto the device so a HandTrackNet forward has no host round trip and the
 * grouped tensors are never materialised.  Same conventions as pn2_hip.h (fp32/int32,
 * contiguous, caller allocates, async on `stream`, PN2_* return codes).
 */
#ifndef PN2_EXT_H
#define PN2_EXT_H

#include "pn2_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Batched rigid alignment  y ~= R x + t  of `num` 3-D point pairs (Kabsch / Horn).
 *   reference: network/models/hand_utils.py:42-66 (solve_rot_and_trans: 3x3 cross-covariance,
 *   torch.svd ON THE CPU with a device->host->device hop per forward, det-corrected rotation).
 * Here: one thread per batch element, Horn's unit-quaternion form (largest eigenvector of the
 * 4x4 symmetric matrix built from the cross-covariance) solved by cyclic Jacobi in fp64.
 * x: (xb, num, 3) with xb == b, or xb == 1 (one template shared by the whole batch);
 * y: (b, num, 3);  R: (b, 3, 3) row-major;  t: (b, 3, 1).
 */
int pn2x_kabsch(int b, int xb, int num, const float *x, const float *y, float *R, float *t, void *stream);
/*
 * Gradient of a loss with respect to y through the fit (R, t) = pn2x_kabsch(x, y), in closed form (no solver library, one
 * launch; what autograd computes through torch.svd in the reference's training losses, hand_network.py:186-190 via
 * hand_utils.py:42-66).  R (b,3,3) as returned by pn2x_kabsch; grad_R (b,3,3) / grad_t (b,3) may be NULL (= zero);
 * grad_y (b,num,3) is written.  x carries no gradient (the palm template).
 */
int pn2x_kabsch_backward(int b, int xb, int num, const float *x, const float *y, const float *R, const float *grad_R,
                         const float *grad_t, float *grad_y, void *stream);

/*
 * Hand frame in one launch: Kabsch fit of the palm template to kp[:, palm_idx] (num <= 16 indices, device int32)
 * followed by the canonicalisation of the cloud and of the keypoints,
 *   xyz2[b,i,:] = R_b^T (points[b,i,:] - t_b) / scale      xyz1[b,k,:] = R_b^T (kp[b,k,:] - t_b) / scale
 * (reference hand_network.py:100,118-119 + hand_utils.py:30-31,42-66: ransac_rt with a CPU SVD, torch.cat,
 * transpose, matmul, divide).  points (b,n,3), kp (b,j,3), palm_template (xb,num,3) with xb in {1,b};
 * outputs R (b,3,3), t (b,3,1), xyz2 (b,n,3), xyz1 (b,j,3), all point-major.
 */
int pn2x_hand_frame(int b, int xb, int num, int n, int j, const float *palm_template, const float *kp,
                    const int *palm_idx, const float *points, float scale, float *R, float *t, float *xyz2,
                    float *xyz1, void *stream);
/* ... and a second copy of xyz2 into three columns of a wider row buffer (row stride copy_ld floats), or NULL. */
int pn2x_hand_frame2(int b, int xb, int num, int n, int j, const float *palm_template, const float *kp,
                     const int *palm_idx, const float *points, float scale, float *R, float *t, float *xyz2,
                     float *xyz1, float *xyz2_copy, int copy_ld, void *stream);
/* ... and a per-cloud flag nonfinite[b] (device int32, b entries, or NULL): 1 when the cloud, the keypoints or the fit of
 * cloud b contain a NaN / Inf.  Handed to pn2x_pose_head2, it turns that frame's predicted keypoints into NaN: the
 * fused inference kernels drop NaNs in their ReLU / max-pool maxima (built -fno-honor-nans), whereas in the reference a
 * non-finite input point spreads through sampling, grouping and the global max-pool to every output of its frame. */
int pn2x_hand_frame3(int b, int xb, int num, int n, int j, const float *palm_template, const float *kp,
                     const int *palm_idx, const float *points, float scale, float *R, float *t, float *xyz2,
                     float *xyz1, float *xyz2_copy, int copy_ld, int *nonfinite, void *stream);

/*
 * Fused grouped MLP + max of one set-abstraction scale, eval mode (BatchNorm folded into the
 * 1x1 convolutions by the caller).  Replaces, for one (radius | kNN) scale, the reference's
 *   group(points, idx) / group(xyz, idx) - centre / cat / [Conv2d 1x1 + BN2d + ReLU] x3 / max over K
 * (pointnet_utils.py:389-403 for SA-MSG, :566-581 for GivenCenterPoints) without ever
 * materialising the (B, C, S, K) grouped tensors.
 *
 * Layer 1 is linear in its input [feat_j | xyz_j - c_s | centre_feat_s], so it is split:
 *   a1f  (b, n, a1f_ld) or NULL  per-point feature term  W1[:, feat] . feat_j  (point-major rows of
 *                            a1f_ld >= c1 floats; a GEMM over the n points instead of the s*k positions)
 *   xyz  (b, n, 3)  or NULL  point coordinates; with cxyz (b, s, 3) centroids and wx (c1, 3) =
 *                            W1[:, xyz] the kernel adds  wx . (xyz_j - c_s)  itself
 *   b1   (c1)       or NULL  layer-1 bias
 *   cadd (b, s, cadd_ld) or NULL  per-centroid term  W1[:, centre] . centre_feat_s
 * and for every centroid s and neighbour j = idx[b,s,k] (idx: (b, s, k) int32) computes
 *   h1 = relu(a1f[j] + wx.(xyz_j - c_s) + b1 + cadd_s);  h2 = relu(w2 h1 + b2);  h3 = w3 h2 + b3;
 *   out[b*out_b + s*out_s + c*out_c] = relu(max_k h3[c])
 * (out strides in floats: channel-major (b,c3,s) = {c3*s, 1, s}; point-major rows of a wider
 * (b, s, ld) buffer = {s*ld, ld, 1}, which lets the caller skip the reference's torch.cat).
 * w2 (c2, c1), w3 (c3, c2) row-major, BN folded.  fp32 throughout (matrix cores:
 * v_mfma_f32_16x16x4_f32, exact fp32).  Supported: k in {16,32,64} and (c1,c2,c3) in
 * {(32,32,64), (64,64,128), (128,128,192)} -- PN2_ERANGE otherwise (query with
 * pn2x_sa_mlp_max_supported; callers keep the unfused operator path for other shapes).
 */
int pn2x_sa_mlp_max(int b, int n, int s, int k, int c1, int c2, int c3, const float *a1f, int a1f_ld,
                    const float *xyz, const float *cxyz, const float *wx, const float *b1, const float *cadd,
                    int cadd_ld, const int *idx,
                    const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                    long out_b, int out_s, int out_c, void *stream);
int pn2x_sa_mlp_max_supported(int k, int c1, int c2, int c3);

/*
 * In-place y[b,c,n] = act(y[b,c,n] + bias[c]) (relu != 0 -> ReLU): the epilogue of an eval-mode,
 * BatchNorm-folded 1x1 convolution (reference: Conv1d + BatchNorm1d + ReLU of the feature-propagation
 * stacks, pointnet_utils.py:460-462) whose GEMM half is a library GEMM on the (b, c, n) tensor.
 */
int pn2x_bias_act(int b, int c, int n, float *y, const float *bias, int relu, void *stream);

/*
 * three_nn that returns interpolation WEIGHTS: idx (b,n,3) as pn2_three_nn, and
 *   weight[t] = (1/(sqrt(d2_t)+1e-8)) / sum_t (1/(sqrt(d2_t)+1e-8))
 * i.e. the sqrt of pointnet2_utils.py:135 plus the three torch kernels of the reference's caller
 * (pointnet_utils.py:447-449) folded into the search kernel's epilogue.  Needs m >= 3.
 */
int pn2x_three_nn_weights(int b, int n, int m, const float *unknown, const float *known, float *weight, int *idx,
                          void *stream);

/*
 * three_interpolate on POINT-MAJOR features: points (b, m, ldp), out (b, n, ldo), c <= ldp, ldo channels
 * per row; out[b,j,ch] = sum_t weight[b,j,t] * points[b, idx[b,j,t], ch]  (same arithmetic as
 * pn2_three_interpolate).  `out` may point at a column block of a wider row (ldo > c).
 */
/*
 * pn2x_three_nn_interpolate_pm: pn2x_three_nn_weights followed by pn2x_three_interpolate_pm in ONE launch, for callers that need
 * the (weight, index) pair for nothing else (the forward of feature propagation, pointnet_utils.py:440-453): out (b, n, ldo) rows
 * <- the inverse-distance blend of the three nearest `known` (b,m,3) points' rows of `points` (b, m, ldp), c columns.  Same floats
 * as the two calls.  PN2_ERANGE unless pn2x_three_nn_interpolate_pm_supported (2^14 <= b n < 2^18 queries, 16 <= m <= 2048 known points in
 * one LDS tile, c / ldp / ldo multiples of 4, points / out 16-byte aligned): the caller then runs the two launches.
 */
int pn2x_three_nn_interpolate_pm_supported(int b, int n, int m, int c, int ldp, int ldo);
int pn2x_three_nn_interpolate_pm(int b, int n, int m, int c, const float *unknown, const float *known, const float *points, int ldp,
                                 float *out, int ldo, void *stream);
int pn2x_three_interpolate_pm(int b, int c, int m, int n, const float *points, int ldp, const int *idx,
                              const float *weight, float *out, int ldo, void *stream);

/*
 * Two-level furthest point sampling without the second pass.  PointNet++ samples level 2 from level 1's samples
 * (reference backbones.py:98-104: FPS 1024 -> 256, then FPS 256 -> 128 over those 256).  FPS is greedy, and every
 * pick is the arg-max over ALL remaining points, so it is also the arg-max over the subset of picked points: as long
 * as no arg-max among the first m2 picks of level 1 was a tie, level 2's sample is exactly level 1's first m2 picks,
 * in order, i.e. idx2 = 0..m2-1 -- same distances, same min() chain, same floats.  With a tie the two passes may
 * break it differently (their tie keys use different point numberings), so the second pass is then really run.
 *   pn2x_furthest_point_sampling_radii : pn2_furthest_point_sampling that also records radii (b, m): radii[i] = the
 *             running-minimum distance of pick i when it was selected (the maximum of step i); radii[0] is unused.
 *   pn2x_fps_prefix_ties : given that run (xyz (b,n,3), picks idx1 (b,m1), radii (b,m1)), decide per cloud whether any
 *             arg-max of picks 1..m2-1 was tied.  With the picks known this has no dependency chain: every point
 *             replays its running minimum against the picks in order (same sqdist / min chain, same floats) and
 *             compares it with radii[i]; fully parallel (the prefix-min is also split over four threads per point), one workgroup
 *             per 64 points.
 *             flags: (b, pn2x_fps_prefix_flags(n)) ints, one per workgroup, non-zero = tie seen (conservative: a
 *             duplicate of a pick counts).
 *   pn2x_furthest_point_sampling_prefix : FPS over xyz (b, n, 3) -> idx (b, m), m <= n, really computed only for
 *             clouds with a non-zero flag; the others get idx = 0..m-1 (the launch returns at once).
 * They cover m2 <= 1024 and the register-resident sampling kernels (n <= 16384 points after padding).
 */
int pn2x_furthest_point_sampling_radii(int b, int n, int m, const float *xyz, int *idx, float *radii, void *stream);
/*
 * pn2x_fps_radii_knn: pn2x_furthest_point_sampling_radii AND pn2x_knn_indices (the nq query points per cloud `query` (b,nq,3) among
 * the same xyz; knn_idx (b,nq,k) sorted by (distance, index), knn_idx2 (b,nq,k2) its prefix when k2 > 0) in ONE launch: the first b
 * workgroups sample, the others search.  The sampling is a serial chain on one compute unit per cloud (pointnet_utils.py:368 ->
 * sampling_gpu.cu:94-209); the keypoints' neighbour lists (pointnet_utils.py:551-556) need nothing but the coordinates, so at small
 * batch (the tracking loop, track_network.py:159-217) they ride along instead of costing 21 us + a launch of their own.
 * Same results as the two calls.  PN2_ERANGE unless pn2x_fps_radii_knn_supported(n, nq, k): 512 < padded n <= 1024.
 */
int pn2x_fps_radii_knn_supported(int n, int nq, int k);
int pn2x_fps_radii_knn(int b, int n, int m, const float *xyz, int *idx, float *radii, int nq, int k, int k2, const float *query,
                       int *knn_idx, int *knn_idx2, void *stream);
int pn2x_fps_prefix_ties(int b, int n, int m1, int m2, const float *xyz, const int *idx1, const float *radii, int *flags, void *stream);
/*
 * pn2x_ball_query_picks_ties: pn2x_ball_query_picks2 (ball query around the picks (b, m) of a sampling run + their coordinates) AND
 * pn2x_fps_prefix_ties (was an arg-max among the first m2 picks tied?  radii (b, m) from the sampling run, flags as above) in ONE
 * launch -- both consume the picks and nothing of each other.  Same outputs as the two calls.  PN2_ERANGE unless
 * pn2x_ball_query_picks_ties_supported(b, n, m, m2): small batches (b n <= 16384, the tracking loop), m2 <= 1024.
 */
int pn2x_ball_query_picks_ties_supported(int b, int n, int m, int m2);
int pn2x_ball_query_picks_ties(int b, int n, int m, float radius, int nsample, const float *xyz, const int *picks, float *new_xyz,
                               int *idx, float *new_xyz_copy, int copy_ld, int m2, const float *radii, int *flags, void *stream);
int pn2x_fps_prefix_flags(int n);
int pn2x_furthest_point_sampling_prefix(int b, int n, int m, const float *xyz, const int *flags, int nflags, int *idx, void *stream);

/*
 * pn2_knn without the distance output, plus (k2 > 0) the first k2 indices of every list as a second contiguous
 * (b, n, k2) tensor: the k-NN list is sorted by (distance, index), so a smaller neighbourhood is its prefix -- the
 * reference searches twice (GivenCenterPoints scales 16 and 64, pointnet_utils.py:560-565).  m <= 2048.
 */
int pn2x_knn_indices(int b, int n, int m, int k, int k2, const float *unknown, const float *known, int *idx, int *idx2, void *stream);

/*
 * pn2_ball_query whose centroids are given as indices into the cloud itself (picks (b, m) int32, values in [0, n):
 * the output of FPS), which is how PointNet++ always calls it (pointnet_utils.py:379-382: sample, gather, query).
 * Also writes the centroids' coordinates new_xyz (b, m, 3) = xyz[picks], so the gather launch disappears.
 */
int pn2x_ball_query_picks(int b, int n, int m, float radius, int nsample, const float *xyz, const int *picks, float *new_xyz,
                          int *idx, void *stream);
/* ... and a second copy of the coordinates into three columns of a wider row buffer (row stride copy_ld floats; the
 * consumer's [feat | xyz] GEMM input, so the reference's torch.cat of pointnet_utils.py:493 is not needed), or NULL. */
int pn2x_ball_query_picks2(int b, int n, int m, float radius, int nsample, const float *xyz, const int *picks, float *new_xyz,
                           int *idx, float *new_xyz_copy, int copy_ld, void *stream);

/*
 * pn2_ball_query through a cell grid (csrc/ball_query_grid.hip): identical output for every input -- the same hit test on
 * the same pairs, emitted in ascending index order with the reference's first-hit padding (ball_query_gpu.cu:34-43) --
 * but only the points in the 3x3x3 cells around a centroid are tested, in blocks of consecutive indices with the
 * reference's early exit after the nsample-th hit.  For clouds of n >= 2048 points (PN2_ERANGE below that and for
 * radius <= 0: callers run pn2_ball_query).  scratch: pn2x_ball_query_grid_scratch_words(b, n) 4-byte words, 16-byte
 * aligned, contents irrelevant (the per-cloud grid is rebuilt by every call).  pn2_ball_query itself takes this path for
 * large problems when it can use library-owned scratch (i.e. outside stream capture).
 */
long pn2x_ball_query_grid_scratch_words(int b, int n);
int pn2x_ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                         void *scratch, long scratch_words, void *stream);

/*
 * Row gather on point-major data: out[b, j, :] = src[b, idx[b,j], :]  (src (b,n,c), idx (b,m), out (b,m,c)).
 * The point-major twin of pn2_gather_points (used for the FPS-selected centroid coordinates).
 */
int pn2x_gather_rows(int b, int n, int m, int c, const float *src, const int *idx, float *out, void *stream);

/*
 * In-place point-major epilogue: y[r, ch] = act(y[r, ch] + bias[r / rows_per_bias, ch]) for y (rows, c) with row
 * stride ldy.  rows_per_bias = rows gives an ordinary per-channel bias; rows_per_bias = points-per-cloud adds a
 * per-cloud vector (the broadcast global feature of the reference's fp3 layer, pointnet_utils.py:443-444).
 */
int pn2x_bias_act_pm(long rows, int c, float *y, int ldy, const float *bias, long rows_per_bias, int relu, void *stream);

/*
 * out[b, ch] = max over the r rows of point-major x (b, r, c): the max over N of the reference's group-all
 * set-abstraction layer (pointnet_utils.py:508, torch.max over the point axis).
 */
int pn2x_max_rows(int b, int r, int c, const float *x, float *out, void *stream);

/* n <= pn2x_copy_multi_max() device-to-device copies (dst[i] <- src[i], bytes[i] bytes, non-overlapping) as ONE launch: the batch
 * hand-over of a graph-captured training step (a few batch tensors + the geometry pack into the captured step's static buffers;
 * the reference's trainer.py:278-302 moves a batch with one .to(device) per tensor).  The table travels in the kernel arguments. */
int pn2x_copy_multi_max(void);
int pn2x_copy_multi(int n, void *const *dst, const void *const *src, const long *bytes, void *stream);

/*
 * out = LN2( LN1( x + y + bias ) ) over the last dimension of row-major (rows, c) data, c <= 1024:
 * torch.nn.functional.layer_norm semantics (biased variance, eps inside the square root), affine (g, b) each.
 * y, bias and the second LayerNorm (g2, b2 both NULL) are optional.  One launch for the reference's residual add +
 * LayerNorm runs in the attention-free transformer blocks (transformer.py:84-95 with attn=False) of the 21-token tail.
 */
int pn2x_add_layernorm(long rows, int c, const float *x, const float *y, const float *bias, const float *g1,
                       const float *b1, float eps1, const float *g2, const float *b2, float eps2, float *out, void *stream);

/*
 * Head of HandTrackNet (hand_network.py:141-147): delta = h W^T + bias (W (3, c): the last Conv1d), kp_hand = delta +
 * xyz1, kp_cam = (kp_hand R^T) * scale + t.  h (b*j, c) token-major; xyz1, kp_hand, kp_cam (b, j, 3); R (b,3,3); t (b,3,1).
 */
int pn2x_pose_head(int b, int j, int c, const float *h, const float *w, const float *bias, const float *xyz1,
                   const float *R, const float *t, float scale, float *kp_hand, float *kp_cam, void *stream);
/* ... with the per-cloud flags of pn2x_hand_frame3 (or NULL): flagged frames get NaN keypoints in both frames. */
int pn2x_pose_head2(int b, int j, int c, const float *h, const float *w, const float *bias, const float *xyz1,
                    const float *R, const float *t, float scale, float *kp_hand, float *kp_cam, const int *nonfinite,
                    void *stream);

/*
 * Both scales of a keypoint-query module (reference PointNetSetAbstractionMsg_GivenCenterPoints, pointnet_utils.py:536-590:
 * two kNN neighbourhood sizes, same layer widths) in ONE persistent launch: every compute unit stays busy and each
 * workgroup loads only its own scale's weights (see sa_fused.hip).  A problem = the per-scale arguments of
 * pn2x_sa_mlp_max.  Supported (pn2x_sa_mlp_max_pair_supported): widths 128-128-192, k = {16, 64}, both problems with the
 * same set of layer-1 operands (a1f + xyz, optionally + cadd); anything else returns PN2_ERANGE (launch them separately).
 */
typedef struct pn2x_sa_problem {
    int n, s, k;
    const float *a1f;
    int a1f_ld;
    const float *xyz, *cxyz, *wx, *b1, *cadd;
    int cadd_ld;
    const int *idx;
    const float *w2, *b2, *w3, *b3;
    float *out;
    long out_b;
    int out_s, out_c;
} pn2x_sa_problem;
/*
 * pn2x_mlp2_rows: out[r, :] = relu(W3 relu(W2 x[r, :c1] + W2e x[r, c1:c1+3] + b2) + b3) for `rows` point-major rows (x: ldx
 * floats apart; out: ldo floats apart, c3 written; BatchNorm folded into W / b by the caller) -- two consecutive
 * [Conv1d 1x1 + BN + ReLU] layers of a feature-propagation MLP (pointnet_utils.py:504-506) in ONE launch: the tile loop of
 * pn2x_sa_mlp_max without neighbourhoods, both weight matrices register-resident, the intermediate activation never
 * written to HBM.  w2 (c2, c1), w3 (c3, c2) row-major, 16-byte aligned.  w2e (c2, 3) or NULL: weights of three more input
 * columns stored right behind the c1 features (the coordinates of an [interpolated | xyz] row; needs ldx >= c1 + 4).
 * x and out 16-byte aligned, ldx and ldo multiples of 4 floats (rows are read and written in 16-byte segments), else PN2_EINVAL.
 * PN2_ERANGE unless pn2x_mlp2_rows_supported(c1, c2, c3).
 */
/*
 * Number of compute units the persistent grids of pn2x_sa_mlp_max / _pair / pn2x_mlp2_rows may occupy (0 = all, the default;
 * also PN2_SA_CUS in the environment).  A workgroup of these kernels fills its CU, so a caller that keeps several batches in
 * flight on different streams gains by leaving a few CUs to the other stream (bench.py: 240 of 256 with two in flight).
 */
int pn2x_sa_set_compute_units(int n);
int pn2x_mlp2_rows_supported(int c1, int c2, int c3);
int pn2x_mlp2_rows(long rows, int c1, int c2, int c3, const float *x, int ldx, const float *w2, const float *w2e, const float *b2,
                   const float *w3, const float *b3, float *out, int ldo, void *stream);
int pn2x_sa_mlp_max_pair_supported(int k0, int k1, int c1, int c2, int c3);
int pn2x_sa_mlp_max_pair(int b, int c1, int c2, int c3, const pn2x_sa_problem *p0, const pn2x_sa_problem *p1, void *stream);

/*
 * ---- training-mode building blocks on point-major activations (hotrack_amd/csrc/train_ops.hip) --------------------------
 * The reference trains every grouped MLP as Conv2d(1x1) + BatchNorm2d + ReLU on channel-major (B, C, S, K) tensors
 * (pointnet_utils.py:399-403, :460-462, :504-506, :577-581).  A 1x1 convolution is a GEMM over all R = B*S*K positions
 * (library GEMM on point-major rows); what sits between two GEMMs is one pair of streaming kernels per direction.
 * Rows are `c` floats wide (c % 4 == 0, c <= 1024) with a row stride `ld*` (multiple of 4), 16-byte aligned.
 *
 * pn2x_bn_stats: per-channel sum_r y[r,:] and sum_r y[r,:]^2 into `sums` -- pn2x_bn_sums_doubles(c) fp64 accumulators zeroed by
 * the caller (several interleaved copies of the 2c sums: workgroups spread their atomics over the copies).
 */
int pn2x_bn_sums_doubles(int c);
int pn2x_bn_stats(long rows, int c, const float *y, int ldy, double *sums, void *stream);
/*
 * pn2x_bn_relu_apply: batch statistics from `sums` (biased variance, eps inside the sqrt -- torch.nn.BatchNorm semantics),
 * h = relu?(gamma * (y - mean) * invstd + beta); writes save_mean / save_invstd (c floats each, for the backward) and,
 * when running_mean != NULL, the running-statistics update running = (1 - momentum) running + momentum batch (variance
 * unbiased) and num_batches_tracked += 1.  conv_bias (or NULL) is the bias of the preceding convolution: it cancels in
 * the normalisation, so y is computed without it and it is only added to the running mean.
 */
int pn2x_bn_relu_apply(long rows, int c, const float *y, int ldy, const double *sums, const float *gamma, const float *beta,
                       const float *conv_bias, float eps, float momentum, float *running_mean, float *running_var,
                       long long *num_batches_tracked, float *save_mean, float *save_invstd, float *h, int ldh, int relu,
                       void *stream);
/*
 * pn2x_bn_relu_bwd: with g = dh * [h > 0] (h recomputed from y), xhat = (y - mean) * invstd:
 *   dbeta = sum_r g, dgamma = sum_r g * xhat, dy = gamma * invstd * (g - dbeta / R - xhat * dgamma / R).
 * sums: pn2x_bn_sums_doubles(c) fp64 accumulators zeroed by the caller (two launches: reduce, apply).  dbias (or NULL): gradient of the bias of
 * the convolution in front of the BatchNorm -- sum_r dy, identically zero -- written as zeros.
 */
int pn2x_bn_relu_bwd(long rows, int c, const float *dh, int ldd, const float *y, int ldy, const float *mean,
                     const float *invstd, const float *gamma, const float *beta, int relu, double *sums, float *dy, int ldo,
                     float *dgamma, float *dbeta, float *dbias, void *stream);
/*
 * Layer 1 of a set-abstraction scale in TRAINING mode, pre-activation, without materialising the grouped input
 * [feat_j | xyz_j - c_s | centre_feat_s] (reference pointnet_utils.py:389-399, :566-577): the same linear split as
 * pn2x_sa_mlp_max (a1f / wx / cadd, each optional, see above), bias omitted because BatchNorm follows.
 *   out (b, s*k, c1) point-major; rel_out (b, s*k, 3) or NULL receives xyz_j - c_s (the backward's d(wx) operand).
 */
int pn2x_sa_layer1(int b, int n, int s, int k, int c1, const float *a1f, int a1f_ld, const float *xyz, const float *cxyz,
                   const float *wx, const float *cadd, int cadd_ld, const int *idx, float *out, float *rel_out, void *stream);
/* the same with the (c1, 3) weight block taken in place from a wider matrix: rows wx_ld floats apart (a column block of the
 * layer's [feature | xyz | centre] weight; no contiguous copy per step) */
int pn2x_sa_layer1_ld(int b, int n, int s, int k, int c1, const float *a1f, int a1f_ld, const float *xyz, const float *cxyz,
                      const float *wx, int wx_ld, const float *cadd, int cadd_ld, const int *idx, float *out, float *rel_out,
                      void *stream);
/* the same, also accumulating the BatchNorm statistics of `out` into sums (pn2x_bn_sums_doubles(c1) doubles, zeroed by the caller,
 * as pn2x_bn_stats would): the layer that follows is a train-mode BatchNorm (pointnet_utils.py:399-401), and its separate
 * statistics pass over the (b s k, c1) tensor -- one launch per scale and step -- is saved */
int pn2x_sa_layer1_stats(int b, int n, int s, int k, int c1, const float *a1f, int a1f_ld, const float *xyz, const float *cxyz,
                         const float *wx, int wx_ld, const float *cadd, int cadd_ld, const int *idx, float *out, float *rel_out,
                         double *sums, void *stream);
/* ... for the TWO neighbourhood sizes of one module (pointnet_utils.py:566-581: same clouds, coordinates and centres; per scale its
 * neighbour list, weight block, per-point / per-centre terms, output, relative coordinates and statistics) in ONE launch: alone, the
 * K = 16 scale is a 12 us launch for a quarter of the K = 64 scale's rows.  Same results as two pn2x_sa_layer1_stats calls. */
int pn2x_sa_layer1_stats_pair(int b, int n, int s, const float *xyz, const float *cxyz, int k_a, int c1_a, const float *a1f_a, int a1f_ld_a,
                              const float *wx_a, int wx_ld_a, const float *cadd_a, int cadd_ld_a, const int *idx_a, float *out_a,
                              float *rel_out_a, double *sums_a, int k_b, int c1_b, const float *a1f_b, int a1f_ld_b, const float *wx_b,
                              int wx_ld_b, const float *cadd_b, int cadd_ld_b, const int *idx_b, float *out_b, float *rel_out_b,
                              double *sums_b, void *stream);
/*
 * Transpose of pn2x_gather_rows (group_points_grad on point-major rows, reference group_points_gpu.cu:8-25):
 *   din[b, idx[b,j], :] += dout[b, j, :]      dout (b, m, ldo), idx (b, m) int32, din (b, n, ldi) accumulated into.
 */
int pn2x_scatter_add_rows(int b, int n, int m, int c, const float *dout, int ldo, const int *idx, float *din, int ldi,
                          void *stream);
/*
 * Transpose of pn2x_three_interpolate_pm (three_interpolate_grad on rows, reference interpolate_gpu.cu:192-214):
 *   dpoints[b, idx[b,j,t], :] += weight[b,j,t] * dout[b, j, :]   dout (b, n, ldo), dpoints (b, m, ldp) accumulated into.
 */
int pn2x_three_interpolate_pm_grad(int b, int c, int m, int n, const float *dout, int ldo, const int *idx,
                                   const float *weight, float *dpoints, int ldp, void *stream);

/*
 * Atomics-free form of the two row scatters above: invert the index list of every cloud once --
 *   pn2x_inverse_index: idx (b, l) with values in [0, n_dst)  ->  offsets (b, n_dst + 1), order (b, l): entries
 *   order[offsets[i] .. offsets[i+1]) are the positions e of idx with idx[e] == i   (n_dst <= 16127) --
 * then every target row SUMS its contributions (pn2x_rows_segment_sum):
 *   t == 1:  din[b, i, :] (+)= sum_{e in list(i)} dout[b, e, :]                      (transpose of pn2x_gather_rows)
 *   t == 3:  din[b, i, :] (+)= sum_{e in list(i)} weight[b, e] * dout[b, e / 3, :]   (transpose of pn2x_three_interpolate_pm;
 *            idx / weight are the (b, m_src, 3) tensors flattened)
 * accumulate == 0 overwrites (nothing to pre-zero), != 0 adds to din.  The order inside a list is unspecified.
 */
int pn2x_inverse_index(int b, int n_dst, int l, const int *idx, int *offsets, int *order, void *stream);
int pn2x_rows_segment_sum(int b, int n_dst, int m_src, int t, int c, const float *dout, int ldo, const int *offsets,
                          const int *order, const float *weight, float *din, int ldi, int accumulate, void *stream);
/* t = 1 for TWO index lists over the same destination rows (the two neighbourhood sizes of a module scatter into two column blocks of
 * one gradient) in one launch where both take the one-thread-per-(destination, quad) kernel; two calls otherwise.  Same results. */
int pn2x_rows_segment_sum_pair(int b, int n_dst, int m_a, int c_a, const float *dout_a, int ldo_a, const int *offsets_a, const int *order_a,
                               float *din_a, int ldi_a, int m_b, int c_b, const float *dout_b, int ldo_b, const int *offsets_b,
                               const int *order_b, float *din_b, int ldi_b, int accumulate, void *stream);

/*
 * Backward of group_points / gather_points (t = 1) and three_interpolate (t = 3) on the reference's channel-major layout with
 * CALLER-provided scratch (pn2x_scatter_cm_scratch_ints int32 elements): same results as pn2_group_points_grad /
 * pn2_gather_points_grad / pn2_three_interpolate_grad (accumulates into grad_points), but nothing is allocated inside, so
 * the call can be captured into a HIP graph.  (The reference-signature entries use a library-owned per-stream scratch and
 * fall back to their LDS-atomic kernels when that scratch would have to grow during a capture.)
 *   grad_out (b, c, m_src); idx (b, m_src) [t = 1] or (b, m_src, 3) with weight (b, m_src, 3) [t = 3]; grad_points (b, c, n_dst).
 * PN2_ERANGE: shape not covered (n_dst > 16127 or lists longer than 64 KiB of LDS) -- use the pn2_* entry.
 */
long pn2x_scatter_cm_scratch_ints(int t, int b, int n_dst, int m_src);
int pn2x_scatter_cm(int t, int b, int c, int n_dst, int m_src, const float *grad_out, const int *idx, const float *weight,
                    float *grad_points, int *scratch, long scratch_ints, void *stream);

/*
 * relu(BatchNorm_train(y)) followed by the max over the k rows of every group (the neighbourhood reduction that ends a
 * set-abstraction scale, reference pointnet_utils.py:403,581), without writing the (groups*k, c) activations:
 *   out (groups, c) = max_k relu(bn(y[g*k + kk, :]));  arg (groups, c) int32 = the first arg-max row kk.
 * Statistics / running-statistics semantics as pn2x_bn_relu_apply (sums from pn2x_bn_stats over all groups*k rows).
 * pn2x_bn_relu_max_bwd: backward through max + ReLU + BatchNorm given dout (groups, c): only the arg-max row of a group
 * receives dout (times the ReLU mask); dy (groups*k, ldo), dgamma, dbeta, dbias as pn2x_bn_relu_bwd.
 */
int pn2x_bn_relu_max(long groups, int k, int c, const float *y, int ldy, const double *sums, const float *gamma, const float *beta,
                     const float *conv_bias, float eps, float momentum, float *running_mean, float *running_var,
                     long long *num_batches_tracked, float *save_mean, float *save_invstd, float *out, int *arg, void *stream);
/* two problems (the two neighbourhood sizes of a module) in ONE launch where both take the few-groups kernel; two calls otherwise */
int pn2x_bn_relu_max_pair(long groups_a, int k_a, int c_a, const float *y_a, int ldy_a, const double *sums_a, const float *gamma_a,
                          const float *beta_a, const float *conv_bias_a, float eps_a, float momentum_a, float *running_mean_a,
                          float *running_var_a, long long *nbt_a, float *save_mean_a, float *save_invstd_a, float *out_a, int *arg_a,
                          long groups_b, int k_b, int c_b, const float *y_b, int ldy_b, const double *sums_b, const float *gamma_b,
                          const float *beta_b, const float *conv_bias_b, float eps_b, float momentum_b, float *running_mean_b,
                          float *running_var_b, long long *nbt_b, float *save_mean_b, float *save_invstd_b, float *out_b, int *arg_b,
                          void *stream);
/* the same with the output rows ldo floats apart (`out` = a column block of a wider buffer: the scales of a multi-scale module,
 * reference pointnet_utils.py:405-409 / :583-590 `torch.cat(new_points_list, dim=1)`, write the halves of one tensor) */
int pn2x_bn_relu_max_ld(long groups, int k, int c, const float *y, int ldy, const double *sums, const float *gamma, const float *beta,
                        const float *conv_bias, float eps, float momentum, float *running_mean, float *running_var,
                        long long *num_batches_tracked, float *save_mean, float *save_invstd, float *out, int ldo, int *arg, void *stream);
int pn2x_bn_relu_max_pair_ld(long groups_a, int k_a, int c_a, const float *y_a, int ldy_a, const double *sums_a, const float *gamma_a,
                             const float *beta_a, const float *conv_bias_a, float eps_a, float momentum_a, float *running_mean_a,
                             float *running_var_a, long long *nbt_a, float *save_mean_a, float *save_invstd_a, float *out_a, int ldo_a,
                             int *arg_a, long groups_b, int k_b, int c_b, const float *y_b, int ldy_b, const double *sums_b,
                             const float *gamma_b, const float *beta_b, const float *conv_bias_b, float eps_b, float momentum_b,
                             float *running_mean_b, float *running_var_b, long long *nbt_b, float *save_mean_b, float *save_invstd_b,
                             float *out_b, int ldo_b, int *arg_b, void *stream);
int pn2x_bn_relu_max_bwd(long groups, int k, int c, const float *dout, const int *arg, const float *y, int ldy, const float *mean,
                         const float *invstd, const float *gamma, const float *beta, double *sums, float *dy, int ldo, float *dgamma,
                         float *dbeta, float *dbias, void *stream);

/*
 * The two launches of pn2x_bn_relu_bwd / pn2x_bn_relu_max_bwd as separate entries (used by the fused training stacks below).
 * pn2x_bn_bwd_reduce: sums += [sum_r g, sum_r g xhat] with g = dh . [relu(bn(y)) > 0] (relu != 0) -- or, with arg != NULL,
 *   dh = d(max over k consecutive rows) (rows / k x c, row stride ldd, arg (rows / k, c) with the same row stride c) routed
 *   to the recorded arg-max row and ReLU-masked.
 * pn2x_bn_bwd_apply: dy = gamma invstd (g' - sum(g)/R - xhat sum(g xhat)/R) with g' = g . mask when relu != 0, g as given
 *   (already masked) when relu == 0; also writes dgamma, dbeta (and zeros to dbias) from the sums.
 */
int pn2x_bn_bwd_reduce(long rows, int c, const float *dh, int ldd, const int *arg, int k, const float *y, int ldy, const float *mean,
                       const float *invstd, const float *gamma, const float *beta, int relu, double *sums, void *stream);
/* ... and, for the max-routed form (arg != NULL), the routed + masked gradient itself written densely to g_out (rows x c, row
 * stride ldg) or NULL: the fused GEMMs of the layer below then read a plain gradient instead of routing it on every load. */
int pn2x_bn_bwd_reduce_g(long rows, int c, const float *dh, int ldd, const int *arg, int k, const float *y, int ldy, const float *mean,
                         const float *invstd, const float *gamma, const float *beta, int relu, double *sums, float *g_out, int ldg,
                         void *stream);
/* The same sums for a max-routed top layer from its arg-max rows only (groups x c gathered reads of y instead of rows x c):
 * dout (groups x c, row stride ldd: may be a column block of a wider gradient), arg (groups x c, row stride lda), y ((groups * k) x c).  The layer below then routes on load (pn2x_tg_bwd, gmode 2). */
int pn2x_bn_bwd_reduce_routed(long groups, int k, int c, const float *dout, int ldd, const int *arg, int lda, const float *y, int ldy,
                              const float *mean, const float *invstd, const float *gamma, const float *beta, double *sums, void *stream);
/* two problems of the same channel count in one launch (two launches when the counts differ); same results */
int pn2x_bn_bwd_reduce_routed_pair(long groups_a, int k_a, int c_a, const float *dout_a, int ldd_a, const int *arg_a, int lda_a,
                                   const float *y_a, int ldy_a, const float *mean_a, const float *invstd_a, const float *gamma_a,
                                   const float *beta_a, double *sums_a, long groups_b, int k_b, int c_b, const float *dout_b, int ldd_b,
                                   const int *arg_b, int lda_b, const float *y_b, int ldy_b, const float *mean_b, const float *invstd_b,
                                   const float *gamma_b, const float *beta_b, double *sums_b, void *stream);
/* out (c x 3) = dy^T rel: dy (rows x c, row stride ldy), rel (rows x 3) -- the gradient of the three xyz columns of a grouped
 * layer-1 weight.  c / 4 must divide 256; scratch of pn2x_rows_outer3_scratch_floats(rows, c) floats (per-workgroup partials). */
long pn2x_rows_outer3_scratch_floats(long rows, int c);
int pn2x_rows_outer3(long rows, int c, const float *dy, int ldy, const float *rel, float *out, float *scratch, long scratch_floats,
                     void *stream);
int pn2x_bn_bwd_apply(long rows, int c, const float *g, int ldg, const float *y, int ldy, const float *mean, const float *invstd,
                      const float *gamma, const float *beta, int relu, const double *sums, float *dy, int ldo, float *dgamma,
                      float *dbeta, float *dbias, void *stream);

/* pn2x_bn_bwd_apply for the first layer of a set-abstraction scale, also forming dwx (c x 3) = dy^T rel from the dy it has in
 * registers (rel (rows x 3): the relative coordinates pn2x_sa_layer1 wrote) -- the separate pn2x_rows_outer3 pass disappears.
 * c <= 256; scratch: pn2x_bn_bwd_apply_rel_scratch_floats(rows, c) floats. */
long pn2x_bn_bwd_apply_rel_scratch_floats(long rows, int c);
int pn2x_bn_bwd_apply_rel(long rows, int c, const float *g, int ldg, const float *y, int ldy, const float *mean, const float *invstd,
                          const float *gamma, const float *beta, int relu, const double *sums, float *dy, int ldo, float *dgamma,
                          float *dbeta, float *dbias, const float *rel, float *scratch, long scratch_floats, float *dwx, void *stream);
/* two problems (the two neighbourhood sizes of a module) in one launch each of the two kernels; same results as two calls */
int pn2x_bn_bwd_apply_rel_pair(long rows_a, int c_a, const float *g_a, int ldg_a, const float *y_a, int ldy_a, const float *mean_a,
                               const float *invstd_a, const float *gamma_a, const float *beta_a, int relu_a, const double *sums_a,
                               float *dy_a, int ldo_a, float *dgamma_a, float *dbeta_a, float *dbias_a, const float *rel_a,
                               float *scratch_a, long scratch_floats_a, float *dwx_a, long rows_b, int c_b, const float *g_b, int ldg_b,
                               const float *y_b, int ldy_b, const float *mean_b, const float *invstd_b, const float *gamma_b,
                               const float *beta_b, int relu_b, const double *sums_b, float *dy_b, int ldo_b, float *dgamma_b,
                               float *dbeta_b, float *dbias_b, const float *rel_b, float *scratch_b, long scratch_floats_b, float *dwx_b,
                               void *stream);

/*
 * Train-mode [Conv 1x1 + BatchNorm + ReLU] layers with the BatchNorm folded into fp32-MFMA GEMMs (csrc/train_gemm.hip;
 * reference composition: pointnet_utils.py:399-403, :460-462, :504-506, :577-581 as Conv2d/Conv1d + BatchNorm + ReLU modules).
 * Layer i >= 2 of a stack, point-major rows; `p` = layer i-1, `i` = layer i; sums_* = pn2x_bn_sums_doubles(c) fp64
 * accumulators (forward: [sum y | sum y^2], backward: [sum g | sum g xhat]) zeroed by the caller.
 *
 * pn2x_tg_fwd    y_i (rows x n) = relu(BN_p(x)) w^T with x = y_p (rows x k) and w (n x k); BN_p's batch statistics come from
 *                sums_in (filled by the producer of x); workgroup 0 writes save_mean / save_invstd of layer p and its
 *                running-statistics update (semantics of pn2x_bn_relu_apply); sums_out (or NULL) receives the statistics of y_i.
 * pn2x_tg_dgrad  g_p (rows x n) = (dY_i w) . [relu(BN_p(y_p)) > 0] and sums_bwd_p, where dY_i (rows x kd) =
 *                gamma_i invstd_i (g - sum(g)/R - xhat_i sum(g xhat_i)/R) is computed on load from
 *                  gmode 0: g = the stored, already masked g_i (rows x kd, row stride ldg);
 *                  gmode 1: g = dh_i . [relu(BN_i(y_i)) > 0] with dh_i stored (rows x kd);
 *                  gmode 2: dh = d(max over kmax consecutive rows) (rows / kmax x kd, row stride ldg; arg likewise) routed to
 *                           the arg-max row and masked;
 *                and sums_bwd_i (complete: from pn2x_bn_bwd_reduce for the top layer, from the dgrad of layer i+1 otherwise).
 * pn2x_tg_wgrad  dw (n x k) = dY_i^T relu(BN_p(y_p)) (both operands computed on load), reduced over row splits through
 *                `partial` (>= pn2x_tg_wgrad_partial_floats(rows, n, k) floats); also dgamma_i, dbeta_i (and zeros to dbias_i).
 * pn2x_tg_supported(c_in, c_out): c_in % 4 == 0, c_in <= 512, c_out a multiple of 32 (for wgrad also c_in % 32 == 0).
 */
int pn2x_tg_supported(int c_in, int c_out);
int pn2x_tg_fwd(long rows, int k, int n, const float *x, int ldx, const float *w, int ldw, float *y, int ldy, const double *sums_in,
                const float *gamma, const float *beta, const float *conv_bias, float eps, float momentum, float *running_mean,
                float *running_var, long long *num_batches_tracked, float *save_mean, float *save_invstd, double *sums_out,
                void *stream);
int pn2x_tg_dgrad(long rows, int kd, int n, int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi, int ldyi,
                  const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i, const double *sums_bwd_i,
                  const float *w, int ldw, const float *yp, int ldyp, const float *mean_p, const float *invstd_p,
                  const float *gamma_p, const float *beta_p, float *gp, int ldgp, double *sums_bwd_p, void *stream);
long pn2x_tg_wgrad_partial_floats(long rows, int n, int k);
int pn2x_tg_wgrad(long rows, int n, int k, int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi, int ldyi,
                  const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i, const double *sums_bwd_i,
                  const float *yp, int ldyp, const float *mean_p, const float *invstd_p, const float *gamma_p, const float *beta_p,
                  float *partial, long partial_floats, float *dw, float *dgamma, float *dbeta, float *dbias, void *stream);

/*
 * The loss / metric dictionary of HandTrackNet.compute_loss (reference hand_network.py:159-221, training mode without MANO
 * terms) in one launch, and its gradient with respect to the predicted hand-frame keypoints in another (csrc/kabsch.hip).
 * pred_hf / init_hf (b,3,21) channel-major hand-frame keypoints, gt_kp / pred_kp (b,21,3) camera frame, R (b,3,3) / t (b,3) /
 * scale: the canonical pose, palm (pb,6,3) with pb in {1,b}: the palm template.  out[9] =
 *   {hand_pred_kp_loss, hand_pred_r_loss, hand_pred_t_loss, hand_pred_kp_diff, hand_init_kp_diff, hand_init_r_diff,
 *    hand_init_t_diff, hand_pred_r_diff, hand_pred_t_diff};  saved (b, 87) floats for the backward.
 * backward: d_pred_hf (b,3,21) = d(grad3[0] out[0] + grad3[1] out[1] + grad3[2] out[2]) / d pred_hf, grad3 on the device.
 */
int pn2x_hand_losses(int b, int pb, const float *pred_hf, const float *init_hf, const float *gt_kp, const float *pred_kp,
                     const float *R, const float *t, float scale, const float *palm, float *out, float *saved, void *stream);
int pn2x_hand_losses_backward(int b, int pb, const float *pred_hf, float scale, const float *palm, const float *saved,
                              const float *grad3, float *d_pred_hf, void *stream);
/* ... with the caller's weighted total folded in: weights (9) -> out has ten entries, out[9] = sum_i weights[i] out[i]; the backward
 * takes dL/d out[0..2] (grad3, may be NULL) and / or dL/d out[9] (grad_total, one float on the device, may be NULL). */
int pn2x_hand_losses2(int b, int pb, const float *pred_hf, const float *init_hf, const float *gt_kp, const float *pred_kp, const float *R,
                      const float *t, float scale, const float *palm, float *out, float *saved, const float *weights, void *stream);
int pn2x_hand_losses_backward2(int b, int pb, const float *pred_hf, float scale, const float *palm, const float *saved, const float *grad3,
                               const float *grad_total, const float *weights, float *d_pred_hf, void *stream);

/*
 * The 21-token tail in TRAINING mode (csrc/tail_train.hip; reference transformer.py:65-67, hand_network.py:139-147 with
 * attn=False): every element-wise run between two GEMMs as one launch per direction.  Rows are tokens (rows x c, contiguous).
 *   pn2x_tail_ln_fwd   u = x + dropout(y + bias) (y, bias optional), out = LN_b(LN_a(u)) (LN_b optional: gb = bb = NULL);
 *                      stats (rows, 4) = mean_a, rstd_a, mean_b, rstd_b for the backward.  seed_dev != NULL: this launch
 *                      advances the device seed counter and writes the new value to seed_out (it must have no dropout itself);
 *                      otherwise seed_in is the per-forward seed the dropout mask is hashed from (site: which dropout).
 *   pn2x_tail_ln_bwd   dx (= du), dy (if y), and -- ATOMICALLY ADDED onto zero-initialised c-float buffers -- dga, dba, dgb,
 *                      dbb, dbias.  c <= 512.
 *   pn2x_tail_relu_drop_fwd / _bwd   h = dropout(relu(z + bias)); dz, dbias (atomically added).  c % 4 == 0.
 * Dropout: kept elements scaled by 1 / (1 - p); the mask is regenerated in the backward from (seed, site, index).
 */
int pn2x_tail_ln_fwd(long rows, int c, const float *x, const float *y, const float *bias, float p, int site, const long long *seed_in,
                     long long *seed_dev, long long *seed_out, const float *ga, const float *ba, float eps_a, const float *gb,
                     const float *bb, float eps_b, float *out, float *stats, void *stream);
int pn2x_tail_ln_bwd(long rows, int c, const float *x, const float *y, const float *bias, float p, int site, const long long *seed_in,
                     const float *ga, const float *ba, float eps_a, const float *gb, const float *bb, float eps_b, const float *stats,
                     const float *dout, float *dx, float *dy, float *dga, float *dba, float *dgb, float *dbb, float *dbias,
                     void *stream);
int pn2x_tail_relu_drop_fwd(long rows, int c, const float *z, const float *bias, float p, int site, const long long *seed_in,
                            float *out, void *stream);
int pn2x_tail_relu_drop_bwd(long rows, int c, const float *z, const float *bias, float p, int site, const long long *seed_in,
                            const float *dh, float *dz, float *dbias, void *stream);
/* The last layer of final_mlp + residual on the initial keypoints + de-canonicalisation, in training (hand_network.py:141-147):
 * h (b*j, c) rows, w (3, c), bias (3), xyz1 (b, 3, j), R (b, 3, 3), t (b, 3), scale (b; scale_stride 0: one value for all) ->
 * kp_hand (b, 3, j) = h w^T + bias + xyz1, kp_cam (b, j, 3) = scale R kp_hand + t.  Backward: g_hand (b, 3, j) and / or g_cam (b, j, 3)
 * -> dh (b*j, c); dw (3, c) and dbias (3) are ADDED to (zero-filled accumulators). */
int pn2x_tail_pose_head_fwd(int b, int j, int c, const float *h, const float *w, const float *bias, const float *xyz1, const float *R,
                            const float *t, const float *scale, int scale_stride, float *kp_hand, float *kp_cam, void *stream);
int pn2x_tail_pose_head_bwd(int b, int j, int c, const float *h, const float *w, const float *g_hand, const float *g_cam, const float *R,
                            const float *scale, int scale_stride, float *dh, float *dw, float *dbias, void *stream);

/*
 * One Adam step over n fp32 tensors (csrc/adam.hip): torch.optim.Adam semantics (L2 weight decay added to the gradient, bias
 * corrections from the per-tensor `step` counters, no amsgrad -- reference trainer.py:49-52), the arithmetic of torch's fused
 * kernel.  p / g / m / v / step: HOST arrays of n device pointers (m = exp_avg, v = exp_avg_sq, step = one fp32 counter per
 * tensor, advanced by this call); numel: host array of n element counts.  The tensor table travels in the kernel arguments.
 */
int pn2x_adam_multi(int n, void *const *p, const void *const *g, void *const *m, void *const *v, void *const *step,
                    const long *numel, double lr, double beta1, double beta2, double eps, double weight_decay, void *stream);
/* advance = 0: the counters are left alone (the caller keeps them in one contiguous buffer and advances them all with
 * pn2x_adam_advance AFTER every pn2x_adam_multi2 of the step: the update kernels read the old values) */
int pn2x_adam_multi2(int n, void *const *p, const void *const *g, void *const *m, void *const *v, void *const *step, const long *numel,
                     double lr, double beta1, double beta2, double eps, double weight_decay, int advance, void *stream);
int pn2x_adam_advance(float *steps, int n, void *stream);
/* pn2x_tg_wgrad with the reduction deferred: n_partials != NULL -> only the partial tiles are written (dw zeroed) and their count
 * returned; pn2x_tg_reduce_multi then sums the partial tiles of SEVERAL layers (host arrays of `count` entries) and emits their
 * dgamma / dbeta (/ zero dbias) in one launch -- the weight gradients are not needed before the optimiser step. */
int pn2x_tg_wgrad2(long rows, int n, int k, int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi, int ldyi,
                   const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i, const double *sums_bwd_i,
                   const float *yp, int ldyp, const float *mean_p, const float *invstd_p, const float *gamma_p, const float *beta_p,
                   float *partial, long partial_floats, float *dw, float *dgamma, float *dbeta, float *dbias, int *n_partials,
                   void *stream);
int pn2x_tg_reduce_multi(int count, const float *const *partial, const int *n_partials, const int *numel, float *const *dw,
                         const double *const *sums_bwd, const int *channels, float *const *dgamma, float *const *dbeta,
                         float *const *dbias, void *stream);
/* ... with entries that are column slices of a wider layer: sums_ld[j] = channels of the whole layer (NULL: = channels[j]) */
int pn2x_tg_reduce_multi2(int count, const float *const *partial, const int *n_partials, const int *numel, float *const *dw,
                          const double *const *sums_bwd, const int *channels, const int *sums_ld, float *const *dgamma,
                          float *const *dbeta, float *const *dbias, void *stream);
/* pn2x_tg_fwd with a different schedule for 64- / 128-channel inputs (csrc/train_fwd.hip: W_i resident in LDS, 64-row tiles whose
 * normalised operand is built once, one 32 x 32 output block per wave).  Same arguments and results (up to summation order). */
int pn2x_tg_fwd2_supported(int c_in, int c_out);
int pn2x_tg_fwd2(long rows, int k, int n, const float *x, int ldx, const float *w, int ldw, float *y, int ldy, const double *sums_in,
                 const float *gamma, const float *beta, const float *conv_bias, float eps, float momentum, float *running_mean,
                 float *running_var, long long *num_batches_tracked, float *save_mean, float *save_invstd, double *sums_out,
                 void *stream);

/* The whole backward of fused layer i in one kernel (csrc/train_bwd.hip): g_{i-1} (gp, with the ReLU mask and the
 * BatchNorm-backward sums of layer i-1, as pn2x_tg_dgrad) AND the weight-gradient partial tiles (as pn2x_tg_wgrad2 with
 * n_partials) from one pass over g_i, Y_i and Y_{i-1}.  gmode 0: g pre-masked (rows x n); gmode 1: g dense, ReLU-masked from Y_i on load; gmode 2: g = d(max over kmax
 * rows) ((rows / kmax) x n) routed on load through arg (same shape / stride) and ReLU-masked from Y_i.  n = channels of layer i, k = channels of layer
 * i-1; w (n x k).  pn2x_tg_bwd_supported(k, n): k in {32, 64, 128} and the instantiated n; pn2x_tg_bwd_partials = the number
 * of (n x k) partial tiles written (partial_floats >= that * n * k), to be summed by pn2x_tg_reduce_multi.  dw is zeroed. */
/* pn2x_tg_bwd_slice: one column slice [c0, c0 + n) of a layer with sums_ld channels (all layer-i pointers offset to the slice by the
 * caller).  A wider layer runs as consecutive slices: raw_out = 1 leaves the unmasked partial data gradient in gp, the next slice
 * passes it as g_add (may alias gp) and the last one applies the mask and accumulates the sums.  gmode 1: dense g, masked from Y_i. */
int pn2x_tg_bwd_slice(long rows, int n, int k, int gmode, const float *g, int ldg, const int *arg, int ldarg, int kmax, const float *yi, int ldyi,
                      const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i, const double *sums_bwd_i,
                      int sums_ld, const float *w, int ldw, const float *yp, int ldyp, const float *mean_p, const float *invstd_p,
                      const float *gamma_p, const float *beta_p, float *gp, int ldgp, double *sums_bwd_p, float *partial,
                      long partial_floats, float *dw, const float *g_add, int ldga, int raw_out, void *stream);
/* Two problems of the same layer shape (and gradient source) in ONE launch -- the two neighbourhood sizes of a keypoint-query
 * module: same widths, different weights and row counts (reference pointnet_utils.py:566-581 runs them as separate Conv2d stacks).
 * Arguments: those of pn2x_tg_bwd_slice / pn2x_tg_fwd2 twice (n, k and gmode must agree); the persistent workgroups are split in
 * proportion to the tile counts.  n_partials[2] receives the number of weight-gradient partial tiles each problem wrote. */
int pn2x_tg_bwd_pair_supported(int c_in, int c_out);
int pn2x_tg_fwd2_pair_supported(int c_in, int c_out);
int pn2x_tg_bwd_slice_pair(long rows0, int n0, int k0, int gmode0, const float *g0, int ldg0, const int *arg0, int ldarg0, int kmax0,
                           const float *yi0, int ldyi0, const float *mean_i0, const float *invstd_i0, const float *gamma_i0,
                           const float *beta_i0, const double *sums_bwd_i0, int sums_ld0, const float *w0, int ldw0, const float *yp0,
                           int ldyp0, const float *mean_p0, const float *invstd_p0, const float *gamma_p0, const float *beta_p0, float *gp0,
                           int ldgp0, double *sums_bwd_p0, float *partial0, long partial_floats0, float *dw0, const float *g_add0, int ldga0,
                           int raw_out0,
                           long rows1, int n1, int k1, int gmode1, const float *g1, int ldg1, const int *arg1, int ldarg1, int kmax1,
                           const float *yi1, int ldyi1, const float *mean_i1, const float *invstd_i1, const float *gamma_i1,
                           const float *beta_i1, const double *sums_bwd_i1, int sums_ld1, const float *w1, int ldw1, const float *yp1,
                           int ldyp1, const float *mean_p1, const float *invstd_p1, const float *gamma_p1, const float *beta_p1, float *gp1,
                           int ldgp1, double *sums_bwd_p1, float *partial1, long partial_floats1, float *dw1, const float *g_add1, int ldga1,
                           int raw_out1, int *n_partials, void *stream);
int pn2x_tg_fwd2_pair(long rows0, int k, int n, const float *x0, int ldx0, const float *w0, int ldw0, float *y0, int ldy0,
                      const double *sums_in0, const float *gamma0, const float *beta0, const float *conv_bias0, float eps0, float momentum0,
                      float *running_mean0, float *running_var0, long long *nbt0, float *save_mean0, float *save_invstd0, double *sums_out0,
                      long rows1, const float *x1, int ldx1, const float *w1, int ldw1, float *y1, int ldy1, const double *sums_in1,
                      const float *gamma1, const float *beta1, const float *conv_bias1, float eps1, float momentum1, float *running_mean1,
                      float *running_var1, long long *nbt1, float *save_mean1, float *save_invstd1, double *sums_out1, void *stream);
int pn2x_tg_bwd_supported(int c_in, int c_out);
int pn2x_tg_bwd_partials(long rows, int c_out, int c_in);
/* Round 5: for c_in in {64, 128} the kernel keeps W_i in registers as the B operand of v_mfma_f32_16x16x4_f32 (16-column slices
 * per wave; csrc/train_bwd.hip, tg_bwd2).  pn2x_tg_bwd_set_variant(0) selects the round-4 kernel (W_i streamed from L2 / LDS) for
 * every shape, 1 (default, or HOTRACK_TGB2) the new one where instantiated.  Process-wide: switch between whole backward passes
 * only (tests, A/B benches) -- the number of partial tiles a launch writes depends on it. */
int pn2x_tg_bwd_set_variant(int v2);
int pn2x_tg_bwd(long rows, int n, int k, int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi, int ldyi,
                const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i, const double *sums_bwd_i,
                const float *w, int ldw, const float *yp,
                int ldyp, const float *mean_p, const float *invstd_p, const float *gamma_p, const float *beta_p, float *gp, int ldgp,
                double *sums_bwd_p, float *partial, long partial_floats, float *dw, void *stream);

/* The weight gradients of `count` plain linear layers in one grouped launch (csrc/train_wgrad.hip):
 *   dw[p] (n[p] x k[p], row stride lddw[p]) = g[p]^T (rows[p] x n[p], row stride ldg[p]) . x[p] (rows[p] x k[p], row stride ldx[p])
 * -- what autograd computes as `grad_out.t() @ input` per torch.nn.functional.linear call (layer 1 of every stack over the
 * un-grouped points, the rearrange linears, the 21-token tail: reference pointnet_utils.py:399-403,460-462,504-506,577-581,
 * blocks.py:226-239, transformer.py:72-82).  Host arrays of `count` <= pn2x_wgrad_multi_max() entries; any n / k / strides
 * (16-byte loads where an operand allows them); dw[p] is overwritten (a column block of a wider weight when lddw[p] > k[p]).
 * Work is cut into (128 x 128 tile, row split) units of equal length over all problems; split problems go through `scratch`
 * (>= pn2x_wgrad_multi_scratch_floats floats, 16-byte aligned) and are summed in a fixed order by a second launch:
 * deterministic, no atomics.  PN2_ESCRATCH when the scratch is too small. */
int pn2x_wgrad_multi_max(void);
long pn2x_wgrad_multi_scratch_floats(int count, const int *rows, const int *n, const int *k);
int pn2x_wgrad_multi(int count, const float *const *g, const int *ldg, const float *const *x, const int *ldx, const int *rows,
                     const int *n, const int *k, float *const *dw, const int *lddw, float *scratch, long scratch_floats, void *stream);

/* y (m x n, row stride ldy) = act(x (m x k) . w^T (w: n x k) + bias (n | NULL)), relu != 0: ReLU -- a dense layer over FEW rows
 * (the B = 1 tracking loop: 21 ... 1024 rows; csrc/linear_small.hip): one workgroup per 32 x 32 output block, its four waves
 * splitting the reduction.  Same result as the BLAS library up to summation order. */
int pn2x_linear_small(int m, int k, int n, const float *x, int ldx, const float *w, int ldw, const float *bias, int relu, float *y,
                      int ldy, void *stream);

/* y (m x n) = act(LN2(LN1(xa + ya + ybias)) . w^T + bias): pn2x_add_layernorm followed by pn2x_linear_small as ONE launch for the
 * 21-token tail at small batch (reference transformer.py:65-67,72-82 with attn = False: norm -> linear1 -> ReLU -> linear2 ->
 * residual + norm; hand_network.py:139-141).  xa, ya (m x k, contiguous rows; ya / ybias may be NULL), LayerNorm 1 (g1, b1, eps1),
 * optional LayerNorm 2 (g2, b2, eps2; g2 NULL: none), k <= 384.  xout (m x k | NULL) receives the normalised rows (the next
 * residual's input).  The LayerNorm arithmetic is pn2x_add_layernorm's, the product pn2x_linear_small's: same bits as the pair. */
int pn2x_ln_linear_small(int m, int k, int n, const float *xa, const float *ya, const float *ybias, const float *g1, const float *b1,
                         float eps1, const float *g2, const float *b2, float eps2, float *xout, const float *w, int ldw,
                         const float *bias, int relu, float *y, int ldy, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PN2_EXT_H */
