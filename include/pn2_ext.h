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
 * Fused grouped MLP + max of one set-abstraction scale, eval mode (BatchNorm folded into the
 * 1x1 convolutions by the caller).  Replaces, for one (radius | kNN) scale, the reference's
 *   group(points, idx) / group(xyz, idx) - centre / cat / [Conv2d 1x1 + BN2d + ReLU] x3 / max over K
 * (pointnet_utils.py:389-403 for SA-MSG, :566-581 for GivenCenterPoints) without ever
 * materialising the (B, C, S, K) grouped tensors.
 *
 * Layer 1 is linear in its input [feat_j | xyz_j - c_s | centre_feat_s], so the caller splits it:
 *   a1  (b, n, c1)  per-point term     W1[:, feat|xyz] . [feat_j ; xyz_j]            (point-major)
 *   c1v (b, s, c1)  per-centroid term  bias1 - W1[:, xyz] . c_s + W1[:, centre] . centre_feat_s
 * and the kernel computes, for every centroid s and neighbour idx[b,s,k] (idx: (b, s, k) int32),
 *   h1 = relu(a1[idx] + c1v);  h2 = relu(w2 h1 + b2);  h3 = w3 h2 + b3;  out = relu(max_k h3)
 * w2 (c2, c1), w3 (c3, c2) row-major; out (b, c3, s).  fp32 throughout (matrix cores:
 * v_mfma_f32_16x16x4_f32, exact fp32).  Supported: k in {16,32,64} and (c1,c2,c3) in
 * {(32,32,64), (64,64,128), (128,128,192)} -- PN2_ERANGE otherwise (query with
 * pn2x_sa_mlp_max_supported; callers keep the unfused operator path for other shapes).
 */
int pn2x_sa_mlp_max(int b, int n, int s, int k, int c1, int c2, int c3, const float *a1, const float *c1v,
                    const int *idx, const float *w2, const float *b2, const float *w3, const float *b3,
                    float *out, void *stream);
int pn2x_sa_mlp_max_supported(int k, int c1, int c2, int c3);

/*
 * In-place y[b,c,n] = act(y[b,c,n] + bias[c]) (relu != 0 -> ReLU): the epilogue of an eval-mode,
 * BatchNorm-folded 1x1 convolution (reference: Conv1d + BatchNorm1d + ReLU of the feature-propagation
 * stacks, pointnet_utils.py:460-462) whose GEMM half is a library GEMM on the (b, c, n) tensor.
 */
int pn2x_bias_act(int b, int c, int n, float *y, const float *bias, int relu, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PN2_EXT_H */
