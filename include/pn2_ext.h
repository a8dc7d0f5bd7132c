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

#ifdef __cplusplus
}
#endif
#endif /* PN2_EXT_H */
