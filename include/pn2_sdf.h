/*
 * pn2_sdf.h -- C ABI of the particle optimisers' signed-distance-volume lookups.
 *
 * The reference evaluates thousands of candidate poses ("particles") per tracking step by transforming the
 * observed cloud into each candidate's object frame and looking every point up in a voxelised SDF:
 *   - gf_optimize_obj.Distance / evaluate / optimize   network/models/optimization_obj.py:184-237, :244-301
 *     (2048 particles x N points, trilinear interpolation in a 201^3 fp16 volume, 10 iterations per frame)
 *   - gf_optimize_hand_pose.query_sdf / get_penetration_loss   network/models/optimization_hand.py:252-268
 *     (5120 particles x 778 MANO vertices, nearest-voxel read from a 151^3 fp16 volume)
 * as ~60 elementwise torch kernels over (P*N)-element temporaries per evaluation.  Each entry below is one
 * launch with no temporaries: transform, address arithmetic, gathers, interpolation and the per-particle
 * reduction are fused; the volume (16 MB at 201^3 fp16) stays resident in L2 / Infinity Cache.
 *
 * Conventions as pn2_hip.h: raw device pointers, contiguous row-major arrays, caller allocates everything,
 * asynchronous on `stream` (hipStream_t as void*, NULL = default stream), returns PN2_OK or a negative PN2_E*
 * code (pn2_strerror), nothing is retained after return.  Volumes: res^3 elements, element (ix,iy,iz) at
 * (ix*res + iy)*res + iz, IEEE binary16 (`vol_f16` = 1, the reference's storage type) or fp32 (`vol_f16` = 0, the
 * type the reference's volume has after `update_shape`, optimization_obj.py:400).  The trilinear entries take a
 * `vol_fmt` instead: 0 / 1 = that linear layout in fp32 / fp16, 2 / 3 = the CORNER layout built from it by
 * pn2s_build_corner_volume (fp32 / fp16), which they read ~3x faster.  Arithmetic is fp32 and follows
 * the reference's torch expressions operation for operation (true division, same association); the 3x3
 * transforms use the fixed chain  o_j = fma(q2, R2j, fma(q1, R1j, q0*R0j)).
 */
#ifndef PN2_SDF_H
#define PN2_SDF_H

#include "pn2_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Corner layout for the trilinear entries ("memory laid out for the lookup"): cell i (16 bytes in fp16, 32 in fp32)
 * holds the eight corner values d000..d111 that Distance() fetches when its base index i000 == i, gathered with the
 * reference's own index arithmetic and clamps (optimization_obj.py:206-222).  A lookup becomes ONE aligned 16-byte
 * load per point instead of eight scattered 2-byte reads in four cache lines, and is bit-identical to a lookup in
 * the linear volume for every input.  8x the memory (201^3 fp16: 130 MB of 288 GB), built once per object.
 *   out: pn2s_corner_volume_elems(res) elements of the volume's type (= 8 * res^3), 16-byte aligned.
 */
long pn2s_corner_volume_elems(int res);
int pn2s_build_corner_volume(const void *vol, int vol_f16, int res, void *out, void *stream);

/*
 * Distance(V)  (optimization_obj.py:184-228): trilinear SDF at m points.
 *   V (m,3) object-frame coordinates; out (m).  x = clamp((v - bbox_min)/stride, 0, res-1) per axis, the eight
 *   corner indices clamped to [0, res^3-1] exactly as the reference does, result clamped to [clamp_lo, clamp_hi]
 *   (reference constants: bbox_min -0.2, clamps -0.05 / 0.05).
 */
int pn2s_trilinear(int m, const float *V, const void *vol, int vol_fmt, int res, float bbox_min, float stride,
                   float clamp_lo, float clamp_hi, float *out, void *stream);

/*
 * evaluate(pcld, r, t)  (optimization_obj.py:230-237), fused: for each of p particles
 *   sdf_energy[i] = mean_j | Distance((pcld[j] - trans[i]) @ rot[i]) |
 * pcld (n,3) is shared by all particles; rot (p,3,3) row-major; trans (p,3).  The reference's `energy` is
 * 500 * sdf_energy (left to the caller).
 */
int pn2s_particle_energy(int p, int n, const float *pcld, const float *rot, const float *trans, const void *vol,
                         int vol_fmt, int res, float bbox_min, float stride, float clamp_lo, float clamp_hi,
                         float *sdf_energy, void *stream);

/*
 * The particle loop of gf_optimize_obj.optimize (optimization_obj.py:253-301, update_shape_flag False), entirely
 * on the device: `iterations` launches, each evaluating p candidate poses
 *   sample_i = [sqrt(1-|v|^2), v = pre_sampled[i,0:3]*search[0:3], pre_sampled[i,3:6]*search[3:6]]
 *   R_i = R @ quat2mat(sample_i[0:4]),  t_i = t + sample_i[4:7]
 * and then (last workgroup to finish) the weighted-mean pose update, SO(3) re-projection and search-size update,
 * with no host synchronisation (the reference syncs on `torch.any` every iteration).
 *   pcld (n,3); pre_sampled (p,6) with row 0 == 0 (particle 0 is the current pose);
 *   pose: in/out, 12 floats = rotation (3,3) row-major then translation (3);
 *   work: scratch, at least pn2s_obj_optimize_work_floats(p) floats, contents undefined on return
 *         except work[0..5] = final search size;
 *   c1, c2, beta: scaling_coefficient1 (0.02), scaling_coefficient2 (2), beta (0.9).
 */
int pn2s_obj_optimize(int p, int n, int iterations, const float *pcld, const float *pre_sampled, const void *vol,
                      int vol_fmt, int res, float bbox_min, float stride, float clamp_lo, float clamp_hi, float c1,
                      float c2, float beta, float *pose, float *work, void *stream);
int pn2s_obj_optimize_work_floats(int p);

/*
 * query_sdf(hand) (+ get_penetration_loss)  (optimization_hand.py:252-268): nearest-voxel read.
 *   hand (b,n,3); obj_r (3,3); obj_t (3):  q = (hand - obj_t) @ obj_r;
 *   index per axis = clamp(q // voxel_scale, -(res/2), res/2) + res/2   (`//` = torch's floor division);
 *   res must be odd (the reference asserts the index range; its volumes are 151^3 / 201^3).
 *   out_idx (b,n) int32 flat voxel index, or NULL;
 *   out_sdf (b,n) in the volume's element type (bit copy), or NULL;
 *   out_pen (b)   max_n |sdf| * (sdf < 0) in the volume's element type, or NULL.
 */
int pn2s_nearest(int b, int n, const float *hand, const float *obj_r, const float *obj_t, const void *vol,
                 int vol_f16, int res, float voxel_scale, int *out_idx, void *out_sdf, void *out_pen, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PN2_SDF_H */
