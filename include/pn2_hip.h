/*
 * pn2_hip.h -- C ABI of libpn2_hip.so: the MI355X (gfx950) implementation of the PointNet++
 * operator stack behind HOTrack's HandTrackNet.
 *
 * This is the drop-in boundary.  Each entry point replaces one pybind function of the
 * reference module `pointnet2_cuda` (network/models/pointnet_lib/src/pointnet2_api.cpp:11-24)
 * and takes exactly the reference wrapper's integer arguments, in the reference's order, with
 * the at::Tensor arguments replaced by raw device pointers and the implicit
 * at::cuda::getCurrentCUDAStream() replaced by an explicit `stream` (a hipStream_t passed as
 * void*; NULL = the default stream).  No torch types, no C++ types.
 *
 * Common contract (same as the reference unless noted):
 *   - all float tensors are fp32, all index tensors int32, contiguous, batch-major;
 *   - the caller allocates every output; the library only borrows pointers for the duration
 *     of the enqueue and keeps no state (re-entrant, fork/spawn safe, nothing at load time).
 *     The entries whose reference signature has no scratch argument but whose kernels want one
 *     (the three *_grad entries, ball_query on large clouds) take a stream-ordered temporary from
 *     the HIP runtime (hipMallocAsync before, hipFreeAsync after the enqueue, both on `stream`)
 *     and fall back to scratch-free kernels while `stream` is being captured into a graph;
 *     pn2_ext.h has the same operations with caller-provided scratch (capture-safe);
 *   - calls are asynchronous on `stream`; nothing synchronises;
 *   - return value: PN2_OK (0) or a negative PN2_E* code.  The reference's int wrappers always
 *     returned 1 and exit(-1)ed the process on a launch failure
 *     (e.g. sampling_gpu.cu:39-43); this library never terminates the process.
 *   - arguments are validated (the reference validated nothing except in ball_query.cpp:10-17).
 */
#ifndef PN2_HIP_H
#define PN2_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PN2_OK 0
#define PN2_EINVAL (-1)   /* bad dimension / parameter                                   */
#define PN2_ENULL (-2)    /* required pointer is NULL                                    */
#define PN2_ERANGE (-3)   /* size outside what the kernels support (e.g. k > 200)        */
#define PN2_ESCRATCH (-4) /* this size needs the optional scratch buffer (FPS temp)      */
#define PN2_ELAUNCH (-5)  /* HIP reported a launch failure (see pn2_last_hip_error)      */

#define PN2_KNN_MAX_K 200 /* interpolate_gpu.cu:30-31: double best[200]; int besti[200]  */

/* ABI version (bumped on any signature change) and error text. */
int pn2_abi_version(void);
const char *pn2_strerror(int code);
/* hipError_t of the last failed launch on this thread (0 if none). */
int pn2_last_hip_error(void);

/*
 * furthest_point_sampling_wrapper(b, n, m, points, temp, idx)
 *   reference: sampling.cpp:38-49 -> sampling_gpu.cu:94-253
 * xyz (b,n,3) -> idx (b,m).  idx[.,0] = 0; iteration winner = max over
 * (min-dist desc, bitrev(k mod bs) asc, k asc), bs = largest power of two <= min(n,1024)
 * -- the reference's shared-memory tree reduction order (sampling_gpu.cu:86-91,143-203).
 * `temp` (b,n) is the reference's HBM scratch (pre-filled 1e10 by the caller,
 * pointnet2_utils.py:28).  On CDNA4 the running distances live in registers, so temp may be
 * NULL and is neither read nor written for n <= 65536 (up to 16384 points the coordinates are
 * register-resident too; from there to 65536 they are re-read from L2 every pick); larger clouds need it
 * (PN2_ESCRATCH otherwise) and it must then be pre-filled with 1e10 as in the reference.
 */
int pn2_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                void *stream);

/*
 * ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx)
 *   reference: ball_query.cpp:14-24 -> ball_query_gpu.cu:9-66
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample): first nsample points (ascending index)
 * with d2 < radius*radius (fp32, strict), padded with the first hit; all-zero row when no
 * point is inside.  Unlike the reference, idx need not be pre-zeroed
 * (pointnet2_utils.py:262): every element of every row is written.
 */
int pn2_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, void *stream);

/*
 * group_points_wrapper(b, c, n, npoints, nsample, points, idx, out)
 *   reference: group_points.cpp:25-37 -> group_points_gpu.cu:47-86
 * out[b,c,p,s] = points[b,c,idx[b,p,s]].  c == 0 is legal (no-op).
 */
int pn2_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int *idx, float *out, void *stream);

/*
 * group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points)
 *   reference: group_points.cpp:11-23 -> group_points_gpu.cu:8-45
 * grad_points[b,c,idx[b,p,s]] += grad_out[b,c,p,s]   (accumulates into the caller's buffer,
 * which the reference's Python pre-zeroes, pointnet2_utils.py:232).
 */
int pn2_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int *idx, float *grad_points, void *stream);

/*
 * gather_points_wrapper(b, c, n, npoints, points, idx, out)
 *   reference: sampling.cpp:11-22 -> sampling_gpu.cu:8-44.   out[b,c,j] = points[b,c,idx[b,j]]
 */
int pn2_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                      float *out, void *stream);

/*
 * gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points)
 *   reference: sampling.cpp:24-35 -> sampling_gpu.cu:46-83.  grad_points[b,c,idx[b,j]] += ...
 */
int pn2_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                           const int *idx, float *grad_points, void *stream);

/*
 * knn_wrapper(b, n, m, k, unknown, known, dist2, idx)
 *   reference: interpolate.cpp:26-36 -> interpolate_gpu.cu:9-79
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,k) SQUARED distances ascending, idx (b,n,k);
 * ties -> lower index first; 1 <= k <= 200; with m < k the tail is dist2 = +inf, idx = 0.
 */
int pn2_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2,
            int *idx, void *stream);

/*
 * three_nn_wrapper(b, n, m, unknown, known, dist2, idx)
 *   reference: interpolate.cpp:14-24 -> interpolate_gpu.cu:81-146
 * Three nearest known points per unknown point; SQUARED distances; same tie/tail rules.
 */
int pn2_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int *idx, void *stream);

/*
 * three_interpolate_wrapper(b, c, m, n, points, idx, weight, out)
 *   reference: interpolate.cpp:39-53 -> interpolate_gpu.cu:149-189
 * points (b,c,m), idx/weight (b,n,3) -> out[b,c,j] = sum_t weight[b,j,t]*points[b,c,idx[b,j,t]]
 */
int pn2_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                          const float *weight, float *out, void *stream);

/*
 * three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points)
 *   reference: interpolate.cpp:55-69 -> interpolate_gpu.cu:192-233
 * grad_points[b,c,idx[b,j,t]] += grad_out[b,c,j]*weight[b,j,t]   (accumulates)
 */
int pn2_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                               const float *weight, float *grad_points, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PN2_HIP_H */
