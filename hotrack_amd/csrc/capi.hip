// capi.hip -- extern "C" entry points of libpn2_hip.so (declared in include/pn2_hip.h).
// Argument validation + dispatch only; kernels live in the sibling .hip files.
#include <stdlib.h>

#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
static thread_local int g_last_hip_error = 0;
void set_last_hip_error(int e) { g_last_hip_error = e; }

int fps_dispatch(int b, int n, int m, const float *xyz, float *temp, int *idx, hipStream_t st, int force_threads,
                 const int *skip_flags, int nflags, float *radii);
int fps_tie_check(int b, int n, int m, int m1, const float *xyz, const int *idx, const float *radii, int *flags, hipStream_t st);
bool fps_knn_supported(int n, int nq, int k);
bool ball_tie_supported(long b, long n, long m, long m2);
int ball_tie_dispatch(int b, int n, int m, float radius, int nsample, const float *xyz, int *idx, const int *picks, float *new_xyz_out,
                      float *new_xyz_copy, int copy_ld, int m2, const float *radii, int *flags, hipStream_t st);
bool three_nn_interp_supported(long b, long n, long m, long c, long ldp, long ldo);
int three_nn_interp_dispatch(int b, int n, int m, int c, const float *unknown, const float *known, const float *points, int ldp,
                             float *out, int ldo, hipStream_t st);
int fps_knn_dispatch(int b, int n, int m, const float *xyz, int *idx, float *radii, int nq, int k, int k2, const float *query,
                     int *kidx, int *kidx2, hipStream_t st);
int ball_query_dispatch(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                        int *idx, hipStream_t st, const int *picks, float *new_xyz_out, float *new_xyz_copy, int copy_ld);
int three_nn_dispatch(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                      hipStream_t st, bool weights);
int interp_pm_dispatch(int b, int c, int m, int n, const float *points, int ldp, const int *idx, const float *weight,
                       float *out, int ldo, hipStream_t st);
int knn_dispatch(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2, int *idx,
                 hipStream_t st, int k2, int *idx2);
int group_fwd_dispatch(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx,
                       float *out, hipStream_t st);
int group_bwd_dispatch(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                       float *grad_points, hipStream_t st);
int interp_fwd_dispatch(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                        float *out, hipStream_t st);
int interp_bwd_dispatch(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                        float *grad_points, hipStream_t st);
}  // namespace pn2

using namespace pn2;

#define PN2_REQ(cond, code) \
    do {                    \
        if (!(cond)) return (code); \
    } while (0)

static inline bool fits_int(long v) { return v >= 0 && v <= 2147483647L; }

extern "C" {

int pn2_abi_version(void) { return 1; }

int pn2_last_hip_error(void) { return g_last_hip_error; }

const char *pn2_strerror(int code) {
    switch (code) {
        case PN2_OK: return "ok";
        case PN2_EINVAL: return "invalid dimension or parameter";
        case PN2_ENULL: return "required pointer is NULL";
        case PN2_ERANGE: return "size outside the supported range";
        case PN2_ESCRATCH: return "this size needs the scratch buffer (temp) to be provided";
        case PN2_ELAUNCH: return "HIP kernel launch failed";
        default: return "unknown error";
    }
}

int pn2_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx, void *stream) {
    PN2_REQ(b >= 0 && n >= 1 && m >= 0, PN2_EINVAL);
    if (b == 0 || m == 0) return PN2_OK;
    PN2_REQ(xyz && idx, PN2_ENULL);
    PN2_REQ(b <= 65535 * 32768L, PN2_ERANGE);
    PN2_REQ(fits_int((long)n * 3), PN2_ERANGE);
    int force = 0;
    if (const char *e = getenv("PN2_FPS_THREADS")) force = atoi(e);  // tuning/experiments only
    return fps_dispatch(b, n, m, xyz, temp, idx, (hipStream_t)stream, force, nullptr, 0, nullptr);
}

// ---- two-level sampling shortcut (include/pn2_ext.h) ---------------------------------------------------------------
int pn2x_furthest_point_sampling_radii(int b, int n, int m, const float *xyz, int *idx, float *radii, void *stream) {
    PN2_REQ(b >= 0 && n >= 1 && m >= 0, PN2_EINVAL);
    if (b == 0 || m == 0) return PN2_OK;
    PN2_REQ(xyz && idx && radii, PN2_ENULL);
    PN2_REQ(fits_int((long)n * 3), PN2_ERANGE);
    return fps_dispatch(b, n, m, xyz, nullptr, idx, (hipStream_t)stream, 0, nullptr, 0, radii);
}

int pn2x_fps_radii_knn_supported(int n, int nq, int k) { return fps_knn_supported(n, nq, k) ? 1 : 0; }

int pn2x_fps_radii_knn(int b, int n, int m, const float *xyz, int *idx, float *radii, int nq, int k, int k2, const float *query,
                       int *knn_idx, int *knn_idx2, void *stream) {
    PN2_REQ(b >= 0 && n >= 1 && m >= 1 && nq >= 1 && k >= 1 && k2 >= 0 && k2 <= k, PN2_EINVAL);
    if (b == 0) return PN2_OK;
    PN2_REQ(xyz && idx && radii && query && knn_idx && (k2 == 0 || knn_idx2), PN2_ENULL);
    PN2_REQ(m <= n && fits_int((long)n * 3) && fits_int((long)b * nq * k), PN2_ERANGE);
    return fps_knn_dispatch(b, n, m, xyz, idx, radii, nq, k, k2, query, knn_idx, k2 ? knn_idx2 : nullptr, (hipStream_t)stream);
}

int pn2x_fps_prefix_ties(int b, int n, int m1, int m2, const float *xyz, const int *idx1, const float *radii, int *flags, void *stream) {
    PN2_REQ(b >= 0 && n >= 1 && m1 >= 1 && m2 >= 1 && m2 <= m1, PN2_EINVAL);
    if (b == 0) return PN2_OK;
    PN2_REQ(xyz && idx1 && radii && flags, PN2_ENULL);
    PN2_REQ(fits_int((long)n * 3) && b <= 65535, PN2_ERANGE);
    return fps_tie_check(b, n, m2, m1, xyz, idx1, radii, flags, (hipStream_t)stream);
}

int pn2x_fps_prefix_flags(int n) { return n < 1 ? PN2_EINVAL : (n + 63) / 64; }  // one per tie-check workgroup (64 points)

int pn2x_furthest_point_sampling_prefix(int b, int n, int m, const float *xyz, const int *flags, int nflags, int *idx, void *stream) {
    PN2_REQ(b >= 0 && n >= 1 && m >= 0 && m <= n && nflags >= 1, PN2_EINVAL);
    if (b == 0 || m == 0) return PN2_OK;
    PN2_REQ(xyz && idx && flags, PN2_ENULL);
    PN2_REQ(fits_int((long)n * 3), PN2_ERANGE);
    return fps_dispatch(b, n, m, xyz, nullptr, idx, (hipStream_t)stream, 0, flags, nflags, nullptr);
}

int pn2_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                   int *idx, void *stream) {
    PN2_REQ(b >= 0 && n >= 1 && m >= 0 && nsample >= 1, PN2_EINVAL);
    PN2_REQ(radius == radius, PN2_EINVAL);  // NaN radius
    if (b == 0 || m == 0) return PN2_OK;
    PN2_REQ(new_xyz && xyz && idx, PN2_ENULL);
    PN2_REQ(b <= 65535, PN2_ERANGE);
    PN2_REQ(fits_int((long)n * 3) && fits_int((long)m * nsample), PN2_ERANGE);
    // large clouds with many centroids: the cell-grid search (identical output, ball_query_grid.hip) when a stream-ordered temporary is
    // available (not while the stream is being captured: callers that capture use pn2x_ball_query_grid with their own scratch)
    if (n >= 4096 && (long)m * n >= (1L << 22) && radius > 0.f) {
        const long words = pn2x_ball_query_grid_scratch_words(b, n);
        if (words > 0 && words < (1L << 30)) {
            StreamScratch own;  // stream-ordered temporary: released when this call returns, after the launches
            int *scratch = own.acquire((size_t)words + 4, (hipStream_t)stream);
            if (scratch) {
                unsigned *aligned = reinterpret_cast<unsigned *>(((uintptr_t)scratch + 15) & ~(uintptr_t)15);
                const int rc = ball_query_grid_dispatch(b, n, m, radius, nsample, new_xyz, xyz, idx, aligned, (size_t)words, (hipStream_t)stream);
                if (rc != PN2_ERANGE) return rc;
            }
        }
    }
    return ball_query_dispatch(b, n, m, radius, nsample, new_xyz, xyz, idx, (hipStream_t)stream, nullptr, nullptr, nullptr, 0);
}

int pn2x_ball_query_picks2(int b, int n, int m, float radius, int nsample, const float *xyz, const int *picks, float *new_xyz,
                           int *idx, float *new_xyz_copy, int copy_ld, void *stream);

int pn2x_ball_query_picks(int b, int n, int m, float radius, int nsample, const float *xyz, const int *picks, float *new_xyz,
                          int *idx, void *stream) {
    return pn2x_ball_query_picks2(b, n, m, radius, nsample, xyz, picks, new_xyz, idx, nullptr, 0, stream);
}

int pn2x_ball_query_picks2(int b, int n, int m, float radius, int nsample, const float *xyz, const int *picks, float *new_xyz,
                           int *idx, float *new_xyz_copy, int copy_ld, void *stream) {
    PN2_REQ(!new_xyz_copy || copy_ld >= 3, PN2_EINVAL);
    PN2_REQ(b >= 0 && n >= 1 && m >= 0 && nsample >= 1, PN2_EINVAL);
    PN2_REQ(radius == radius, PN2_EINVAL);
    if (b == 0 || m == 0) return PN2_OK;
    PN2_REQ(xyz && picks && new_xyz && idx, PN2_ENULL);
    PN2_REQ(b <= 65535, PN2_ERANGE);
    PN2_REQ(fits_int((long)n * 3) && fits_int((long)m * nsample), PN2_ERANGE);
    return ball_query_dispatch(b, n, m, radius, nsample, nullptr, xyz, idx, (hipStream_t)stream, picks, new_xyz, new_xyz_copy, copy_ld);
}

int pn2x_ball_query_picks_ties_supported(int b, int n, int m, int m2) { return ball_tie_supported(b, n, m, m2) ? 1 : 0; }

int pn2x_ball_query_picks_ties(int b, int n, int m, float radius, int nsample, const float *xyz, const int *picks, float *new_xyz,
                               int *idx, float *new_xyz_copy, int copy_ld, int m2, const float *radii, int *flags, void *stream) {
    PN2_REQ(!new_xyz_copy || copy_ld >= 3, PN2_EINVAL);
    PN2_REQ(b >= 0 && n >= 1 && m >= 1 && nsample >= 1 && m2 >= 1 && m2 <= m, PN2_EINVAL);
    PN2_REQ(radius == radius, PN2_EINVAL);
    if (b == 0) return PN2_OK;
    PN2_REQ(xyz && picks && new_xyz && idx && radii && flags, PN2_ENULL);
    PN2_REQ(b <= 65535 && fits_int((long)n * 3) && fits_int((long)m * nsample), PN2_ERANGE);
    return ball_tie_dispatch(b, n, m, radius, nsample, xyz, idx, picks, new_xyz, new_xyz_copy, copy_ld, m2, radii, flags, (hipStream_t)stream);
}

int pn2_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx,
                     float *out, void *stream) {
    PN2_REQ(b >= 0 && c >= 0 && n >= 1 && npoints >= 0 && nsample >= 0, PN2_EINVAL);
    if (b == 0 || c == 0 || npoints == 0 || nsample == 0) return PN2_OK;
    PN2_REQ(points && idx && out, PN2_ENULL);
    PN2_REQ(b <= 65535 && fits_int((long)npoints * nsample), PN2_ERANGE);
    return group_fwd_dispatch(b, c, n, npoints, nsample, points, idx, out, (hipStream_t)stream);
}

int pn2_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                          float *grad_points, void *stream) {
    PN2_REQ(b >= 0 && c >= 0 && n >= 1 && npoints >= 0 && nsample >= 0, PN2_EINVAL);
    if (b == 0 || c == 0 || npoints == 0 || nsample == 0) return PN2_OK;
    PN2_REQ(grad_out && idx && grad_points, PN2_ENULL);
    PN2_REQ(b <= 65535 && c <= 65535 * 16 && fits_int((long)npoints * nsample), PN2_ERANGE);
    return group_bwd_dispatch(b, c, n, npoints, nsample, grad_out, idx, grad_points, (hipStream_t)stream);
}

int pn2_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx, float *out,
                      void *stream) {
    return pn2_group_points(b, c, n, npoints, 1, points, idx, out, stream);
}

int pn2_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                           float *grad_points, void *stream) {
    return pn2_group_points_grad(b, c, n, npoints, 1, grad_out, idx, grad_points, stream);
}

int pn2_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2, int *idx,
            void *stream) {
    PN2_REQ(b >= 0 && n >= 0 && m >= 0, PN2_EINVAL);
    PN2_REQ(k >= 1, PN2_EINVAL);
    PN2_REQ(k <= PN2_KNN_MAX_K, PN2_ERANGE);
    if (b == 0 || n == 0) return PN2_OK;
    PN2_REQ(unknown && dist2 && idx && (known || m == 0), PN2_ENULL);
    PN2_REQ(b <= 65535 && fits_int((long)n * k) && fits_int((long)m * 3), PN2_ERANGE);
    return knn_dispatch(b, n, m, k, unknown, known, dist2, idx, (hipStream_t)stream, 0, nullptr);
}

int pn2x_knn_indices(int b, int n, int m, int k, int k2, const float *unknown, const float *known, int *idx, int *idx2, void *stream) {
    PN2_REQ(b >= 0 && n >= 0 && m >= 1 && k >= 1 && k2 >= 0 && k2 <= k, PN2_EINVAL);
    PN2_REQ(k <= PN2_KNN_MAX_K, PN2_ERANGE);
    if (b == 0 || n == 0) return PN2_OK;
    PN2_REQ(unknown && known && idx && (idx2 || k2 == 0), PN2_ENULL);
    PN2_REQ(b <= 65535 && fits_int((long)n * k) && fits_int((long)m * 3), PN2_ERANGE);
    return knn_dispatch(b, n, m, k, unknown, known, nullptr, idx, (hipStream_t)stream, k2, k2 > 0 ? idx2 : nullptr);
}

int pn2_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                 void *stream) {
    PN2_REQ(b >= 0 && n >= 0 && m >= 0, PN2_EINVAL);
    if (b == 0 || n == 0) return PN2_OK;
    PN2_REQ(unknown && dist2 && idx && (known || m == 0), PN2_ENULL);
    PN2_REQ(b <= 65535 && fits_int((long)n * 3) && fits_int((long)m * 3), PN2_ERANGE);
    return three_nn_dispatch(b, n, m, unknown, known, dist2, idx, (hipStream_t)stream, false);
}

int pn2_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                          float *out, void *stream) {
    PN2_REQ(b >= 0 && c >= 0 && m >= 1 && n >= 0, PN2_EINVAL);
    if (b == 0 || c == 0 || n == 0) return PN2_OK;
    PN2_REQ(points && idx && weight && out, PN2_ENULL);
    PN2_REQ(b <= 65535 && fits_int((long)n * 3), PN2_ERANGE);
    return interp_fwd_dispatch(b, c, m, n, points, idx, weight, out, (hipStream_t)stream);
}

int pn2_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                               const float *weight, float *grad_points, void *stream) {
    PN2_REQ(b >= 0 && c >= 0 && m >= 1 && n >= 0, PN2_EINVAL);
    if (b == 0 || c == 0 || n == 0) return PN2_OK;
    PN2_REQ(grad_out && idx && weight && grad_points, PN2_ENULL);
    PN2_REQ(b <= 65535 && c <= 65535 * 16 && fits_int((long)n * 3), PN2_ERANGE);
    return interp_bwd_dispatch(b, c, n, m, grad_out, idx, weight, grad_points, (hipStream_t)stream);
}

/* ---- MI355X-side extensions (include/pn2_ext.h) that reuse the operator kernels ---------------------- */
int pn2x_three_nn_weights(int b, int n, int m, const float *unknown, const float *known, float *weight, int *idx,
                          void *stream) {
    PN2_REQ(b >= 0 && n >= 0 && m >= 3, PN2_EINVAL);  // fewer than 3 known points give inf distances -> NaN weights
    if (b == 0 || n == 0) return PN2_OK;
    PN2_REQ(unknown && known && weight && idx, PN2_ENULL);
    PN2_REQ(b <= 65535 && fits_int((long)n * 3) && fits_int((long)m * 3), PN2_ERANGE);
    return three_nn_dispatch(b, n, m, unknown, known, weight, idx, (hipStream_t)stream, true);
}

int pn2x_three_nn_interpolate_pm_supported(int b, int n, int m, int c, int ldp, int ldo) {
    return three_nn_interp_supported(b, n, m, c, ldp, ldo) ? 1 : 0;
}

int pn2x_three_nn_interpolate_pm(int b, int n, int m, int c, const float *unknown, const float *known, const float *points, int ldp,
                                 float *out, int ldo, void *stream) {
    PN2_REQ(b >= 0 && n >= 0 && m >= 3 && c >= 1 && ldp >= c && ldo >= c, PN2_EINVAL);
    if (b == 0 || n == 0) return PN2_OK;
    PN2_REQ(unknown && known && points && out, PN2_ENULL);
    PN2_REQ(b <= 65535 && fits_int((long)n * 3) && fits_int((long)m * 3), PN2_ERANGE);
    return three_nn_interp_dispatch(b, n, m, c, unknown, known, points, ldp, out, ldo, (hipStream_t)stream);
}

int pn2x_three_interpolate_pm(int b, int c, int m, int n, const float *points, int ldp, const int *idx,
                              const float *weight, float *out, int ldo, void *stream) {
    PN2_REQ(b >= 0 && c >= 0 && m >= 1 && n >= 0 && ldp >= c && ldo >= c, PN2_EINVAL);
    if (b == 0 || c == 0 || n == 0) return PN2_OK;
    PN2_REQ(points && idx && weight && out, PN2_ENULL);
    PN2_REQ(fits_int((long)n * 3), PN2_ERANGE);
    return interp_pm_dispatch(b, c, m, n, points, ldp, idx, weight, out, ldo, (hipStream_t)stream);
}

}  // extern "C"
