// kabsch.hip -- batched 3-D rigid alignment on the device (no host SVD hop).
// See include/pn2_ext.h: pn2x_kabsch.  Reference algorithm: hand_utils.py:42-66.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {

// Rigid fit y ~= R x + t of `num` point pairs; x, y given through accessors (pointer + stride in floats).
// Horn's quaternion form; the 4x4 symmetric eigenproblem by cyclic Jacobi.  The matrix and the accumulated
// rotations are fp64, but each rotation ANGLE is computed in fp32 (hardware divide / sqrt) and only its cosine is
// re-normalised in fp64 with Newton steps (multiplies only): every applied rotation is orthogonal to ~1e-16, an
// inexact angle merely leaves a tiny off-diagonal for the next sweep.  No fp64 divide / sqrt in the loop
// (they made the first version of this kernel 20 us; this one is ~4 us).
// FAST = false: the Jacobi sweep only (the loss kernel runs 1024 threads per workgroup: 128 registers per lane, the fast path's
// extra live values would spill)
template <bool FAST = true>
__device__ void kabsch_solve(int num, const float *x, const float *y, int ystride, double (&R)[3][3], double (&t)[3]) {
    double cx[3] = {0, 0, 0}, cy[3] = {0, 0, 0};
    for (int p = 0; p < num; ++p)
        for (int a = 0; a < 3; ++a) { cx[a] += x[3 * p + a]; cy[a] += y[(size_t)ystride * p + a]; }
    const double inv = 1.0 / num;
    for (int a = 0; a < 3; ++a) { cx[a] *= inv; cy[a] *= inv; }
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // S[a][b] = sum (x_a - cx_a)(y_b - cy_b)
    for (int p = 0; p < num; ++p)
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) S[a][c] += ((double)x[3 * p + a] - cx[a]) * ((double)y[(size_t)ystride * p + c] - cy[c]);

    double A[4][4];  // Horn 1987: rotation = eigenvector of the largest eigenvalue of N
    A[0][0] = S[0][0] + S[1][1] + S[2][2];
    A[0][1] = S[1][2] - S[2][1];
    A[0][2] = S[2][0] - S[0][2];
    A[0][3] = S[0][1] - S[1][0];
    A[1][1] = S[0][0] - S[1][1] - S[2][2];
    A[1][2] = S[0][1] + S[1][0];
    A[1][3] = S[2][0] + S[0][2];
    A[2][2] = -S[0][0] + S[1][1] - S[2][2];
    A[2][3] = S[1][2] + S[2][1];
    A[3][3] = -S[0][0] - S[1][1] + S[2][2];
    for (int r = 1; r < 4; ++r)
        for (int c = 0; c < r; ++c) A[r][c] = A[c][r];
    double q0 = 1.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    // Fast path: the largest eigenvalue as the largest root of the characteristic polynomial (Newton from an upper bound: the
    // polynomial is convex beyond its largest root, so the iteration descends monotonically onto it) and its eigenvector as a
    // column of adj(N - lambda I) -- about 300 flops against the ~30 dependent rotations of the Jacobi sweep below (~12 us of the
    // 18.6 us this launch takes at the head of every frame).  adj(N - lambda I) = prod_{k != max}(lambda_k - lambda) v v^T: when
    // the top eigenvalue is not well separated (near-degenerate fits) its columns lose digits, and the Jacobi sweep takes over.
    bool solved = false;
    if constexpr (FAST) {
        double gx = 0, gy = 0;
        for (int p = 0; p < num; ++p)
            for (int a = 0; a < 3; ++a) {
                const double dx = (double)x[3 * p + a] - cx[a], dy = (double)y[(size_t)ystride * p + a] - cy[a];
                gx += dx * dx;
                gy += dy * dy;
            }
        double s2 = 0;
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) s2 += S[a][c] * S[a][c];
        const double detS = S[0][0] * (S[1][1] * S[2][2] - S[1][2] * S[2][1]) - S[0][1] * (S[1][0] * S[2][2] - S[1][2] * S[2][0]) +
                            S[0][2] * (S[1][0] * S[2][1] - S[1][1] * S[2][0]);
        // 3x3 minors of a 4x4 matrix M: rows r0<r1<r2, columns c0<c1<c2
        auto det3 = [](const double (&M)[4][4], int r0, int r1, int r2, int c0, int c1, int c2) {
            return M[r0][c0] * (M[r1][c1] * M[r2][c2] - M[r1][c2] * M[r2][c1]) - M[r0][c1] * (M[r1][c0] * M[r2][c2] - M[r1][c2] * M[r2][c0]) +
                   M[r0][c2] * (M[r1][c0] * M[r2][c1] - M[r1][c1] * M[r2][c0]);
        };
        const double C2 = -2.0 * s2, C1 = -8.0 * detS;
        const double C0 = A[0][0] * det3(A, 1, 2, 3, 1, 2, 3) - A[0][1] * det3(A, 1, 2, 3, 0, 2, 3) + A[0][2] * det3(A, 1, 2, 3, 0, 1, 3) -
                          A[0][3] * det3(A, 1, 2, 3, 0, 1, 2);
        double lam = 0.5 * (gx + gy);  // >= the largest eigenvalue (Cauchy-Schwarz on the correlation)
        bool conv = false;
        for (int it = 0; it < 40 && lam > 0.0; ++it) {
            const double l2 = lam * lam;
            const double pv = (l2 + C2) * l2 + C1 * lam + C0, dp = (4.0 * l2 + 2.0 * C2) * lam + C1;
            if (!(dp > 0.0)) break;
            const double ln = lam - pv / dp;
            conv = fabs(ln - lam) <= 1e-15 * fabs(ln);
            lam = ln;
            if (conv) break;
        }
        if (conv) {
            double (&M)[4][4] = A;  // N - lambda I in place (the diagonal is put back for the Jacobi sweep if this path gives up)
            const double d0 = A[0][0], d1 = A[1][1], d2 = A[2][2], d3 = A[3][3];
            A[0][0] = d0 - lam; A[1][1] = d1 - lam; A[2][2] = d2 - lam; A[3][3] = d3 - lam;
            double best = -1.0, v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // column j of the adjugate = the cofactors of row j
                const int r0 = j == 0 ? 1 : 0, r1 = j <= 1 ? 2 : 1, r2 = j <= 2 ? 3 : 2;
                const double w0 = det3(M, r0, r1, r2, 1, 2, 3), w1 = -det3(M, r0, r1, r2, 0, 2, 3), w2 = det3(M, r0, r1, r2, 0, 1, 3),
                             w3 = -det3(M, r0, r1, r2, 0, 1, 2);
                const double sg = (j & 1) ? -1.0 : 1.0;
                const double n2 = w0 * w0 + w1 * w1 + w2 * w2 + w3 * w3;
                if (n2 > best) { best = n2; v[0] = sg * w0; v[1] = sg * w1; v[2] = sg * w2; v[3] = sg * w3; }
            }
            const double l3 = lam * lam * lam;
            if (best > 1e-12 * l3 * l3) {  // product of the three eigenvalue gaps >~ 1e-6 lambda^3: the column carries >= 9 digits
                q0 = v[0]; q1 = v[1]; q2 = v[2]; q3 = v[3];
                solved = true;
            }
            A[0][0] = d0; A[1][1] = d1; A[2][2] = d2; A[3][3] = d3;
        }
    }
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    double diag2 = 0;
    for (int p = 0; p < 4; ++p)
        for (int q = 0; q < 4; ++q) diag2 += A[p][q] * A[p][q];
    for (int sweep = 0; sweep < 16 && !solved; ++sweep) {
        double off = 0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off <= 1e-30 * diag2) break;  // converged to fp64 round-off relative to the matrix norm
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const float apq = (float)A[p][q];
                if (apq == 0.f) continue;
                const float theta = (float)(A[q][q] - A[p][p]) / (2.0f * apq);
                const float tf = (theta >= 0.f ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
                const double tt = (double)tf;
                const double n2 = tt * tt + 1.0;
                double c = (double)rsqrtf((float)n2);      // fp32 seed
                c = c * (1.5 - 0.5 * n2 * c * c);            // Newton steps for 1/sqrt(n2) in fp64
                c = c * (1.5 - 0.5 * n2 * c * c);
                const double sn = tt * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk;
                    A[q][k] = sn * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    if (!solved) {
        double best = A[0][0];
        q0 = V[0][0]; q1 = V[1][0]; q2 = V[2][0]; q3 = V[3][0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (A[k][k] > best) { best = A[k][k]; q0 = V[0][k]; q1 = V[1][k]; q2 = V[2][k]; q3 = V[3][k]; }
    }
    const double nq = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    q0 *= nq; q1 *= nq; q2 *= nq; q3 *= nq;
    R[0][0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3;
    R[0][1] = 2 * (q1 * q2 - q0 * q3);
    R[0][2] = 2 * (q1 * q3 + q0 * q2);
    R[1][0] = 2 * (q1 * q2 + q0 * q3);
    R[1][1] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3;
    R[1][2] = 2 * (q2 * q3 - q0 * q1);
    R[2][0] = 2 * (q1 * q3 - q0 * q2);
    R[2][1] = 2 * (q2 * q3 + q0 * q1);
    R[2][2] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
    for (int a = 0; a < 3; ++a) t[a] = cy[a] - (R[a][0] * cx[0] + R[a][1] * cx[1] + R[a][2] * cx[2]);
}

__global__ void __launch_bounds__(64)
kabsch_kernel(int b, int xb, int num, const float *__restrict__ x_all, const float *__restrict__ y_all,
              float *__restrict__ R_all, float *__restrict__ t_all) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= b) return;
    double R[3][3], t[3];
    kabsch_solve(num, x_all + (size_t)(xb == 1 ? 0 : i) * num * 3, y_all + (size_t)i * num * 3, 3, R, t);
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) R_all[(size_t)i * 9 + 3 * a + c] = (float)R[a][c];
        t_all[(size_t)i * 3 + a] = (float)t[a];
    }
}

// Backward of the fit (R, t) = Kabsch(x, y) with respect to y, in closed form (pn2_ext.h: pn2x_kabsch_backward).
// With w = sum_i (x_i - cx)(y_i - cy)^T, R w = Sym symmetric (R^T is the polar factor of w):
//     dL/dw = 2 R^T hat(u),  u = K^-1 axial((B - B^T) / 2),  K = tr(Sym) I - Sym,  B = R G^T,
// where G = dL/dR - (dL/dt) cx^T collects the rotation gradient (t = cy - R cx), and then
//     dL/dy_i = (x_i - cx)^T dL/dw + (dL/dt)^T / num        (sum_i (x_i - cx) = 0, so centring y adds nothing).
// The same derivative autograd takes through an SVD; ~90 element-wise torch launches on (B,3,3) tensors otherwise.
// one fit; y rows are ystride floats apart, dy likewise (dystride); gR (row-major 3x3) / gt may be null
__device__ void kabsch_backward_one(int num, const float *x, const float *y, int ystride, const float *Rf, const double *gR, const double *gtv,
                                    float *dy, int dystride, float dy_scale, bool accumulate) {
    double cx[3] = {0, 0, 0}, cy[3] = {0, 0, 0};
    for (int p = 0; p < num; ++p)
        for (int a = 0; a < 3; ++a) { cx[a] += x[3 * p + a]; cy[a] += y[(size_t)ystride * p + a]; }
    const double inv = 1.0 / num;
    for (int a = 0; a < 3; ++a) { cx[a] *= inv; cy[a] *= inv; }
    double w[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int p = 0; p < num; ++p)
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) w[a][c] += ((double)x[3 * p + a] - cx[a]) * ((double)y[(size_t)ystride * p + c] - cy[c]);
    double R[3][3], G[3][3], gt[3];
    for (int a = 0; a < 3; ++a) gt[a] = gtv ? gtv[a] : 0.0;
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
            R[a][c] = Rf[3 * a + c];
            G[a][c] = (gR ? gR[3 * a + c] : 0.0) - gt[a] * cx[c];
        }
    double sym[3][3], Bm[3][3];
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
            double s = 0, q = 0;
            for (int k = 0; k < 3; ++k) { s += R[a][k] * w[k][c]; q += R[a][k] * G[c][k]; }
            sym[a][c] = s;
            Bm[a][c] = q;
        }
    for (int a = 0; a < 3; ++a)
        for (int c = a + 1; c < 3; ++c) sym[a][c] = sym[c][a] = 0.5 * (sym[a][c] + sym[c][a]);
    const double tr = sym[0][0] + sym[1][1] + sym[2][2];
    double K[3][3];
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) K[a][c] = (a == c ? tr : 0.0) - sym[a][c];
    const double pv[3] = {0.5 * (Bm[2][1] - Bm[1][2]), 0.5 * (Bm[0][2] - Bm[2][0]), 0.5 * (Bm[1][0] - Bm[0][1])};
    // u = K^-1 p by the adjugate (K is symmetric positive definite away from the degenerate fits)
    const double A00 = K[1][1] * K[2][2] - K[1][2] * K[2][1], A01 = K[0][2] * K[2][1] - K[0][1] * K[2][2], A02 = K[0][1] * K[1][2] - K[0][2] * K[1][1];
    const double A10 = K[1][2] * K[2][0] - K[1][0] * K[2][2], A11 = K[0][0] * K[2][2] - K[0][2] * K[2][0], A12 = K[0][2] * K[1][0] - K[0][0] * K[1][2];
    const double A20 = K[1][0] * K[2][1] - K[1][1] * K[2][0], A21 = K[0][1] * K[2][0] - K[0][0] * K[2][1], A22 = K[0][0] * K[1][1] - K[0][1] * K[1][0];
    const double det = K[0][0] * A00 + K[0][1] * A10 + K[0][2] * A20;
    const double u0 = (A00 * pv[0] + A01 * pv[1] + A02 * pv[2]) / det, u1 = (A10 * pv[0] + A11 * pv[1] + A12 * pv[2]) / det,
                 u2 = (A20 * pv[0] + A21 * pv[1] + A22 * pv[2]) / det;
    const double hat[3][3] = {{0, -u2, u1}, {u2, 0, -u0}, {-u1, u0, 0}};
    double dW[3][3];
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += R[k][a] * hat[k][c];
            dW[a][c] = 2.0 * s;
        }
    for (int p = 0; p < num; ++p)
        for (int c = 0; c < 3; ++c) {
            double s = gt[c] * inv;
            for (int a = 0; a < 3; ++a) s += ((double)x[3 * p + a] - cx[a]) * dW[a][c];
            float *o = dy + (size_t)dystride * p + c;
            *o = (accumulate ? *o : 0.f) + dy_scale * (float)s;
        }
}

__global__ void __launch_bounds__(64)
kabsch_bwd_kernel(int b, int xb, int num, const float *__restrict__ x_all, const float *__restrict__ y_all,
                  const float *__restrict__ R_all, const float *__restrict__ gR_all, const float *__restrict__ gt_all,
                  float *__restrict__ dy_all) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= b) return;
    double gR[9], gt[3];
    for (int a = 0; a < 9; ++a) gR[a] = gR_all ? (double)gR_all[(size_t)i * 9 + a] : 0.0;
    for (int a = 0; a < 3; ++a) gt[a] = gt_all ? (double)gt_all[(size_t)i * 3 + a] : 0.0;
    kabsch_backward_one(num, x_all + (size_t)(xb == 1 ? 0 : i) * num * 3, y_all + (size_t)i * num * 3, 3, R_all + (size_t)i * 9, gR, gt,
                        dy_all + (size_t)i * num * 3, 3, 1.f, false);
}

// ---- the loss / metric dictionary of HandTrackNet.compute_loss (reference hand_network.py:159-221) in two launches ---------------
// Forward: per cloud b (two lanes: role 0 fits the ground-truth palm, role 1 the predicted palm)
//   gt_hf = R_c^T (gt - t_c) / s; pred_s = s pred_hf, gt_s = s gt_hf, init_s = s init_hf           (canonicalize, hand_utils.py:30-31)
//   out[0] hand_pred_kp_loss = mean |pred_s - gt_s|           out[1] hand_pred_r_loss = mean |R - R_gt|    out[2] hand_pred_t_loss = mean |t - t_gt|
//   out[3] hand_pred_kp_diff = mean_k ||pred_kp - gt_kp||      out[4] hand_init_kp_diff = mean_k ||init_s - gt_s||
//   out[5] hand_init_r_diff = mean angle(R_gt) [deg]           out[6] hand_init_t_diff = mean ||t_gt||
//   out[7] hand_pred_r_diff = mean angle(R^T R_gt) [deg]       out[8] hand_pred_t_diff = mean ||t - t_gt||
// with (R, t) = Kabsch(palm template -> palm keypoints of pred_s), (R_gt, t_gt) likewise of gt_s (hand_network.py:186-190).
// The torch composition of the same dictionary is ~75 launches of 4-5 us in a captured training step (forward + backward).
constexpr int kHlJ = 21, kHlPalm = 6;
__constant__ int kHlPalmIdx[kHlPalm] = {0, 1, 5, 9, 13, 17};  // hand_utils.handkp2palmkp

constexpr int kHlChunk = 128;  // clouds per pass of the single workgroup

__global__ void __launch_bounds__(512)
hand_loss_fwd_kernel(int B, int pb, const float *__restrict__ pred_hf, const float *__restrict__ init_hf, const float *__restrict__ gt_kp,
                     const float *__restrict__ pred_kp, const float *__restrict__ Rc, const float *__restrict__ tc, float s,
                     const float *__restrict__ palm, float *__restrict__ out, float *__restrict__ saved,
                     const float *__restrict__ weights) {
    // saved per cloud: [0:63) gt_s (3,21 channel-major) | [63:72) R | [72:75) t | [75:84) R_gt | [84:87) t_gt
    // One workgroup of 8 waves (512 threads: the closed-form rigid fit needs more than the 128 registers a 1024-thread workgroup
    // leaves a thread), clouds in passes of 128: (1) a wave per cloud with lanes 0..20 = the keypoints (all loads of a cloud
    // in flight together), palm points to LDS; (2) 2 x 128 threads each solve ONE rigid fit (ground truth / predicted) -- all fits of a
    // pass run side by side (closed form; the Jacobi sweep -- ~15 us -- only for ill-separated fits); (3) a thread per cloud forms the rotation / translation terms.
    __shared__ float acc[9];
    __shared__ float yl[kHlChunk][2][kHlPalm * 3];
    __shared__ float fit[kHlChunk][2][12];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 9) acc[threadIdx.x] = 0.f;
    float part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b0 = 0; b0 < B; b0 += kHlChunk) {
        const int nb = (B - b0) < kHlChunk ? (B - b0) : kHlChunk;
        __syncthreads();
        for (int i = w; i < nb; i += 8) {
            const int b = b0 + i;
            const float *Rb = Rc + 9 * (size_t)b, *tb = tc + 3 * (size_t)b;
            float *sv = saved + 87 * (size_t)b;
            float l1 = 0.f, dinit = 0.f, dpred = 0.f;
            if (lane < kHlJ) {
                const int k = lane;
                const float *g = gt_kp + ((size_t)b * kHlJ + k) * 3, *pk = pred_kp + ((size_t)b * kHlJ + k) * 3;
                const float d0 = g[0] - tb[0], d1 = g[1] - tb[1], d2 = g[2] - tb[2];
                float gs[3], ps[3], n2 = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gs[c] = ((d0 * Rb[c] + d1 * Rb[3 + c] + d2 * Rb[6 + c]) / s) * s;  // canonicalize (hand_utils.py:30-31), then * s
                    ps[c] = pred_hf[((size_t)b * 3 + c) * kHlJ + k] * s;
                    const float is = init_hf[((size_t)b * 3 + c) * kHlJ + k] * s;
                    sv[c * kHlJ + k] = gs[c];
                    l1 += fabsf(ps[c] - gs[c]);
                    n2 += (is - gs[c]) * (is - gs[c]);
                }
                dinit = sqrtf(n2);
                dpred = sqrtf((pk[0] - g[0]) * (pk[0] - g[0]) + (pk[1] - g[1]) * (pk[1] - g[1]) + (pk[2] - g[2]) * (pk[2] - g[2]));
#pragma unroll
                for (int j = 0; j < kHlPalm; ++j)
                    if (kHlPalmIdx[j] == k)
#pragma unroll
                        for (int c = 0; c < 3; ++c) { yl[i][0][3 * j + c] = gs[c]; yl[i][1][3 * j + c] = ps[c]; }
            }
            l1 = wave_sum_f32(l1); dinit = wave_sum_f32(dinit); dpred = wave_sum_f32(dpred);
            if (lane == 0) { part[0] += l1; part[3] += dpred; part[4] += dinit; }
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * nb) {  // thread 2i: ground-truth fit of cloud i, 2i + 1: predicted fit
            const int i = threadIdx.x >> 1, role = threadIdx.x & 1, b = b0 + i;
            float y[kHlPalm * 3];
            for (int e = 0; e < kHlPalm * 3; ++e) y[e] = yl[i][role][e];
            double R[3][3], t[3];
            kabsch_solve<true>(kHlPalm, palm + (size_t)(pb == 1 ? 0 : b) * kHlPalm * 3, y, 3, R, t);
            float *dst = saved + 87 * (size_t)b + (role == 0 ? 75 : 63);
            for (int a = 0; a < 3; ++a) {
                for (int c = 0; c < 3; ++c) dst[3 * a + c] = fit[i][role][3 * a + c] = (float)R[a][c];
                dst[9 + a] = fit[i][role][9 + a] = (float)t[a];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            const float *Rg = fit[threadIdx.x][0], *tg = Rg + 9, *Rp = fit[threadIdx.x][1], *tp = Rp + 9;
            float tr_gt = Rg[0] + Rg[4] + Rg[8], tr_rel = 0.f, tn = 0.f, dn = 0.f;
            for (int a = 0; a < 9; ++a) {
                part[1] += fabsf(Rp[a] - Rg[a]);
                tr_rel += Rp[a] * Rg[a];  // trace(R^T R_gt)
            }
            for (int a = 0; a < 3; ++a) {
                part[2] += fabsf(tp[a] - tg[a]);
                tn += tg[a] * tg[a];
                dn += (tp[a] - tg[a]) * (tp[a] - tg[a]);
            }
            const float k180 = 57.29577951308232f;
            part[5] += acosf(fminf(fmaxf((tr_gt - 1.f) * 0.5f, -1.f), 1.f)) * k180;
            part[6] += sqrtf(tn);
            part[7] += acosf(fminf(fmaxf((tr_rel - 1.f) * 0.5f, -1.f), 1.f)) * k180;
            part[8] += sqrtf(dn);
        }
    }
    __syncthreads();
    for (int i = 0; i < 9; ++i)
        if (part[i] != 0.f) atomicAdd(&acc[i], part[i]);
    __syncthreads();
    if (threadIdx.x < 9) {
        const float denom[9] = {(float)B * 63.f, (float)B * 9.f, (float)B * 3.f, (float)B * 21.f, (float)B * 21.f, (float)B, (float)B, (float)B, (float)B};
        const float v = acc[threadIdx.x] / denom[threadIdx.x];
        out[threadIdx.x] = v;
        if (weights) acc[threadIdx.x] = v * weights[threadIdx.x];
    }
    if (weights) {  // out[9] = sum_i weights[i] out[i]: the trainer's weighted total (trainer.py:157-165) without further launches
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (int i = 0; i < 9; ++i) tot += acc[i];
            out[9] = tot;
        }
    }
}

// d(sum_i w_i out[i], i < 3) / d pred_hf, w = (dL/d kp_loss, dL/d r_loss, dL/d t_loss) read from the device (grad (3,))
__global__ void __launch_bounds__(64)
hand_loss_bwd_kernel(int B, int pb, const float *__restrict__ pred_hf, float s, const float *__restrict__ palm,
                     const float *__restrict__ saved, const float *__restrict__ grad, float *__restrict__ d_pred_hf,
                     const float *__restrict__ grad_total, const float *__restrict__ weights) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float *sv = saved + 87 * (size_t)b;
    // dL/d out[i] = grad[i] (+ dL/d total * weights[i])
    const float gt_ = grad_total ? grad_total[0] : 0.f;
    const float g0 = (grad ? grad[0] : 0.f) + (grad_total ? gt_ * weights[0] : 0.f);
    const float g1 = (grad ? grad[1] : 0.f) + (grad_total ? gt_ * weights[1] : 0.f);
    const float g2 = (grad ? grad[2] : 0.f) + (grad_total ? gt_ * weights[2] : 0.f);
    const float wk = g0 / ((float)B * 63.f), wr = g1 / ((float)B * 9.f), wt = g2 / ((float)B * 3.f);
    float *d = d_pred_hf + (size_t)b * 63;
    auto sgn = [](float v) { return (float)((v > 0.f) - (v < 0.f)); };
    for (int e = 0; e < 63; ++e) d[e] = wk * sgn(pred_hf[(size_t)b * 63 + e] * s - sv[e]) * s;
    double gR[9], gt[3];
    for (int a = 0; a < 9; ++a) gR[a] = (double)(wr * sgn(sv[63 + a] - sv[75 + a]));
    for (int a = 0; a < 3; ++a) gt[a] = (double)(wt * sgn(sv[72 + a] - sv[84 + a]));
    float yl[kHlPalm * 3], dyl[kHlPalm * 3];
    for (int j = 0; j < kHlPalm; ++j)
        for (int c = 0; c < 3; ++c) yl[3 * j + c] = pred_hf[((size_t)b * 3 + c) * kHlJ + kHlPalmIdx[j]] * s;
    kabsch_backward_one(kHlPalm, palm + (size_t)(pb == 1 ? 0 : b) * kHlPalm * 3, yl, 3, sv + 63, gR, gt, dyl, 3, 1.f, false);
    for (int j = 0; j < kHlPalm; ++j)
        for (int c = 0; c < 3; ++c) d[c * kHlJ + kHlPalmIdx[j]] += dyl[3 * j + c] * s;
}

// pn2x_hand_frame: Kabsch on the palm keypoints + canonicalisation of the whole cloud, one workgroup per cloud.
// Replaces ransac_rt + canonicalize (reference hand_network.py:100,118-119; hand_utils.py:30-31,42-66): the CPU
// SVD hop, the cat/transpose and ~8 small torch kernels become one launch.
//   out rows:  xyz_out[b, i, :] = R^T (p_i - t) / scale   written point-major (row-vector form (p - t) R / scale)
__global__ void __launch_bounds__(256)
hand_frame_kernel(int xb, int num, int n, int j, const float *__restrict__ tmpl_all, const float *__restrict__ kp_all,
                  const int *__restrict__ palm_idx, const float *__restrict__ pts_all, float scale,
                  float *__restrict__ R_all, float *__restrict__ t_all, float *__restrict__ xyz2_all,
                  float *__restrict__ xyz1_all, float *__restrict__ xyz2_copy, int copy_ld, int *__restrict__ nonfinite) {
    __shared__ float sR[9], st[3];
    __shared__ float sy[16 * 3];
    const int b = blockIdx.x;
    const float *kp = kp_all + (size_t)b * j * 3;
    if (threadIdx.x < num * 3) sy[threadIdx.x] = kp[palm_idx[threadIdx.x / 3] * 3 + threadIdx.x % 3];
    __syncthreads();
    if (threadIdx.x == 0) {
        double R[3][3], t[3];
        kabsch_solve(num, tmpl_all + (size_t)(xb == 1 ? 0 : b) * num * 3, sy, 3, R, t);
        for (int a = 0; a < 3; ++a) {
            for (int c = 0; c < 3; ++c) {
                sR[3 * a + c] = (float)R[a][c];
                R_all[(size_t)b * 9 + 3 * a + c] = (float)R[a][c];
            }
            st[a] = (float)t[a];
            t_all[(size_t)b * 3 + a] = (float)t[a];
        }
    }
    __syncthreads();
    const float r00 = sR[0], r01 = sR[1], r02 = sR[2], r10 = sR[3], r11 = sR[4], r12 = sR[5], r20 = sR[6], r21 = sR[7], r22 = sR[8];
    const float t0 = st[0], t1 = st[1], t2 = st[2];
    const float *pts = pts_all + (size_t)b * n * 3;
    float *o2 = xyz2_all + (size_t)b * n * 3;
    float *o1 = xyz1_all + (size_t)b * j * 3;
    int bad = 0;
    for (int i = threadIdx.x; i < n + j; i += 256) {
        const float *p = i < n ? pts + 3 * i : kp + 3 * (i - n);
        float *o = i < n ? o2 + 3 * i : o1 + 3 * (i - n);
        const float d0 = p[0] - t0, d1 = p[1] - t1, d2 = p[2] - t2;
        // (R^T d)_c = sum_a R[a][c] d_a, accumulated in index order like the reference's matmul
        const float v0 = (d0 * r00 + d1 * r10 + d2 * r20) / scale;
        const float v1 = (d0 * r01 + d1 * r11 + d2 * r21) / scale;
        const float v2 = (d0 * r02 + d1 * r12 + d2 * r22) / scale;
        o[0] = v0; o[1] = v1; o[2] = v2;
        bad |= !(fabsf(v0 + v1 + v2) < __builtin_inff());  // NaN / Inf in the cloud, the keypoints or the fit
        if (xyz2_copy && i < n) {  // second copy straight into a consumer's row buffer (three columns of a wider row)
            float *c = xyz2_copy + ((size_t)b * n + i) * copy_ld;
            c[0] = v0; c[1] = v1; c[2] = v2;
        }
    }
    if (nonfinite) {  // per-cloud flag for pn2x_pose_head2: a frame with a non-finite input yields NaN keypoints (as the
        bad = __syncthreads_or(bad);  // reference, where the NaN spreads through sampling / grouping / the global max-pool)
        if (threadIdx.x == 0) nonfinite[b] = bad;
    }
}

}  // namespace pn2

extern "C" int pn2x_kabsch(int b, int xb, int num, const float *x, const float *y, float *R, float *t, void *stream) {
    if (b < 0 || num < 1 || !(xb == b || xb == 1)) return PN2_EINVAL;
    if (b == 0) return PN2_OK;
    if (!x || !y || !R || !t) return PN2_ENULL;
    hipLaunchKernelGGL(pn2::kabsch_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, b, xb, num, x, y, R, t);
    return pn2::check_launch();
}

extern "C" int pn2x_kabsch_backward(int b, int xb, int num, const float *x, const float *y, const float *R, const float *grad_R,
                                    const float *grad_t, float *grad_y, void *stream) {
    if (b < 0 || num < 1 || !(xb == b || xb == 1)) return PN2_EINVAL;
    if (b == 0) return PN2_OK;
    if (!x || !y || !R || !grad_y) return PN2_ENULL;
    hipLaunchKernelGGL(pn2::kabsch_bwd_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, b, xb, num, x, y, R, grad_R, grad_t, grad_y);
    return pn2::check_launch();
}

extern "C" int pn2x_hand_losses(int b, int pb, const float *pred_hf, const float *init_hf, const float *gt_kp, const float *pred_kp,
                                const float *R, const float *t, float scale, const float *palm, float *out, float *saved, void *stream) {
    return pn2x_hand_losses2(b, pb, pred_hf, init_hf, gt_kp, pred_kp, R, t, scale, palm, out, saved, nullptr, stream);
}

// weights (9) != NULL: out has TEN entries, out[9] = sum_i weights[i] out[i]
extern "C" int pn2x_hand_losses2(int b, int pb, const float *pred_hf, const float *init_hf, const float *gt_kp, const float *pred_kp,
                                 const float *R, const float *t, float scale, const float *palm, float *out, float *saved,
                                 const float *weights, void *stream) {
    if (b < 1 || !(pb == 1 || pb == b) || !(scale > 0.f)) return PN2_EINVAL;
    if (!pred_hf || !init_hf || !gt_kp || !pred_kp || !R || !t || !palm || !out || !saved) return PN2_ENULL;
    hipLaunchKernelGGL(pn2::hand_loss_fwd_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, b, pb, pred_hf, init_hf, gt_kp, pred_kp, R, t,
                       scale, palm, out, saved, weights);
    return pn2::check_launch();
}

extern "C" int pn2x_hand_losses_backward(int b, int pb, const float *pred_hf, float scale, const float *palm, const float *saved,
                                         const float *grad3, float *d_pred_hf, void *stream) {
    return pn2x_hand_losses_backward2(b, pb, pred_hf, scale, palm, saved, grad3, nullptr, nullptr, d_pred_hf, stream);
}

// dL/d out[i] = grad3[i] (grad3 may be NULL) + grad_total[0] * weights[i] (grad_total may be NULL; weights as given to the forward)
extern "C" int pn2x_hand_losses_backward2(int b, int pb, const float *pred_hf, float scale, const float *palm, const float *saved,
                                          const float *grad3, const float *grad_total, const float *weights, float *d_pred_hf,
                                          void *stream) {
    if (b < 1 || !(pb == 1 || pb == b) || !(scale > 0.f)) return PN2_EINVAL;
    if (!pred_hf || !palm || !saved || !d_pred_hf || (!grad3 && !grad_total) || (grad_total && !weights)) return PN2_ENULL;
    hipLaunchKernelGGL(pn2::hand_loss_bwd_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, b, pb, pred_hf, scale, palm, saved,
                       grad3, d_pred_hf, grad_total, weights);
    return pn2::check_launch();
}

extern "C" int pn2x_hand_frame3(int b, int xb, int num, int n, int j, const float *palm_template, const float *kp,
                                const int *palm_idx, const float *points, float scale, float *R, float *t, float *xyz2,
                                float *xyz1, float *xyz2_copy, int copy_ld, int *nonfinite, void *stream);
extern "C" int pn2x_hand_frame2(int b, int xb, int num, int n, int j, const float *palm_template, const float *kp,
                                const int *palm_idx, const float *points, float scale, float *R, float *t, float *xyz2,
                                float *xyz1, float *xyz2_copy, int copy_ld, void *stream);

extern "C" int pn2x_hand_frame(int b, int xb, int num, int n, int j, const float *palm_template, const float *kp,
                               const int *palm_idx, const float *points, float scale, float *R, float *t, float *xyz2,
                               float *xyz1, void *stream) {
    return pn2x_hand_frame2(b, xb, num, n, j, palm_template, kp, palm_idx, points, scale, R, t, xyz2, xyz1, nullptr, 0, stream);
}

extern "C" int pn2x_hand_frame2(int b, int xb, int num, int n, int j, const float *palm_template, const float *kp,
                                const int *palm_idx, const float *points, float scale, float *R, float *t, float *xyz2,
                                float *xyz1, float *xyz2_copy, int copy_ld, void *stream) {
    return pn2x_hand_frame3(b, xb, num, n, j, palm_template, kp, palm_idx, points, scale, R, t, xyz2, xyz1, xyz2_copy, copy_ld, nullptr, stream);
}

extern "C" int pn2x_hand_frame3(int b, int xb, int num, int n, int j, const float *palm_template, const float *kp,
                                const int *palm_idx, const float *points, float scale, float *R, float *t, float *xyz2,
                                float *xyz1, float *xyz2_copy, int copy_ld, int *nonfinite, void *stream) {
    if (xyz2_copy && copy_ld < 3) return PN2_EINVAL;
    if (b < 0 || num < 1 || num > 16 || n < 0 || j < 1 || !(xb == b || xb == 1) || !(scale > 0.f)) return PN2_EINVAL;
    if (b == 0) return PN2_OK;
    if (!palm_template || !kp || !palm_idx || !points || !R || !t || !xyz2 || !xyz1) return PN2_ENULL;
    hipLaunchKernelGGL(pn2::hand_frame_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, xb, num, n, j, palm_template, kp,
                       palm_idx, points, scale, R, t, xyz2, xyz1, xyz2_copy, copy_ld, nonfinite);
    return pn2::check_launch();
}
