// kabsch.hip -- batched 3-D rigid alignment on the device (no host SVD hop).
// See include/pn2_ext.h: pn2x_kabsch.  Reference algorithm: hand_utils.py:42-66.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {

__global__ void __launch_bounds__(64)
kabsch_kernel(int b, int xb, int num, const float *__restrict__ x_all, const float *__restrict__ y_all,
              float *__restrict__ R_all, float *__restrict__ t_all) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= b) return;
    const float *x = x_all + (size_t)(xb == 1 ? 0 : i) * num * 3;
    const float *y = y_all + (size_t)i * num * 3;
    double cx[3] = {0, 0, 0}, cy[3] = {0, 0, 0};
    for (int p = 0; p < num; ++p)
        for (int a = 0; a < 3; ++a) { cx[a] += x[3 * p + a]; cy[a] += y[3 * p + a]; }
    for (int a = 0; a < 3; ++a) { cx[a] /= num; cy[a] /= num; }
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // S[a][b] = sum (x_a - cx_a)(y_b - cy_b)
    for (int p = 0; p < num; ++p)
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) S[a][c] += ((double)x[3 * p + a] - cx[a]) * ((double)y[3 * p + c] - cy[c]);

    // Horn 1987: rotation = eigenvector of the largest eigenvalue of N
    double A[4][4];
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
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    double diag2 = 0;
    for (int p = 0; p < 4; ++p)
        for (int q = 0; q < 4; ++q) diag2 += A[p][q] * A[p][q];
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        // converged to fp64 round-off relative to the matrix norm (quadratic convergence: ~4-5 sweeps)
        if (off <= 1e-30 * diag2) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int best = 0;
    for (int k = 1; k < 4; ++k)
        if (A[k][k] > A[best][best]) best = k;
    double q0 = V[0][best], q1 = V[1][best], q2 = V[2][best], q3 = V[3][best];
    const double nq = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    q0 /= nq; q1 /= nq; q2 /= nq; q3 /= nq;
    double R[3][3];
    R[0][0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3;
    R[0][1] = 2 * (q1 * q2 - q0 * q3);
    R[0][2] = 2 * (q1 * q3 + q0 * q2);
    R[1][0] = 2 * (q1 * q2 + q0 * q3);
    R[1][1] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3;
    R[1][2] = 2 * (q2 * q3 - q0 * q1);
    R[2][0] = 2 * (q1 * q3 - q0 * q2);
    R[2][1] = 2 * (q2 * q3 + q0 * q1);
    R[2][2] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
    float *Ro = R_all + (size_t)i * 9;
    float *to = t_all + (size_t)i * 3;
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) Ro[3 * a + c] = (float)R[a][c];
        to[a] = (float)(cy[a] - (R[a][0] * cx[0] + R[a][1] * cx[1] + R[a][2] * cx[2]));
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
