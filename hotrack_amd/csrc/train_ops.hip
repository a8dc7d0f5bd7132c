// train_ops.hip -- training-mode building blocks on POINT-MAJOR activations (rows x channels, fp32) for gfx950.
//
// The reference trains the grouped MLPs as Conv2d(1x1) + BatchNorm2d + ReLU over channel-major (B, C, S, K) tensors
// (network/models/pointnet_utils.py:399-403, :460-462, :504-506, :577-581) -- per layer a convolution-library call with
// layout transposes, a BatchNorm kernel, a ReLU kernel, and as many again in backward.  Here a 1x1 convolution is what it
// is, a GEMM over all R = B*S*K positions (library GEMM, point-major rows), and everything between two GEMMs is ONE pair
// of streaming kernels per direction:
//
//   forward    bn_stats        per-channel sum / sum of squares over the R rows (fp32 partials, fp64 atomics)
//              bn_relu_apply   H = relu(gamma * (Y - mean) * invstd + beta); the first workgroup also writes the saved
//                              mean / invstd, the running-statistics update and num_batches_tracked
//   backward   bn_relu_bwd_reduce   sum(g), sum(g * xhat)  with  g = dH * [H > 0]
//              bn_relu_bwd_apply    dY = gamma * invstd * (g - sum(g)/R - xhat * sum(g*xhat)/R); dgamma, dbeta
//
// plus the transposes of the point-major row gathers the forward uses (pn2x_gather_rows, pn2x_three_interpolate_pm):
//   scatter_add_rows      dIn[b, idx[b,j], :] += dOut[b, j, :]                   (group_points_grad on rows)
//   interp_pm_bwd         dPts[b, idx[b,j,t], :] += w[b,j,t] * dOut[b, j, :]      (three_interpolate_grad on rows)
// All HBM-bound: 16-byte accesses along the contiguous channel axis, one pass over each operand.
#include <cstdlib>
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {

constexpr int kTT = 256;
constexpr int kRep = kBnRep;  // the fp64 accumulators exist kRep times (pn2_common.h)

struct BnCh {  // per-thread constants of its 4 channels
    float mean[4], invstd[4], g[4], b[4];
};

__device__ __forceinline__ void bn_consts(BnCh &k, int C, int c0, long rows, const double *__restrict__ sums, float eps,
                                          const float *__restrict__ gamma, const float *__restrict__ beta, double *var_out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < kRep; ++r) {
            s1 += sums[(size_t)r * 2 * C + c0 + i];
            s2 += sums[(size_t)r * 2 * C + C + c0 + i];
        }
        const double m = s1 / (double)rows;
        double v = s2 / (double)rows - m * m;
        v = v > 0.0 ? v : 0.0;
        k.mean[i] = (float)m;
        k.invstd[i] = (float)(1.0 / sqrt(v + (double)eps));
        k.g[i] = gamma[c0 + i];
        k.b[i] = beta[c0 + i];
        if (var_out) var_out[i] = v;
    }
}

// ReLU that propagates NaN like torch's (fmaxf(NaN, 0) is 0): a diverged step must show up in the loss, not vanish in a max
__device__ __forceinline__ float relu_nan(float h) { return !(h <= 0.f) ? h : 0.f; }

__device__ __forceinline__ float bn_act(float y, const BnCh &k, int i) {
    return ((y - k.mean[i]) * k.invstd[i]) * k.g[i] + k.b[i];  // torch's evaluation order
}

// block-level reduction of per-thread float4 pairs over the threads that share a channel quad, then fp64 atomics
__device__ __forceinline__ void block_reduce_to_sums(float4 a, float4 b, int Q, int rpp, int C, double *__restrict__ sums) {
    __shared__ float4 red[2][kTT];
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if ((int)threadIdx.x < Q) {
        double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
        for (int r = 0; r < rpp; ++r) {
            const float4 u = red[0][r * Q + threadIdx.x], v = red[1][r * Q + threadIdx.x];
            s[0] += u.x; s[1] += u.y; s[2] += u.z; s[3] += u.w;
            t[0] += v.x; t[1] += v.y; t[2] += v.z; t[3] += v.w;
        }
        const int c0 = threadIdx.x * 4;
        double *dst = sums + (size_t)(blockIdx.x % kRep) * 2 * C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // hardware global_atomic_add_f64 (the plain atomicAdd(double) is a CAS loop)
            unsafeAtomicAdd(dst + c0 + i, s[i]);
            unsafeAtomicAdd(dst + C + c0 + i, t[i]);
        }
    }
}

__global__ void __launch_bounds__(kTT)
bn_stats_kernel(long rows, int C, const float *__restrict__ Y, int ld, int rows_per_block, double *__restrict__ sums) {
    const int Q = C >> 2, rpp = kTT / Q;
    const int q = threadIdx.x % Q, rr = threadIdx.x / Q;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float4 s = make_float4(0, 0, 0, 0), t = s;
    if (rr < rpp) {
        long r = r0 + rr;
        for (; r + 3L * rpp < r1; r += 4L * rpp) {  // four independent 16-byte loads in flight per thread
            const float *p = Y + r * ld + 4 * q;
            const float4 v0 = *reinterpret_cast<const float4 *>(p);
            const float4 v1 = *reinterpret_cast<const float4 *>(p + (long)rpp * ld);
            const float4 v2 = *reinterpret_cast<const float4 *>(p + 2L * rpp * ld);
            const float4 v3 = *reinterpret_cast<const float4 *>(p + 3L * rpp * ld);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
            t.x += (v0.x * v0.x + v1.x * v1.x) + (v2.x * v2.x + v3.x * v3.x);
            t.y += (v0.y * v0.y + v1.y * v1.y) + (v2.y * v2.y + v3.y * v3.y);
            t.z += (v0.z * v0.z + v1.z * v1.z) + (v2.z * v2.z + v3.z * v3.z);
            t.w += (v0.w * v0.w + v1.w * v1.w) + (v2.w * v2.w + v3.w * v3.w);
        }
        for (; r < r1; r += rpp) {
            const float4 v = *reinterpret_cast<const float4 *>(Y + r * ld + 4 * q);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            t.x += v.x * v.x; t.y += v.y * v.y; t.z += v.z * v.z; t.w += v.w * v.w;
        }
    }
    block_reduce_to_sums(s, t, Q, rpp, C, sums);
}

__global__ void __launch_bounds__(kTT)
bn_relu_apply_kernel(long rows, int C, const float *__restrict__ Y, int ldy, const double *__restrict__ sums,
                     const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ conv_bias,
                     float eps, float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                     long long *__restrict__ nbt, float *__restrict__ save_mean, float *__restrict__ save_invstd,
                     float *__restrict__ H, int ldh, int rows_per_block, int relu) {
    const int Q = C >> 2, rpp = kTT / Q;
    const int q = threadIdx.x % Q, rr = threadIdx.x / Q;
    if (rr >= rpp) return;
    BnCh k;
    double var[4];
    bn_consts(k, C, 4 * q, rows, sums, eps, gamma, beta, var);
    if (blockIdx.x == 0 && rr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * q + i;
            save_mean[c] = k.mean[i];
            save_invstd[c] = k.invstd[i];
            if (running_mean) {  // torch: running = (1 - m) * running + m * batch; the variance unbiased (n / (n - 1))
                const float bm = k.mean[i] + (conv_bias ? conv_bias[c] : 0.f);  // Y excludes the conv bias (it cancels in BN)
                const float bv = (float)(rows > 1 ? var[i] * ((double)rows / (double)(rows - 1)) : var[i]);
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * bm;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * bv;
            }
        }
        if (q == 0 && nbt) *nbt += 1;
    }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    auto act = [&](const float4 v) {
        float4 h;
        h.x = bn_act(v.x, k, 0); h.y = bn_act(v.y, k, 1); h.z = bn_act(v.z, k, 2); h.w = bn_act(v.w, k, 3);
        if (relu) { h.x = relu_nan(h.x); h.y = relu_nan(h.y); h.z = relu_nan(h.z); h.w = relu_nan(h.w); }
        return h;
    };
    long r = r0 + rr;
    for (; r + 3L * rpp < r1; r += 4L * rpp) {  // four independent 16-byte loads in flight per thread
        const float *p = Y + r * ldy + 4 * q;
        const float4 v0 = *reinterpret_cast<const float4 *>(p);
        const float4 v1 = *reinterpret_cast<const float4 *>(p + (long)rpp * ldy);
        const float4 v2 = *reinterpret_cast<const float4 *>(p + 2L * rpp * ldy);
        const float4 v3 = *reinterpret_cast<const float4 *>(p + 3L * rpp * ldy);
        float *o = H + r * ldh + 4 * q;
        *reinterpret_cast<float4 *>(o) = act(v0);
        *reinterpret_cast<float4 *>(o + (long)rpp * ldh) = act(v1);
        *reinterpret_cast<float4 *>(o + 2L * rpp * ldh) = act(v2);
        *reinterpret_cast<float4 *>(o + 3L * rpp * ldh) = act(v3);
    }
    for (; r < r1; r += rpp) *reinterpret_cast<float4 *>(H + r * ldh + 4 * q) = act(*reinterpret_cast<const float4 *>(Y + r * ldy + 4 * q));
}

// gradient of  out[g, c] = max_k h[g*K + k, c]  seen from row r = g*K + k: dout[g, c] where k is the recorded arg-max, else 0
__device__ __forceinline__ float4 max_grad(const float *__restrict__ dOut, int ldd, const int *__restrict__ arg, int C, int K, long r, int q) {
    const long grp = r / K;
    const int k = (int)(r - grp * K);
    const float4 d = *reinterpret_cast<const float4 *>(dOut + grp * ldd + 4 * q);
    const int4 a = *reinterpret_cast<const int4 *>(arg + grp * C + 4 * q);
    return make_float4(a.x == k ? d.x : 0.f, a.y == k ? d.y : 0.f, a.z == k ? d.z : 0.f, a.w == k ? d.w : 0.f);
}

// relu(BatchNorm(y)) followed by the max over the K rows of every group (the neighbourhood reduction of a set-abstraction
// scale, pointnet_utils.py:403,581): the (rows x C) activations are never written; arg records the first arg-max row.
__global__ void __launch_bounds__(kTT)
bn_relu_max_kernel(long groups, int K, int C, const float *__restrict__ Y, int ldy, const double *__restrict__ sums,
                   const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ conv_bias, float eps,
                   float momentum, float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ nbt,
                   float *__restrict__ save_mean, float *__restrict__ save_invstd, float *__restrict__ out, int ldo, int *__restrict__ arg) {
    const int Q = C >> 2;
    const long item = (long)blockIdx.x * kTT + threadIdx.x;
    const long rows = groups * K;
    const int q = (int)(item % Q);
    BnCh k;
    double var[4];
    bn_consts(k, C, 4 * q, rows, sums, eps, gamma, beta, var);
    if (item < Q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * q + i;
            save_mean[c] = k.mean[i];
            save_invstd[c] = k.invstd[i];
            if (running_mean) {
                const float bm = k.mean[i] + (conv_bias ? conv_bias[c] : 0.f);
                const float bv = (float)(rows > 1 ? var[i] * ((double)rows / (double)(rows - 1)) : var[i]);
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * bm;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * bv;
            }
        }
        if (q == 0 && nbt) *nbt += 1;
    }
    const long grp = item / Q;
    if (grp >= groups) return;
    const float *__restrict__ y = Y + grp * K * ldy + 4 * q;
    float m[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    int am[4] = {0, 0, 0, 0};
    auto take = [&](const float4 v, int kk) {
        const float h[4] = {bn_act(v.x, k, 0), bn_act(v.y, k, 1), bn_act(v.z, k, 2), bn_act(v.w, k, 3)};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (h[i] > m[i] || h[i] != h[i]) { m[i] = h[i]; am[i] = kk; }  // a NaN sticks (h > NaN is false afterwards), as torch's max
    };
    int kk = 0;
    for (; kk + 3 < K; kk += 4) {  // four rows in flight (the scan is one dependent-in-time load per row otherwise); same order
        const float4 v0 = *reinterpret_cast<const float4 *>(y + (long)kk * ldy);
        const float4 v1 = *reinterpret_cast<const float4 *>(y + (long)(kk + 1) * ldy);
        const float4 v2 = *reinterpret_cast<const float4 *>(y + (long)(kk + 2) * ldy);
        const float4 v3 = *reinterpret_cast<const float4 *>(y + (long)(kk + 3) * ldy);
        take(v0, kk); take(v1, kk + 1); take(v2, kk + 2); take(v3, kk + 3);
    }
    for (; kk < K; ++kk) take(*reinterpret_cast<const float4 *>(y + (long)kk * ldy), kk);
    *reinterpret_cast<float4 *>(out + grp * ldo + 4 * q) = make_float4(relu_nan(m[0]), relu_nan(m[1]), relu_nan(m[2]), relu_nan(m[3]));
    *reinterpret_cast<int4 *>(arg + grp * C + 4 * q) = make_int4(am[0], am[1], am[2], am[3]);
}

// The same reduction with the K rows of a group split over S lanes (S a power of two, 4 <= S <= 64): few groups x many rows per group
// (sa3: 32 groups x 128 rows, the keypoint queries: 672 x 64) leave the one-thread-per-(group, quad) kernel with a few thousand
// threads walking their rows one load after the other.  A workgroup takes 256 / S consecutive (group, quad) items; lane ks of an
// item scans rows ks, ks + S, ...; the partial (max, first arg-max) pairs meet in LDS.  Same result as the serial scan: the
// maximum, its FIRST row among equals, and a NaN sticks (with the last NaN row, as the serial update leaves it).  The channel
// constants are computed once per workgroup (the serial kernel derives them per thread from the fp64 sums).
struct MaxSplitArgs {
    long groups; int K, C, S; const float *Y; int ldy; const double *sums; const float *gamma, *beta, *conv_bias; float eps, momentum;
    float *running_mean, *running_var; long long *nbt; float *save_mean, *save_invstd, *out; int *arg; int ldo;  // out rows ldo floats apart (a column block of a wider buffer)
};
__device__ __forceinline__ void bn_relu_max_split_body(const MaxSplitArgs &A, unsigned bx) {
    const long groups = A.groups;
    const int K = A.K, C = A.C, S = A.S, ldy = A.ldy;
    const float *__restrict__ Y = A.Y, *__restrict__ gamma = A.gamma, *__restrict__ beta = A.beta, *__restrict__ conv_bias = A.conv_bias;
    const double *__restrict__ sums = A.sums;
    const float eps = A.eps, momentum = A.momentum;
    float *__restrict__ running_mean = A.running_mean, *__restrict__ running_var = A.running_var;
    long long *__restrict__ nbt = A.nbt;
    float *__restrict__ save_mean = A.save_mean, *__restrict__ save_invstd = A.save_invstd, *__restrict__ out = A.out;
    int *__restrict__ arg = A.arg;
    __shared__ float cst[4][kTT];        // mean, invstd, gamma, beta of the channels this workgroup touches (<= 256: 64 quads)
    __shared__ float pm[kTT][4];
    __shared__ int pa[kTT][4];
    const int Q = C >> 2;
    const int ipb = kTT / S;             // items per workgroup
    const long rows = groups * K;
    const long item0 = (long)bx * ipb;
    // channels of this workgroup: items item0 .. item0 + ipb - 1 -> quads (item % Q): ipb >= Q covers all C channels (C <= 256),
    // else the ipb consecutive quads starting at item0 % Q (wrapping)
    const int nq = ipb < Q ? ipb : Q;
    const int q0 = ipb < Q ? (int)(item0 % Q) : 0;
    for (int t = threadIdx.x; t < 4 * nq; t += kTT) {
        const int c = (4 * q0 + t) % C;
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < kRep; ++r) {
            s1 += sums[(size_t)r * 2 * C + c];
            s2 += sums[(size_t)r * 2 * C + C + c];
        }
        const double m = s1 / (double)rows;
        double v = s2 / (double)rows - m * m;
        v = v > 0.0 ? v : 0.0;
        const float mean = (float)m, invstd = (float)(1.0 / sqrt(v + (double)eps));
        cst[0][t] = mean; cst[1][t] = invstd; cst[2][t] = gamma[c]; cst[3][t] = beta[c];
    }
    if (bx == 0) {  // the consumer finalises the producer's statistics: saved mean / invstd, running estimates
        for (int c = threadIdx.x; c < C; c += kTT) {
            double s1 = 0.0, s2 = 0.0;
            for (int r = 0; r < kRep; ++r) {
                s1 += sums[(size_t)r * 2 * C + c];
                s2 += sums[(size_t)r * 2 * C + C + c];
            }
            const double m = s1 / (double)rows;
            double v = s2 / (double)rows - m * m;
            v = v > 0.0 ? v : 0.0;
            const float mean = (float)m;
            save_mean[c] = mean;
            save_invstd[c] = (float)(1.0 / sqrt(v + (double)eps));
            if (running_mean) {
                const float bm = mean + (conv_bias ? conv_bias[c] : 0.f);
                const float bv = (float)(rows > 1 ? v * ((double)rows / (double)(rows - 1)) : v);
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * bm;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * bv;
            }
        }
        if (threadIdx.x == 0 && nbt) *nbt += 1;
    }
    __syncthreads();
    const int il = threadIdx.x % ipb, ks = threadIdx.x / ipb;
    const long item = item0 + il;
    const bool live = item < groups * Q;
    const int q = (int)(item % Q);
    const long grp = item / Q;
    float m[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    int am[4] = {0, 0, 0, 0};
    if (live) {
        const int t0 = ipb < Q ? 4 * ((q - q0 + Q) % Q) : 4 * q;
        BnCh k;
#pragma unroll
        for (int i = 0; i < 4; ++i) { k.mean[i] = cst[0][t0 + i]; k.invstd[i] = cst[1][t0 + i]; k.g[i] = cst[2][t0 + i]; k.b[i] = cst[3][t0 + i]; }
        const float *__restrict__ y = Y + grp * K * ldy + 4 * q;
        auto take = [&](const float4 v, int kk) {
            const float h[4] = {bn_act(v.x, k, 0), bn_act(v.y, k, 1), bn_act(v.z, k, 2), bn_act(v.w, k, 3)};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (h[i] > m[i] || h[i] != h[i]) { m[i] = h[i]; am[i] = kk; }
        };
        int kk = ks;
        for (; kk + 3 * S < K; kk += 4 * S) {  // this lane's rows four at a time (loads in flight together), same order
            const float4 v0 = *reinterpret_cast<const float4 *>(y + (long)kk * ldy);
            const float4 v1 = *reinterpret_cast<const float4 *>(y + (long)(kk + S) * ldy);
            const float4 v2 = *reinterpret_cast<const float4 *>(y + (long)(kk + 2 * S) * ldy);
            const float4 v3 = *reinterpret_cast<const float4 *>(y + (long)(kk + 3 * S) * ldy);
            take(v0, kk); take(v1, kk + S); take(v2, kk + 2 * S); take(v3, kk + 3 * S);
        }
        for (; kk < K; kk += S) take(*reinterpret_cast<const float4 *>(y + (long)kk * ldy), kk);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { pm[threadIdx.x][i] = m[i]; pa[threadIdx.x][i] = am[i]; }
    __syncthreads();
    if (ks == 0 && live) {
        for (int l = 1; l < S; ++l) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float o = pm[l * ipb + il][i];
                const int oa = pa[l * ipb + il][i];
                const bool on = o != o, mn = m[i] != m[i];
                if (on || mn) {                       // a NaN wins; among NaNs the last row (the serial update's result)
                    if (on && (!mn || oa > am[i])) { m[i] = o; am[i] = oa; }
                } else if (o > m[i] || (o == m[i] && oa < am[i])) {  // the first row among equal maxima
                    m[i] = o; am[i] = oa;
                }
            }
        }
        *reinterpret_cast<float4 *>(out + grp * A.ldo + 4 * q) = make_float4(relu_nan(m[0]), relu_nan(m[1]), relu_nan(m[2]), relu_nan(m[3]));
        *reinterpret_cast<int4 *>(arg + grp * C + 4 * q) = make_int4(am[0], am[1], am[2], am[3]);
    }
}
__global__ void __launch_bounds__(kTT)
bn_relu_max_split_kernel(MaxSplitArgs a) { bn_relu_max_split_body(a, blockIdx.x); }
__global__ void __launch_bounds__(kTT)
bn_relu_max_split_pair_kernel(MaxSplitArgs a, MaxSplitArgs b, unsigned na) {
    if (blockIdx.x < na) bn_relu_max_split_body(a, blockIdx.x);
    else bn_relu_max_split_body(b, blockIdx.x - na);
}

struct BnSaved {
    float mean[4], invstd[4], g[4], b[4];
};

__device__ __forceinline__ void load_saved(BnCh &k, int c0, const float *__restrict__ mean, const float *__restrict__ invstd,
                                           const float *__restrict__ gamma, const float *__restrict__ beta) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k.mean[i] = mean[c0 + i]; k.invstd[i] = invstd[c0 + i]; k.g[i] = gamma[c0 + i]; k.b[i] = beta[c0 + i];
    }
}

__global__ void __launch_bounds__(kTT)
bn_relu_bwd_reduce_kernel(long rows, int C, const float *__restrict__ dH, int ldd, const float *__restrict__ Y, int ldy,
                          const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
                          const float *__restrict__ beta, int rows_per_block, int relu, double *__restrict__ sums,
                          const int *__restrict__ arg, int K, float *__restrict__ g_out = nullptr, int ldg = 0) {
    const int Q = C >> 2, rpp = kTT / Q;
    const int q = threadIdx.x % Q, rr = threadIdx.x / Q;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float4 s = make_float4(0, 0, 0, 0), t = s;
    if (rr < rpp) {
        BnCh k;
        load_saved(k, 4 * q, mean, invstd, gamma, beta);
        auto acc = [&](const float4 y, float4 g, long row = -1) {
            if (relu) {
                if (!(bn_act(y.x, k, 0) > 0.f)) g.x = 0.f;
                if (!(bn_act(y.y, k, 1) > 0.f)) g.y = 0.f;
                if (!(bn_act(y.z, k, 2) > 0.f)) g.z = 0.f;
                if (!(bn_act(y.w, k, 3) > 0.f)) g.w = 0.f;
            }
            if (g_out && row >= 0) *reinterpret_cast<float4 *>(g_out + row * ldg + 4 * q) = g;  // the masked / routed gradient, dense
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            t.x += g.x * ((y.x - k.mean[0]) * k.invstd[0]);
            t.y += g.y * ((y.y - k.mean[1]) * k.invstd[1]);
            t.z += g.z * ((y.z - k.mean[2]) * k.invstd[2]);
            t.w += g.w * ((y.w - k.mean[3]) * k.invstd[3]);
        };
        long r = r0 + rr;
        if (arg) {  // dH is d(max over K) (rows / K groups): only the arg-max row of a group receives it
            for (; r < r1; r += rpp)
                acc(*reinterpret_cast<const float4 *>(Y + r * ldy + 4 * q), max_grad(dH, ldd, arg, C, K, r, q), r);
        }
        for (; r + 3L * rpp < r1; r += 4L * rpp) {  // eight independent 16-byte loads in flight per thread
            const float *py = Y + r * ldy + 4 * q, *pg = dH + r * ldd + 4 * q;
            const float4 y0 = *reinterpret_cast<const float4 *>(py), g0 = *reinterpret_cast<const float4 *>(pg);
            const float4 y1 = *reinterpret_cast<const float4 *>(py + (long)rpp * ldy), g1 = *reinterpret_cast<const float4 *>(pg + (long)rpp * ldd);
            const float4 y2 = *reinterpret_cast<const float4 *>(py + 2L * rpp * ldy), g2 = *reinterpret_cast<const float4 *>(pg + 2L * rpp * ldd);
            const float4 y3 = *reinterpret_cast<const float4 *>(py + 3L * rpp * ldy), g3 = *reinterpret_cast<const float4 *>(pg + 3L * rpp * ldd);
            acc(y0, g0); acc(y1, g1); acc(y2, g2); acc(y3, g3);
        }
        for (; r < r1; r += rpp)
            acc(*reinterpret_cast<const float4 *>(Y + r * ldy + 4 * q), *reinterpret_cast<const float4 *>(dH + r * ldd + 4 * q));
    }
    block_reduce_to_sums(s, t, Q, rpp, C, sums);
}

__global__ void __launch_bounds__(kTT)
bn_relu_bwd_apply_kernel(long rows, int C, const float *__restrict__ dH, int ldd, const float *__restrict__ Y, int ldy,
                         const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
                         const float *__restrict__ beta, const double *__restrict__ sums, int rows_per_block, int relu,
                         float *__restrict__ dY, int ldo, float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dbias,
                         const int *__restrict__ arg, int K) {
    const int Q = C >> 2, rpp = kTT / Q;
    const int q = threadIdx.x % Q, rr = threadIdx.x / Q;
    if (rr >= rpp) return;
    BnCh k;
    load_saved(k, 4 * q, mean, invstd, gamma, beta);
    float sg[4], sgx[4], scale[4];
    const float inv_r = (float)(1.0 / (double)rows);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double a = 0.0, b = 0.0;
        for (int r = 0; r < kRep; ++r) {
            a += sums[(size_t)r * 2 * C + 4 * q + i];
            b += sums[(size_t)r * 2 * C + C + 4 * q + i];
        }
        sg[i] = (float)a;
        sgx[i] = (float)b;
        scale[i] = k.g[i] * k.invstd[i];
    }
    if (blockIdx.x == 0 && rr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dgamma[4 * q + i] = sgx[i];
            dbeta[4 * q + i] = sg[i];
            if (dbias) dbias[4 * q + i] = 0.f;  // gradient of a bias in front of BatchNorm: sum_r dy = 0 identically
        }
    }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    auto dy_of = [&](const float4 y, const float4 g) {
        const float yy[4] = {y.x, y.y, y.z, y.w};
        float gg[4] = {g.x, g.y, g.z, g.w}, o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (relu && !(bn_act(yy[i], k, i) > 0.f)) gg[i] = 0.f;
            const float xhat = (yy[i] - k.mean[i]) * k.invstd[i];
            o[i] = scale[i] * (gg[i] - sg[i] * inv_r - xhat * (sgx[i] * inv_r));
        }
        return make_float4(o[0], o[1], o[2], o[3]);
    };
    long r = r0 + rr;
    if (!arg) {
        for (; r + 3L * rpp < r1; r += 4L * rpp) {  // eight independent 16-byte loads in flight per thread
            const float *py = Y + r * ldy + 4 * q, *pg = dH + r * ldd + 4 * q;
            const float4 y0 = *reinterpret_cast<const float4 *>(py), g0 = *reinterpret_cast<const float4 *>(pg);
            const float4 y1 = *reinterpret_cast<const float4 *>(py + (long)rpp * ldy), g1 = *reinterpret_cast<const float4 *>(pg + (long)rpp * ldd);
            const float4 y2 = *reinterpret_cast<const float4 *>(py + 2L * rpp * ldy), g2 = *reinterpret_cast<const float4 *>(pg + 2L * rpp * ldd);
            const float4 y3 = *reinterpret_cast<const float4 *>(py + 3L * rpp * ldy), g3 = *reinterpret_cast<const float4 *>(pg + 3L * rpp * ldd);
            float *o = dY + r * ldo + 4 * q;
            *reinterpret_cast<float4 *>(o) = dy_of(y0, g0);
            *reinterpret_cast<float4 *>(o + (long)rpp * ldo) = dy_of(y1, g1);
            *reinterpret_cast<float4 *>(o + 2L * rpp * ldo) = dy_of(y2, g2);
            *reinterpret_cast<float4 *>(o + 3L * rpp * ldo) = dy_of(y3, g3);
        }
    }
    for (; r < r1; r += rpp) {
        const float4 y = *reinterpret_cast<const float4 *>(Y + r * ldy + 4 * q);
        const float4 g = arg ? max_grad(dH, ldd, arg, C, K, r, q) : *reinterpret_cast<const float4 *>(dH + r * ldd + 4 * q);
        *reinterpret_cast<float4 *>(dY + r * ldo + 4 * q) = dy_of(y, g);
    }
}

// The same pass for the first layer of a set-abstraction scale, which ALSO forms the partial sums of d(W_xyz) = dY^T rel (rel
// (rows x 3): the relative coordinates sa_layer1 saved): dY is in registers here anyway, and the separate pass over it
// (rows_outer3_kernel) re-read 33 MB per scale.  partial: [gridDim.x][C][3], summed by rows_outer3_sum_kernel.
struct ApplyRelArgs {
    long rows; int C; const float *dH; int ldd; const float *Y; int ldy;
    const float *mean, *invstd, *gamma, *beta; const double *sums; int rows_per_block, relu;
    float *dY; int ldo; float *dgamma, *dbeta, *dbias; const float *rel; float *partial;
};
// (bx: this problem's workgroup index -- two problems of one module's two neighbourhood sizes share a launch, below)
__device__ __forceinline__ void bn_relu_bwd_apply_rel_body(const ApplyRelArgs &A, unsigned bx) {
    const long rows = A.rows;
    const int C = A.C, ldd = A.ldd, ldy = A.ldy, rows_per_block = A.rows_per_block, relu = A.relu, ldo = A.ldo;
    const float *__restrict__ dH = A.dH, *__restrict__ Y = A.Y, *__restrict__ mean = A.mean, *__restrict__ invstd = A.invstd;
    const float *__restrict__ gamma = A.gamma, *__restrict__ beta = A.beta, *__restrict__ rel = A.rel;
    const double *__restrict__ sums = A.sums;
    float *__restrict__ dY = A.dY, *__restrict__ dgamma = A.dgamma, *__restrict__ dbeta = A.dbeta, *__restrict__ dbias = A.dbias;
    float *__restrict__ partial = A.partial;

    __shared__ float red[12][kTT];
    const int Q = C >> 2, rpp = kTT / Q;
    const int q = threadIdx.x % Q, rr = threadIdx.x / Q;
    float a[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (rr < rpp) {
        BnCh k;
        load_saved(k, 4 * q, mean, invstd, gamma, beta);
        float sg[4], sgx[4], scale[4];
        const float inv_r = (float)(1.0 / (double)rows);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double sa = 0.0, sb = 0.0;
            for (int r = 0; r < kRep; ++r) {
                sa += sums[(size_t)r * 2 * C + 4 * q + i];
                sb += sums[(size_t)r * 2 * C + C + 4 * q + i];
            }
            sg[i] = (float)sa;
            sgx[i] = (float)sb;
            scale[i] = k.g[i] * k.invstd[i];
        }
        if (bx == 0 && rr == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dgamma[4 * q + i] = sgx[i];
                dbeta[4 * q + i] = sg[i];
                if (dbias) dbias[4 * q + i] = 0.f;
            }
        }
        const long r0 = (long)bx * rows_per_block;
        const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
        auto one_row = [&](long r, float4 y, float4 g, float e0, float e1, float e2) {
            const float yy[4] = {y.x, y.y, y.z, y.w};
            float gg[4] = {g.x, g.y, g.z, g.w}, o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (relu && !(bn_act(yy[i], k, i) > 0.f)) gg[i] = 0.f;
                const float xhat = (yy[i] - k.mean[i]) * k.invstd[i];
                o[i] = scale[i] * (gg[i] - sg[i] * inv_r - xhat * (sgx[i] * inv_r));  // same expression as bn_relu_bwd_apply_kernel
                a[i][0] += o[i] * e0; a[i][1] += o[i] * e1; a[i][2] += o[i] * e2;
            }
            *reinterpret_cast<float4 *>(dY + r * ldo + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
        };
        // four rows per round, their ten loads in flight together (same values, same order of the d(W_xyz) additions)
        typedef float v4 __attribute__((ext_vector_type(4)));
        long r = r0 + rr;
        for (; r + 3L * rpp < r1; r += 4L * rpp) {
            v4 y[4], g[4];
            float e[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long ru = r + (long)u * rpp;
                y[u] = *reinterpret_cast<const v4 *>(Y + ru * ldy + 4 * q);
                g[u] = *reinterpret_cast<const v4 *>(dH + ru * ldd + 4 * q);
                e[u][0] = rel[ru * 3 + 0]; e[u][1] = rel[ru * 3 + 1]; e[u][2] = rel[ru * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                one_row(r + (long)u * rpp, make_float4(y[u].x, y[u].y, y[u].z, y[u].w), make_float4(g[u].x, g[u].y, g[u].z, g[u].w),
                        e[u][0], e[u][1], e[u][2]);
        }
        for (; r < r1; r += rpp)
            one_row(r, *reinterpret_cast<const float4 *>(Y + r * ldy + 4 * q), *reinterpret_cast<const float4 *>(dH + r * ldd + 4 * q),
                    rel[r * 3 + 0], rel[r * 3 + 1], rel[r * 3 + 2]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) red[i * 3 + t][threadIdx.x] = a[i][t];
    __syncthreads();
    if (rr == 0) {
        float *o = partial + (size_t)bx * 3 * C + 12 * q;  // out[c][t], c = 4 q + i
#pragma unroll
        for (int kk = 0; kk < 12; ++kk) {
            float s = red[kk][q];
            for (int l = 1; l < rpp; ++l) s += red[kk][l * Q + q];
            o[kk] = s;
        }
    }
}

__global__ void __launch_bounds__(kTT)
bn_relu_bwd_apply_rel_kernel(ApplyRelArgs a) { bn_relu_bwd_apply_rel_body(a, blockIdx.x); }
// two problems in one launch: workgroups [0, na) run a, the rest b
__global__ void __launch_bounds__(kTT)
bn_relu_bwd_apply_rel_pair_kernel(ApplyRelArgs a, ApplyRelArgs b, unsigned na) {
    if (blockIdx.x < na) bn_relu_bwd_apply_rel_body(a, blockIdx.x);
    else bn_relu_bwd_apply_rel_body(b, blockIdx.x - na);
}

// ---- transposes of the row gathers ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kTT)
scatter_add_rows_kernel(int n, int m, int Q, const float *__restrict__ dOut, int ldo, const int *__restrict__ idx,
                        float *__restrict__ dIn, int ldi) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * kTT + threadIdx.x;
    if (e >= (long)m * Q) return;
    const int j = (int)(e / Q), q = (int)(e % Q);
    const int row = idx[(size_t)b * m + j];
    const float4 v = *reinterpret_cast<const float4 *>(dOut + ((size_t)b * m + j) * ldo + 4 * q);
    float *dst = dIn + ((size_t)b * n + row) * ldi + 4 * q;
    unsafeAtomicAdd(dst + 0, v.x);
    unsafeAtomicAdd(dst + 1, v.y);
    unsafeAtomicAdd(dst + 2, v.z);
    unsafeAtomicAdd(dst + 3, v.w);
}

__global__ void __launch_bounds__(kTT)
interp_pm_bwd_kernel(int m, int n, int Q, const float *__restrict__ dOut, int ldo, const int *__restrict__ idx,
                     const float *__restrict__ weight, float *__restrict__ dPts, int ldp) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * kTT + threadIdx.x;
    if (e >= (long)n * Q) return;
    const int j = (int)(e / Q), q = (int)(e % Q);
    const float4 v = *reinterpret_cast<const float4 *>(dOut + ((size_t)b * n + j) * ldo + 4 * q);
    const int *id = idx + ((size_t)b * n + j) * 3;
    const float *w = weight + ((size_t)b * n + j) * 3;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float *dst = dPts + ((size_t)b * m + id[t]) * ldp + 4 * q;
        const float wt = w[t];
        unsafeAtomicAdd(dst + 0, wt * v.x);
        unsafeAtomicAdd(dst + 1, wt * v.y);
        unsafeAtomicAdd(dst + 2, wt * v.z);
        unsafeAtomicAdd(dst + 3, wt * v.w);
    }
}

// ---- layer 1 of a set-abstraction scale, pre-activation, never materialising the grouped input ---------------------------
// y1[b, (s,k), :] = a1f[b, idx[b,s,k], :] + wx . (xyz[b, idx] - cxyz[b,s]) + cadd[b,s,:]     (each term optional)
// (the same linear split as the eval kernel, pn2_ext.h pn2x_sa_mlp_max; bias omitted: BatchNorm follows).  Also writes the
// relative coordinates rel (b, s*k, 3) the backward needs for d(wx).
__global__ void __launch_bounds__(kTT)
sa_layer1_kernel(int n, int S, int K, int Q, const float *__restrict__ a1f, int a1f_ld, const float *__restrict__ xyz,
                 const float *__restrict__ cxyz, const float *__restrict__ wx, int wx_ld, const float *__restrict__ cadd, int cadd_ld,
                 const int *__restrict__ idx, float *__restrict__ out, float *__restrict__ rel_out) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * kTT + threadIdx.x;
    const int sk = S * K;
    if (e >= (long)sk * Q) return;
    const int r = (int)(e / Q), q = (int)(e % Q);
    const int s = r / K;
    const int j = idx[(size_t)b * sk + r];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a1f) acc = *reinterpret_cast<const float4 *>(a1f + ((size_t)b * n + j) * a1f_ld + 4 * q);
    if (xyz) {
        const float *p = xyz + ((size_t)b * n + j) * 3, *c = cxyz + ((size_t)b * S + s) * 3;
        const float rx = p[0] - c[0], ry = p[1] - c[1], rz = p[2] - c[2];
        const float *w0 = wx + (size_t)(4 * q) * wx_ld, *w1 = w0 + wx_ld, *w2 = w1 + wx_ld, *w3 = w2 + wx_ld;  // rows of a (C1, 3) block, wx_ld apart
        acc.x += w0[0] * rx + w0[1] * ry + w0[2] * rz;
        acc.y += w1[0] * rx + w1[1] * ry + w1[2] * rz;
        acc.z += w2[0] * rx + w2[1] * ry + w2[2] * rz;
        acc.w += w3[0] * rx + w3[1] * ry + w3[2] * rz;
        if (rel_out && q == 0) {
            float *o = rel_out + ((size_t)b * sk + r) * 3;
            o[0] = rx; o[1] = ry; o[2] = rz;
        }
    }
    if (cadd) {
        const float4 v = *reinterpret_cast<const float4 *>(cadd + ((size_t)b * S + s) * cadd_ld + 4 * q);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + ((size_t)b * sk + r) * (4 * Q) + 4 * q) = acc;
}

// The same with the BatchNorm statistics of the result taken on the way (pn2x_sa_layer1_stats): a workgroup owns rows_per_block
// consecutive slots of one cloud, a thread a channel quad of every rpp-th of them, and the per-thread sums of y and y^2 go through
// block_reduce_to_sums like pn2x_bn_stats' -- the separate pass over y1 (one launch per scale and step) disappears.
struct SaLayer1Args {
    int n, S, K, Q; const float *a1f; int a1f_ld; const float *xyz, *cxyz, *wx; int wx_ld; const float *cadd; int cadd_ld;
    const int *idx; float *out, *rel_out; int rows_per_block; double *sums;
};
__device__ __forceinline__ void sa_layer1_stats_body(const SaLayer1Args &A, unsigned bx, unsigned by) {
    const int n = A.n, S = A.S, K = A.K, Q = A.Q, a1f_ld = A.a1f_ld, wx_ld = A.wx_ld, cadd_ld = A.cadd_ld, rows_per_block = A.rows_per_block;
    const float *__restrict__ a1f = A.a1f, *__restrict__ xyz = A.xyz, *__restrict__ cxyz = A.cxyz, *__restrict__ wx = A.wx;
    const float *__restrict__ cadd = A.cadd;
    const int *__restrict__ idx = A.idx;
    float *__restrict__ out = A.out, *__restrict__ rel_out = A.rel_out;
    double *__restrict__ sums = A.sums;

    const int b = (int)by;
    const int sk = S * K, rpp = kTT / Q;  // kTT % Q == 0 (checked by the launcher)
    const int q = threadIdx.x % Q, rr = threadIdx.x / Q;
    const int r0 = (int)bx * rows_per_block, r1 = min(sk, r0 + rows_per_block);
    // Rows go through the workgroup in batches of kTT.  Phase A: thread t fetches what row r + t needs besides its feature quads --
    // neighbour index, the neighbour's and the centre's coordinates -- ONCE (the Q threads that share a row used to fetch them Q
    // times: 56 load instructions per 128-byte output row at sa1, which made this kernel load-issue bound at 1.4 TB/s), leaves
    // index and relative coordinates in LDS and writes the relative coordinates out with one 12-byte run per thread.  Phase B:
    // thread (q, rr) forms channel quad q of rows rr, rr + rpp, ... of the batch from LDS + its own feature / centre-term quads,
    // four rows per round with their gathers in flight together.  Same floats, same order of the statistics' additions as one
    // thread per (row, quad).
    __shared__ int s_j[kTT];
    __shared__ float s_rel[kTT][3];
    float w[4][3] = {{0.f}};
    if (xyz) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) w[i][c] = wx[(size_t)(4 * q + i) * wx_ld + c];
    }
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sq = sm;
    typedef float v4 __attribute__((ext_vector_type(4)));  // (an array of float4 STRUCTS stays in scratch memory)
    auto finish = [&](int r, int lr, v4 a, v4 cv) {
        float4 acc = make_float4(a.x, a.y, a.z, a.w);
        if (xyz) {
            const float rx = s_rel[lr][0], ry = s_rel[lr][1], rz = s_rel[lr][2];
            acc.x += w[0][0] * rx + w[0][1] * ry + w[0][2] * rz;  // the expressions of sa_layer1_kernel: same floats
            acc.y += w[1][0] * rx + w[1][1] * ry + w[1][2] * rz;
            acc.z += w[2][0] * rx + w[2][1] * ry + w[2][2] * rz;
            acc.w += w[3][0] * rx + w[3][1] * ry + w[3][2] * rz;
        }
        if (cadd) { acc.x += cv.x; acc.y += cv.y; acc.z += cv.z; acc.w += cv.w; }
        *reinterpret_cast<float4 *>(out + ((size_t)b * sk + r) * (4 * Q) + 4 * q) = acc;
        sm.x += acc.x; sm.y += acc.y; sm.z += acc.z; sm.w += acc.w;
        sq.x += acc.x * acc.x; sq.y += acc.y * acc.y; sq.z += acc.z * acc.z; sq.w += acc.w * acc.w;
    };
    const v4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int rb = r0; rb < r1; rb += kTT) {
        __syncthreads();  // (the previous batch's rows are consumed)
        {
            const int r = rb + (int)threadIdx.x;
            if (r < r1) {
                const int j = idx[(size_t)b * sk + r];
                s_j[threadIdx.x] = j;
                if (xyz) {
                    const float *p = xyz + ((size_t)b * n + j) * 3, *c = cxyz + ((size_t)b * S + r / K) * 3;
                    const float rx = p[0] - c[0], ry = p[1] - c[1], rz = p[2] - c[2];
                    s_rel[threadIdx.x][0] = rx; s_rel[threadIdx.x][1] = ry; s_rel[threadIdx.x][2] = rz;
                    if (rel_out) {
                        float *o = rel_out + ((size_t)b * sk + r) * 3;
                        o[0] = rx; o[1] = ry; o[2] = rz;
                    }
                }
            }
        }
        __syncthreads();
        const int nb = min(kTT, r1 - rb);
        int lr = rr;
        for (; lr + 3 * rpp < nb; lr += 4 * rpp) {
            v4 a[4], cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int l = lr + u * rpp;
                a[u] = a1f ? *reinterpret_cast<const v4 *>(a1f + ((size_t)b * n + s_j[l]) * a1f_ld + 4 * q) : z4;
                cv[u] = cadd ? *reinterpret_cast<const v4 *>(cadd + ((size_t)b * S + (rb + l) / K) * cadd_ld + 4 * q) : z4;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) finish(rb + lr + u * rpp, lr + u * rpp, a[u], cv[u]);
        }
        for (; lr < nb; lr += rpp) {
            const v4 a = a1f ? *reinterpret_cast<const v4 *>(a1f + ((size_t)b * n + s_j[lr]) * a1f_ld + 4 * q) : z4;
            const v4 cv = cadd ? *reinterpret_cast<const v4 *>(cadd + ((size_t)b * S + (rb + lr) / K) * cadd_ld + 4 * q) : z4;
            finish(rb + lr, lr, a, cv);
        }
    }
    block_reduce_to_sums(sm, sq, Q, rpp, 4 * Q, sums);
}

__global__ void __launch_bounds__(kTT)
sa_layer1_stats_kernel(SaLayer1Args a) { sa_layer1_stats_body(a, blockIdx.x, blockIdx.y); }
// the two neighbourhood sizes of one module in one launch: workgroup columns [0, na) run a, the rest b (same clouds along y)
__global__ void __launch_bounds__(kTT)
sa_layer1_stats_pair_kernel(SaLayer1Args a, SaLayer1Args b, unsigned na) {
    if (blockIdx.x < na) sa_layer1_stats_body(a, blockIdx.x, blockIdx.y);
    else sa_layer1_stats_body(b, blockIdx.x - na, blockIdx.y);
}

// LDS-slab variant of the two row scatters: a workgroup owns (cloud, cc channels), accumulates every contribution to its
// [n_dst][cc] slab with ds_add_f32 and adds the slab to the destination once with coalesced 16-byte read-modify-writes --
// no global atomics (same idea as group_bwd_lds_kernel / interp_bwd_lds_kernel on the channel-major operators).
template <bool INTERP>
__global__ void __launch_bounds__(kTT)
scatter_rows_lds_kernel(int n_dst, int m_src, int C, int cc, const float *__restrict__ dOut, int ldo, const int *__restrict__ idx,
                        const float *__restrict__ weight, float *__restrict__ dIn, int ldi) {
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [n_dst][cc]
    const int b = blockIdx.y, c0 = blockIdx.x * cc;
    const int nc = (C - c0) < cc ? (C - c0) : cc;
    const int Qc = nc >> 2;
    for (int i = threadIdx.x; i < n_dst * cc; i += kTT) slab[i] = 0.f;
    __syncthreads();
    const int total = m_src * Qc;
    for (int e = threadIdx.x; e < total; e += kTT) {
        const int j = e / Qc, q = e - j * Qc;
        const float4 v = *reinterpret_cast<const float4 *>(dOut + ((size_t)b * m_src + j) * ldo + c0 + 4 * q);
        if constexpr (INTERP) {
            const int *id = idx + ((size_t)b * m_src + j) * 3;
            const float *w = weight + ((size_t)b * m_src + j) * 3;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float *dst = slab + lds_index(id[t], n_dst) * cc + 4 * q;
                const float wt = w[t];
                atomicAdd(dst + 0, wt * v.x); atomicAdd(dst + 1, wt * v.y); atomicAdd(dst + 2, wt * v.z); atomicAdd(dst + 3, wt * v.w);
            }
        } else {
            float *dst = slab + lds_index(idx[(size_t)b * m_src + j], n_dst) * cc + 4 * q;
            atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_dst * Qc; e += kTT) {
        const int row = e / Qc, q = e - row * Qc;
        float4 *dst = reinterpret_cast<float4 *>(dIn + ((size_t)b * n_dst + row) * ldi + c0 + 4 * q);
        const float4 a = *reinterpret_cast<const float4 *>(slab + row * cc + 4 * q);
        float4 d = *dst;
        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
        *dst = d;
    }
}

// channels per workgroup for the slab kernels: the slab must fit 64 KiB; prefer >= 512 workgroups; 0 = use the atomic kernel
static int slab_channels(int b, int n_dst, int C) {
    int cc = (16384 / n_dst) & ~3;
    if (cc < 4) return 0;
    if (cc > 64) cc = 64;
    if (cc > C) cc = C;
    while (cc > 4 && (long)b * ((C + cc - 1) / cc) < 512) cc = ((cc / 2) + 3) & ~3;
    return cc;
}

// ---- atomics-free row scatters: inverse index (CSR) + owner-computes segment sum --------------------------------------------
// fp32 LDS / L2 atomics retire about one lane per clock per CU on this part, so the slab kernels above top out around
// 0.2-0.25 T adds/s.  Here the index list of a cloud is inverted once (counting sort by target row: offsets + order), and
// every (target row, channel quad) SUMS its contributions with plain coalesced 16-byte row reads -- no atomics in the data
// path, nothing to pre-zero, every output row written exactly once.
__device__ __forceinline__ void inverse_index_body(int n_dst, int L, const int *__restrict__ idx, int *__restrict__ offsets,
                                                   int *__restrict__ order) {
    extern __shared__ int cnt[];  // [n_dst + 1] counts -> running cursors; then [kTT] chunk totals
    int *part = cnt + n_dst + 1;
    for (int i = threadIdx.x; i <= n_dst; i += kTT) cnt[i] = 0;
    __syncthreads();
    auto key = [&](int e) { const int k = idx[e]; return k < 0 ? 0 : (k >= n_dst ? n_dst - 1 : k); };  // (out-of-range indices cannot corrupt LDS)
    for (int e = threadIdx.x; e < L; e += kTT) atomicAdd(&cnt[key(e)], 1);
    __syncthreads();
    // exclusive scan: each thread owns a contiguous chunk
    const int chunk = (n_dst + kTT - 1) / kTT;
    const int i0 = threadIdx.x * chunk, i1 = (i0 + chunk) < n_dst ? (i0 + chunk) : n_dst;
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += cnt[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < kTT; ++t) { const int v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = i0; i < i1; ++i) {
        const int v = cnt[i];
        offsets[i] = run;
        cnt[i] = run;  // cursor
        run += v;
    }
    if (threadIdx.x == 0) offsets[n_dst] = L;
    __syncthreads();
    for (int e = threadIdx.x; e < L; e += kTT) order[atomicAdd(&cnt[key(e)], 1)] = e;
}

__global__ void __launch_bounds__(kTT)
inverse_index_kernel(int n_dst, int L, const int *__restrict__ idx_all, int *__restrict__ offsets_all, int *__restrict__ order_all) {
    const int b = blockIdx.x;
    inverse_index_body(n_dst, L, idx_all + (size_t)b * L, offsets_all + (size_t)b * (n_dst + 1), order_all + (size_t)b * L);
}

// The same sort for every chunk of mt consecutive positions of a cloud's list on its own (scatter_cm.hip, long lists), with
// 16-bit outputs (mt <= 4 * kInvT): workgroup (cloud b, chunk k) -> offsets [b][k][off_stride], order [b][k * mt ...] holding
// positions RELATIVE to the chunk.  1024 threads: a thread keeps its <= 4 entries' targets in registers between the histogram and the
// scatter, the counters' prefix sums go per thread (n_dst / 1024 counters) -> per wave (shuffles) -> over the 16 wave totals; the
// 256-thread version above spent 28 us on 32 serial counters per thread, a 256-entry LDS scan and a second read of the index list.
constexpr int kInvT = 1024;
__global__ void __launch_bounds__(kInvT)
inverse_index_chunked_kernel(int n_dst, int L, int mt, int nchunks, int off_stride, const int *__restrict__ idx_all,
                             unsigned short *__restrict__ offsets_all, unsigned short *__restrict__ order_all) {
    extern __shared__ int cnt[];  // [n_dst + 1] counts -> running cursors; then [16] wave totals
    int *wtot = cnt + n_dst + 1;
    const int b = blockIdx.x / nchunks, k = blockIdx.x - b * nchunks;
    const int e0 = k * mt, len = (L - e0) < mt ? (L - e0) : mt;
    const int *__restrict__ idx = idx_all + (size_t)b * L + e0;
    unsigned short *__restrict__ offsets = offsets_all + (size_t)blockIdx.x * off_stride;
    unsigned short *__restrict__ order = order_all + (size_t)b * L + e0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int key[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {  // (requested before the counters are cleared: the loads overlap the clearing)
        const int e = tid + x * kInvT;
        int t = e < len ? idx[e] : 0;
        key[x] = t < 0 ? 0 : (t >= n_dst ? n_dst - 1 : t);  // (out-of-range indices cannot corrupt LDS)
    }
    for (int i = tid; i <= n_dst; i += kInvT) cnt[i] = 0;
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 4; ++x)
        if (tid + x * kInvT < len) atomicAdd(&cnt[key[x]], 1);
    __syncthreads();
    const int chunk = (n_dst + kInvT - 1) / kInvT;
    const int i0 = tid * chunk, i1 = (i0 + chunk) < n_dst ? (i0 + chunk) : n_dst;
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += cnt[i];
    int incl = sum;  // inclusive scan over the wave's 64 thread sums
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wtot[w];  // (<= 15 broadcast reads)
    int run = base + incl - sum;
    for (int i = i0; i < i1; ++i) {
        const int v = cnt[i];
        offsets[i] = (unsigned short)run;
        cnt[i] = run;  // cursor
        run += v;
    }
    if (tid == 0) offsets[n_dst] = (unsigned short)len;
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int e = tid + x * kInvT;
        if (e < len) order[atomicAdd(&cnt[key[x]], 1)] = (unsigned short)e;
    }
}

// acc += sum over p = p0, p0 + step, ... < p1 of [w] . row(order[p]), in that order.  Four list entries per round: their source
// indices first, then the four row segments (and weights) in flight together -- an entry is two dependent memory round trips, and
// one entry at a time made the long segments (ball-query padding) latency chains.  Same additions in the same order.
template <int T>
__device__ __forceinline__ void segment_rows_sum(float4 &acc, int p0, int p1, int step, const int *__restrict__ order,
                                                 const float *__restrict__ src, int ldo, const float *__restrict__ weight) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    int p = p0;
    for (; p + 3 * step < p1; p += 4 * step) {
        int e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = order[p + u * step];
        v4 v[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = T == 1 ? e[u] : e[u] / T;
            v[u] = *reinterpret_cast<const v4 *>(src + (size_t)j * ldo);
            w[u] = T == 1 ? 1.f : weight[e[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (T == 1) {
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
            } else {
                acc.x += w[u] * v[u].x; acc.y += w[u] * v[u].y; acc.z += w[u] * v[u].z; acc.w += w[u] * v[u].w;
            }
        }
    }
    for (; p < p1; p += step) {
        const int e = order[p];
        const int j = T == 1 ? e : e / T;
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)j * ldo);
        if constexpr (T == 1) {
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        } else {
            const float w = weight[e];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
    }
}

struct SegSumArgs {
    int n_dst, m_src, Q; const float *dOut; int ldo; const int *offsets_all, *order_all; const float *weight_all; float *dIn; int ldi, accumulate;
};
template <int T>  // index entries per source row: 1 = plain row scatter, 3 = three-NN interpolation (weighted)
__device__ __forceinline__ void rows_segment_sum_body(const SegSumArgs &A, unsigned bx, unsigned by) {
    const int n_dst = A.n_dst, m_src = A.m_src, Q = A.Q, ldo = A.ldo, ldi = A.ldi, accumulate = A.accumulate;
    const float *__restrict__ dOut = A.dOut, *__restrict__ weight_all = A.weight_all;
    const int *__restrict__ offsets_all = A.offsets_all, *__restrict__ order_all = A.order_all;
    float *__restrict__ dIn = A.dIn;
    const int b = (int)by;
    const long item = (long)bx * kTT + threadIdx.x;
    if (item >= (long)n_dst * Q) return;
    const int i = (int)(item / Q), q = (int)(item % Q);
    const int *__restrict__ offsets = offsets_all + (size_t)b * (n_dst + 1);
    const int *__restrict__ order = order_all + (size_t)b * m_src * T;
    const float *__restrict__ src = dOut + (size_t)b * m_src * ldo + 4 * q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int p1 = offsets[i + 1];
    segment_rows_sum<T>(acc, offsets[i], p1, 1, order, src, ldo, weight_all + (size_t)b * m_src * T);
    float4 *dst = reinterpret_cast<float4 *>(dIn + ((size_t)b * n_dst + i) * ldi + 4 * q);
    if (accumulate) {
        const float4 d = *dst;
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
    }
    *dst = acc;
}
template <int T>
__global__ void __launch_bounds__(kTT)
rows_segment_sum_kernel(SegSumArgs a) { rows_segment_sum_body<T>(a, blockIdx.x, blockIdx.y); }
// two scatters of one module's two neighbourhood sizes (different column blocks of the same destination rows) in one launch
__global__ void __launch_bounds__(kTT)
rows_segment_sum_pair_kernel(SegSumArgs a, SegSumArgs b, unsigned na) {
    if (blockIdx.x < na) rows_segment_sum_body<1>(a, blockIdx.x, blockIdx.y);
    else rows_segment_sum_body<1>(b, blockIdx.x - na, blockIdx.y);
}

// The same sums with a destination's segment split over EL lanes (EL a power of two, Q * EL divides 256): ball-query padding
// repeats a ball's first index up to nsample times, so some destinations receive hundreds of rows while most receive a few --
// with one thread per (destination, quad) the launch waits for those threads (59 us for 33 MB at sa2).
template <int T>
__global__ void __launch_bounds__(kTT)
rows_segment_sum_split_kernel(int n_dst, int m_src, int Q, int EL, const float *__restrict__ dOut, int ldo,
                              const int *__restrict__ offsets_all, const int *__restrict__ order_all,
                              const float *__restrict__ weight_all, float *__restrict__ dIn, int ldi, int accumulate) {
    __shared__ float4 red[kTT];
    const int b = blockIdx.y;
    const int dpb = kTT / (Q * EL);  // destinations per workgroup
    const int q = threadIdx.x % Q, el = (threadIdx.x / Q) % EL, dl = threadIdx.x / (Q * EL);
    const int i = blockIdx.x * dpb + dl;
    const int *__restrict__ offsets = offsets_all + (size_t)b * (n_dst + 1);
    const int *__restrict__ order = order_all + (size_t)b * m_src * T;
    const float *__restrict__ src = dOut + (size_t)b * m_src * ldo + 4 * q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n_dst) {
        const int p1 = offsets[i + 1];
        segment_rows_sum<T>(acc, offsets[i] + el, p1, EL, order, src, ldo, weight_all + (size_t)b * m_src * T);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (el == 0 && i < n_dst) {
        for (int l = 1; l < EL; ++l) {
            const float4 o = red[(dl * EL + l) * Q + q];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        float4 *dst = reinterpret_cast<float4 *>(dIn + ((size_t)b * n_dst + i) * ldi + 4 * q);
        if (accumulate) {
            const float4 d = *dst;
            acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        }
        *dst = acc;
    }
}

static int rows_per_block_for(long rows, int C, bool stats_only = false) {
    const int rpp = kTT / (C >> 2);
    // ~1024 workgroups on a large problem; 512 for the reductions: every workgroup ends in 2 C fp64 atomics onto 8 accumulator copies,
    // and with 1024 of them that tail was ~20 us of a step's reduction launches (sweep 256 / 512 / 1024 / 2048: 2.562 / 2.560 / 2.580 /
    // 2.580 ms per step)
    const long tg = stats_only ? 512 : 1024;
    long rpb = (rows + tg - 1) / tg;
    // A thread walks rows_per_block / rpp rows.  Small tensors (the 4096 - 8192-row stacks of a 32-cloud step: <= 8 MB) used
    // to get 16 rows per thread like the large ones -- 32 - 64 workgroups, each thread a chain of 16 dependent-in-time round
    // trips: 9 - 10 us for a 4 MB tensor.  Four rows per thread (all four loads in flight, see the kernels) fill the chip.
    // (not the statistics pass: its cost at this size is the 2 C fp64 atomics per workgroup: 38 -> 52 us over a step's seven launches)
    const long min_rows = ((rows * C <= (2L << 20) && !stats_only) ? 4L : 16L) * rpp;
    if (rpb < min_rows) rpb = min_rows;
    return (int)rpb;
}

static bool bad_c(int C) { return C < 4 || C % 4 != 0 || C > 1024; }

}  // namespace pn2

extern "C" int pn2x_bn_stats(long rows, int c, const float *y, int ldy, double *sums, void *stream) {
    using namespace pn2;
    if (rows < 1 || bad_c(c) || ldy < c || ldy % 4) return PN2_EINVAL;
    if (!y || !sums) return PN2_ENULL;
    if (((uintptr_t)y) % 16) return PN2_EINVAL;
    const int rpb = rows_per_block_for(rows, c, true);
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(kTT), 0, (hipStream_t)stream, rows, c, y, ldy, rpb, sums);
    return check_launch();
}

extern "C" int pn2x_bn_relu_apply(long rows, int c, const float *y, int ldy, const double *sums, const float *gamma,
                                  const float *beta, const float *conv_bias, float eps, float momentum, float *running_mean,
                                  float *running_var, long long *num_batches_tracked, float *save_mean, float *save_invstd,
                                  float *h, int ldh, int relu, void *stream) {
    using namespace pn2;
    if (rows < 1 || bad_c(c) || ldy < c || ldy % 4 || ldh < c || ldh % 4) return PN2_EINVAL;
    if (!y || !sums || !gamma || !beta || !save_mean || !save_invstd || !h) return PN2_ENULL;
    if (((uintptr_t)y | (uintptr_t)h) % 16) return PN2_EINVAL;
    const int rpb = rows_per_block_for(rows, c);
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(kTT), 0, (hipStream_t)stream, rows, c, y, ldy,
                       sums, gamma, beta, conv_bias, eps, momentum, running_mean, running_var, num_batches_tracked, save_mean,
                       save_invstd, h, ldh, rpb, relu);
    return check_launch();
}

extern "C" int pn2x_bn_relu_bwd(long rows, int c, const float *dh, int ldd, const float *y, int ldy, const float *mean,
                                const float *invstd, const float *gamma, const float *beta, int relu, double *sums, float *dy,
                                int ldo, float *dgamma, float *dbeta, float *dbias, void *stream) {
    using namespace pn2;
    if (rows < 1 || bad_c(c) || ldy < c || ldy % 4 || ldd < c || ldd % 4 || ldo < c || ldo % 4) return PN2_EINVAL;
    if (!dh || !y || !mean || !invstd || !gamma || !beta || !sums || !dy || !dgamma || !dbeta) return PN2_ENULL;
    if (((uintptr_t)y | (uintptr_t)dh | (uintptr_t)dy) % 16) return PN2_EINVAL;
    const int rpb = rows_per_block_for(rows, c);
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, grid, dim3(kTT), 0, (hipStream_t)stream, rows, c, dh, ldd, y, ldy, mean, invstd, gamma,
                       beta, rpb, relu, sums, (const int *)nullptr, 1);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, grid, dim3(kTT), 0, (hipStream_t)stream, rows, c, dh, ldd, y, ldy, mean, invstd, gamma,
                       beta, sums, rpb, relu, dy, ldo, dgamma, dbeta, dbias, (const int *)nullptr, 1);
    return check_launch();
}

// The two halves of pn2x_bn_relu_bwd / pn2x_bn_relu_max_bwd as separate entries, for the fused training stacks
// (train_gemm.hip): the top layer of a stack needs only the reduction (its dY is computed on load by the fused GEMMs),
// the first layer only the apply step (its pre-masked gradient and sums come from the dgrad epilogue of layer 2).
extern "C" int pn2x_bn_bwd_reduce(long rows, int c, const float *dh, int ldd, const int *arg, int k, const float *y, int ldy,
                                  const float *mean, const float *invstd, const float *gamma, const float *beta, int relu, double *sums,
                                  void *stream) {
    return pn2x_bn_bwd_reduce_g(rows, c, dh, ldd, arg, k, y, ldy, mean, invstd, gamma, beta, relu, sums, nullptr, 0, stream);
}

extern "C" int pn2x_bn_bwd_reduce_g(long rows, int c, const float *dh, int ldd, const int *arg, int k, const float *y, int ldy,
                                    const float *mean, const float *invstd, const float *gamma, const float *beta, int relu, double *sums,
                                    float *g_out, int ldg, void *stream) {
    using namespace pn2;
    if (rows < 1 || bad_c(c) || ldy < c || ldy % 4 || ldd < c || ldd % 4 || (arg && (k < 1 || rows % k))) return PN2_EINVAL;
    if (!dh || !y || !mean || !invstd || !gamma || !beta || !sums) return PN2_ENULL;
    if (((uintptr_t)y | (uintptr_t)dh | (uintptr_t)arg | (uintptr_t)g_out) % 16) return PN2_EINVAL;
    if (g_out && (ldg < c || ldg % 4)) return PN2_EINVAL;
    const int rpb = rows_per_block_for(rows, c, rows * c > (2L << 20));
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(kTT), 0, (hipStream_t)stream, rows, c, dh,
                       ldd, y, ldy, mean, invstd, gamma, beta, rpb, arg ? 1 : relu, sums, arg, arg ? k : 1, arg ? g_out : nullptr, ldg);
    return check_launch();
}

namespace pn2 {
// The BatchNorm-backward sums of a max-pooled top layer from the arg-max rows ALONE: the routed gradient is non-zero in one row
// per (group, channel), so sum(g) and sum(g xhat) need groups x C gathered pre-activations, not the rows x C tensor.
struct RoutedArgs {
    long groups; int K, C; const float *dout; int ldd; const int *arg; int lda; const float *y; int ldy;
    const float *mean, *invstd, *gamma, *beta; long groups_per_block; double *sums;
};
__device__ __forceinline__ void bn_bwd_reduce_routed_body(const RoutedArgs &A, unsigned bx, unsigned by) {
    const long groups = A.groups, groups_per_block = A.groups_per_block;
    const int K = A.K, C = A.C, ldd = A.ldd, lda = A.lda, ldy = A.ldy;
    const float *__restrict__ dout = A.dout, *__restrict__ y = A.y, *__restrict__ mean = A.mean, *__restrict__ invstd = A.invstd;
    const float *__restrict__ gamma = A.gamma, *__restrict__ beta = A.beta;
    const int *__restrict__ arg = A.arg;
    double *__restrict__ sums = A.sums;
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = (int)by * 64 + cl;
    const long g0 = (long)bx * groups_per_block;
    const long g1 = (g0 + groups_per_block) < groups ? (g0 + groups_per_block) : groups;
    float s = 0.f, q = 0.f;
    if (c < C) {
        const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
        for (long g = g0 + gl; g < g1; g += 4) {
            const int a = arg[g * lda + c];
            const float xhat = (y[(g * K + a) * (long)ldy + c] - m) * is;
            const float gg = (xhat * ga + be > 0.f) ? dout[g * ldd + c] : 0.f;  // [relu(BN(y)) > 0], torch's evaluation order
            s += gg;
            q += gg * xhat;
        }
    }
    red[0][gl][cl] = s;
    red[1][gl][cl] = q;
    __syncthreads();
    if (gl == 0 && c < C) {
        const double sd = (double)red[0][0][cl] + (double)red[0][1][cl] + (double)red[0][2][cl] + (double)red[0][3][cl];
        const double qd = (double)red[1][0][cl] + (double)red[1][1][cl] + (double)red[1][2][cl] + (double)red[1][3][cl];
        double *dst = sums + (size_t)(bx % kBnRep) * 2 * C;
        unsafeAtomicAdd(dst + c, sd);
        unsafeAtomicAdd(dst + C + c, qd);
    }
}
__global__ void __launch_bounds__(kTT)
bn_bwd_reduce_routed_kernel(RoutedArgs a) { bn_bwd_reduce_routed_body(a, blockIdx.x, blockIdx.y); }
// two problems with the same channel count (the two neighbourhood sizes of a module) in one launch
__global__ void __launch_bounds__(kTT)
bn_bwd_reduce_routed_pair_kernel(RoutedArgs a, RoutedArgs b, unsigned na) {
    if (blockIdx.x < na) bn_bwd_reduce_routed_body(a, blockIdx.x, blockIdx.y);
    else bn_bwd_reduce_routed_body(b, blockIdx.x - na, blockIdx.y);
}
}  // namespace pn2

namespace pn2 {
// out (C x 3) = dy^T (C x rows) rel (rows x 3): the gradient of the xyz columns of a grouped layer-1 weight (a GEMM with a
// 3-wide output and a 100k+ deep reduction: the library's split-K kernels take 27-49 us for what is one pass over dy).
// Stage 1: per-workgroup partial sums (no atomics: a thousand workgroups on the same 3 C addresses serialise in L2);
// stage 2: one workgroup sums the partials.
__global__ void __launch_bounds__(kTT)
rows_outer3_kernel(long rows, int C, const float *__restrict__ dy, int ldy, const float *__restrict__ rel, long rows_per_block,
                   float *__restrict__ partial) {
    __shared__ float red[12][kTT];
    const int Q = C >> 2, q = threadIdx.x % Q, rl = threadIdx.x / Q, lanes = kTT / Q;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = (r0 + rows_per_block) < rows ? (r0 + rows_per_block) : rows;
    float a[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    auto add = [&](float4 d, const float *e) {
        const float e0 = e[0], e1 = e[1], e2 = e[2];
        a[0][0] += d.x * e0; a[0][1] += d.x * e1; a[0][2] += d.x * e2;
        a[1][0] += d.y * e0; a[1][1] += d.y * e1; a[1][2] += d.y * e2;
        a[2][0] += d.z * e0; a[2][1] += d.z * e1; a[2][2] += d.z * e2;
        a[3][0] += d.w * e0; a[3][1] += d.w * e1; a[3][2] += d.w * e2;
    };
    long r = r0 + rl;
    for (; r + 3L * lanes < r1; r += 4L * lanes) {  // four independent 16-byte loads in flight per thread
        const float *p = dy + r * ldy + 4 * q;
        const float4 d0 = *reinterpret_cast<const float4 *>(p), d1 = *reinterpret_cast<const float4 *>(p + (long)lanes * ldy);
        const float4 d2 = *reinterpret_cast<const float4 *>(p + 2L * lanes * ldy), d3 = *reinterpret_cast<const float4 *>(p + 3L * lanes * ldy);
        add(d0, rel + r * 3); add(d1, rel + (r + lanes) * 3); add(d2, rel + (r + 2L * lanes) * 3); add(d3, rel + (r + 3L * lanes) * 3);
    }
    for (; r < r1; r += lanes) add(*reinterpret_cast<const float4 *>(dy + r * ldy + 4 * q), rel + r * 3);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 3; ++t) red[e * 3 + t][threadIdx.x] = a[e][t];
    __syncthreads();
    if (rl == 0) {
        float *o = partial + (size_t)blockIdx.x * 3 * C + 12 * q;  // out[c][t], c = 4 q + e
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            float s = red[k][q];
            for (int l = 1; l < lanes; ++l) s += red[k][l * Q + q];
            o[k] = s;
        }
    }
}

__device__ __forceinline__ void rows_outer3_sum_body(int blocks, int n, const float *__restrict__ partial, float *__restrict__ out, unsigned bx) {
    __shared__ float red[16][16];
    // one workgroup per 16 outputs: thread = (output, one of sixteen lanes over the partials), four loads in flight per lane
    const int el = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int e = bx * 16 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int p = pl;
        for (; p + 48 < blocks; p += 64) {
            s0 += partial[(size_t)p * n + e];
            s1 += partial[(size_t)(p + 16) * n + e];
            s2 += partial[(size_t)(p + 32) * n + e];
            s3 += partial[(size_t)(p + 48) * n + e];
        }
        for (; p < blocks; p += 16) s0 += partial[(size_t)p * n + e];
    }
    red[pl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (pl == 0 && e < n) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) s += red[l][el];
        out[e] = s;
    }
}
__global__ void __launch_bounds__(kTT)
rows_outer3_sum_kernel(int blocks, int n, const float *__restrict__ partial, float *__restrict__ out) {
    rows_outer3_sum_body(blocks, n, partial, out, blockIdx.x);
}
__global__ void __launch_bounds__(kTT)
rows_outer3_sum_pair_kernel(int blocks_a, int n_a, const float *__restrict__ partial_a, float *__restrict__ out_a, int blocks_b, int n_b,
                            const float *__restrict__ partial_b, float *__restrict__ out_b, unsigned na) {
    if (blockIdx.x < na) rows_outer3_sum_body(blocks_a, n_a, partial_a, out_a, blockIdx.x);
    else rows_outer3_sum_body(blocks_b, n_b, partial_b, out_b, blockIdx.x - na);
}
}  // namespace pn2

extern "C" long pn2x_rows_outer3_scratch_floats(long rows, int c) {
    if (rows < 1 || c < 1) return -1;
    return 512L * 3 * c;
}

extern "C" int pn2x_rows_outer3(long rows, int c, const float *dy, int ldy, const float *rel, float *out, float *scratch,
                                long scratch_floats, void *stream) {
    using namespace pn2;
    if (rows < 1 || c < 4 || c % 4 || kTT % (c / 4) || ldy < c || ldy % 4) return PN2_EINVAL;
    if (!dy || !rel || !out || !scratch) return PN2_ENULL;
    if ((uintptr_t)dy % 16) return PN2_EINVAL;
    if (scratch_floats < 512L * 3 * c) return PN2_ESCRATCH;
    long blocks = 512;
    const int lanes = kTT / (c / 4);
    long rpb = (rows + blocks - 1) / blocks;
    rpb = (rpb + 4L * lanes - 1) / (4L * lanes) * (4L * lanes);
    blocks = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(rows_outer3_kernel, dim3((unsigned)blocks), dim3(kTT), 0, (hipStream_t)stream, rows, c, dy, ldy, rel, rpb, scratch);
    hipLaunchKernelGGL(rows_outer3_sum_kernel, dim3((3 * c + 15) / 16), dim3(kTT), 0, (hipStream_t)stream, (int)blocks, 3 * c, scratch, out);
    return check_launch();
}

namespace pn2 {
static int routed_args(long groups, int k, int c, const float *dout, int ldd, const int *arg, int lda, const float *y, int ldy,
                       const float *mean, const float *invstd, const float *gamma, const float *beta, double *sums, RoutedArgs &a,
                       unsigned &blocks_out) {
    if (groups < 1 || k < 1 || bad_c(c) || ldy < c || ldd < c || lda < c || groups * k > 0x7fffffffL) return PN2_EINVAL;
    if (!dout || !arg || !y || !mean || !invstd || !gamma || !beta || !sums) return PN2_ENULL;
    const int ny = (c + 63) / 64;
    long blocks = (4L * num_compute_units() + ny - 1) / ny;
    long gpb = (groups + blocks - 1) / blocks;
    gpb = (gpb + 3) / 4 * 4;
    if (gpb < 16) gpb = 16;
    blocks = (groups + gpb - 1) / gpb;
    blocks_out = (unsigned)blocks;
    a = RoutedArgs{groups, k, c, dout, ldd, arg, lda, y, ldy, mean, invstd, gamma, beta, gpb, sums};
    return PN2_OK;
}
}  // namespace pn2

extern "C" int pn2x_bn_bwd_reduce_routed(long groups, int k, int c, const float *dout, int ldd, const int *arg, int lda, const float *y, int ldy,
                                         const float *mean, const float *invstd, const float *gamma, const float *beta, double *sums,
                                         void *stream) {
    using namespace pn2;
    RoutedArgs a;
    unsigned blocks = 0;
    if (int rc = routed_args(groups, k, c, dout, ldd, arg, lda, y, ldy, mean, invstd, gamma, beta, sums, a, blocks)) return rc;
    hipLaunchKernelGGL(bn_bwd_reduce_routed_kernel, dim3(blocks, (c + 63) / 64), dim3(kTT), 0, (hipStream_t)stream, a);
    return check_launch();
}

// two problems of the same channel count in one launch (two calls when the counts differ); same results
extern "C" int pn2x_bn_bwd_reduce_routed_pair(long groups_a, int k_a, int c_a, const float *dout_a, int ldd_a, const int *arg_a, int lda_a,
                                              const float *y_a, int ldy_a, const float *mean_a, const float *invstd_a, const float *gamma_a,
                                              const float *beta_a, double *sums_a, long groups_b, int k_b, int c_b, const float *dout_b,
                                              int ldd_b, const int *arg_b, int lda_b, const float *y_b, int ldy_b, const float *mean_b,
                                              const float *invstd_b, const float *gamma_b, const float *beta_b, double *sums_b, void *stream) {
    using namespace pn2;
    RoutedArgs a, b;
    unsigned na = 0, nb = 0;
    if (int rc = routed_args(groups_a, k_a, c_a, dout_a, ldd_a, arg_a, lda_a, y_a, ldy_a, mean_a, invstd_a, gamma_a, beta_a, sums_a, a, na)) return rc;
    if (int rc = routed_args(groups_b, k_b, c_b, dout_b, ldd_b, arg_b, lda_b, y_b, ldy_b, mean_b, invstd_b, gamma_b, beta_b, sums_b, b, nb)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if ((c_a + 63) / 64 != (c_b + 63) / 64) {
        hipLaunchKernelGGL(bn_bwd_reduce_routed_kernel, dim3(na, (c_a + 63) / 64), dim3(kTT), 0, st, a);
        hipLaunchKernelGGL(bn_bwd_reduce_routed_kernel, dim3(nb, (c_b + 63) / 64), dim3(kTT), 0, st, b);
        return check_launch();
    }
    hipLaunchKernelGGL(bn_bwd_reduce_routed_pair_kernel, dim3(na + nb, (c_a + 63) / 64), dim3(kTT), 0, st, a, b, na);
    return check_launch();
}

extern "C" int pn2x_bn_bwd_apply(long rows, int c, const float *g, int ldg, const float *y, int ldy, const float *mean,
                                 const float *invstd, const float *gamma, const float *beta, int relu, const double *sums, float *dy,
                                 int ldo, float *dgamma, float *dbeta, float *dbias, void *stream) {
    using namespace pn2;
    if (rows < 1 || bad_c(c) || ldy < c || ldy % 4 || ldg < c || ldg % 4 || ldo < c || ldo % 4) return PN2_EINVAL;
    if (!g || !y || !mean || !invstd || !gamma || !beta || !sums || !dy || !dgamma || !dbeta) return PN2_ENULL;
    if (((uintptr_t)y | (uintptr_t)g | (uintptr_t)dy) % 16) return PN2_EINVAL;
    const int rpb = rows_per_block_for(rows, c);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(kTT), 0, (hipStream_t)stream, rows, c, g, ldg,
                       y, ldy, mean, invstd, gamma, beta, sums, rpb, relu, dy, ldo, dgamma, dbeta, dbias, (const int *)nullptr, 1);
    return check_launch();
}

// pn2x_bn_bwd_apply that also returns dwx (c x 3) = dy^T rel (rel (rows x 3)); scratch: pn2x_bn_bwd_apply_rel_scratch_floats(rows, c)
extern "C" long pn2x_bn_bwd_apply_rel_scratch_floats(long rows, int c) {
    using namespace pn2;
    if (rows < 1 || bad_c(c)) return -1;
    const int rpb = rows_per_block_for(rows, c);
    return ((rows + rpb - 1) / rpb) * 3L * c;
}

namespace pn2 {
static int apply_rel_args(long rows, int c, const float *g, int ldg, const float *y, int ldy, const float *mean, const float *invstd,
                          const float *gamma, const float *beta, int relu, const double *sums, float *dy, int ldo, float *dgamma,
                          float *dbeta, float *dbias, const float *rel, float *scratch, long scratch_floats, float *dwx, ApplyRelArgs &a,
                          long &blocks) {
    if (rows < 1 || bad_c(c) || c > 256 || ldy < c || ldy % 4 || ldg < c || ldg % 4 || ldo < c || ldo % 4) return PN2_EINVAL;
    if (!g || !y || !mean || !invstd || !gamma || !beta || !sums || !dy || !dgamma || !dbeta || !rel || !scratch || !dwx) return PN2_ENULL;
    if (((uintptr_t)y | (uintptr_t)g | (uintptr_t)dy) % 16) return PN2_EINVAL;
    const int rpb = rows_per_block_for(rows, c);
    blocks = (rows + rpb - 1) / rpb;
    if (scratch_floats < blocks * 3L * c) return PN2_ESCRATCH;
    a = ApplyRelArgs{rows, c, g, ldg, y, ldy, mean, invstd, gamma, beta, sums, rpb, relu, dy, ldo, dgamma, dbeta, dbias, rel, scratch};
    return PN2_OK;
}
}  // namespace pn2

extern "C" int pn2x_bn_bwd_apply_rel(long rows, int c, const float *g, int ldg, const float *y, int ldy, const float *mean,
                                     const float *invstd, const float *gamma, const float *beta, int relu, const double *sums, float *dy,
                                     int ldo, float *dgamma, float *dbeta, float *dbias, const float *rel, float *scratch,
                                     long scratch_floats, float *dwx, void *stream) {
    using namespace pn2;
    ApplyRelArgs a;
    long blocks = 0;
    if (int rc = apply_rel_args(rows, c, g, ldg, y, ldy, mean, invstd, gamma, beta, relu, sums, dy, ldo, dgamma, dbeta, dbias, rel, scratch,
                                scratch_floats, dwx, a, blocks)) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_relu_bwd_apply_rel_kernel, dim3((unsigned)blocks), dim3(kTT), 0, st, a);
    hipLaunchKernelGGL(rows_outer3_sum_kernel, dim3((3 * c + 15) / 16), dim3(kTT), 0, st, (int)blocks, 3 * c, scratch, dwx);
    return check_launch();
}

// Two problems (the two neighbourhood sizes of a keypoint-query module: pointnet_utils.py:566-581) in ONE launch each: the K = 16
// scale alone is a 16 us launch for a quarter of the K = 64 scale's rows.
extern "C" int pn2x_bn_bwd_apply_rel_pair(long rows_a, int c_a, const float *g_a, int ldg_a, const float *y_a, int ldy_a, const float *mean_a,
                                          const float *invstd_a, const float *gamma_a, const float *beta_a, int relu_a, const double *sums_a,
                                          float *dy_a, int ldo_a, float *dgamma_a, float *dbeta_a, float *dbias_a, const float *rel_a,
                                          float *scratch_a, long scratch_floats_a, float *dwx_a, long rows_b, int c_b, const float *g_b,
                                          int ldg_b, const float *y_b, int ldy_b, const float *mean_b, const float *invstd_b,
                                          const float *gamma_b, const float *beta_b, int relu_b, const double *sums_b, float *dy_b, int ldo_b,
                                          float *dgamma_b, float *dbeta_b, float *dbias_b, const float *rel_b, float *scratch_b,
                                          long scratch_floats_b, float *dwx_b, void *stream) {
    using namespace pn2;
    ApplyRelArgs a, b;
    long na = 0, nb = 0;
    if (int rc = apply_rel_args(rows_a, c_a, g_a, ldg_a, y_a, ldy_a, mean_a, invstd_a, gamma_a, beta_a, relu_a, sums_a, dy_a, ldo_a, dgamma_a,
                                dbeta_a, dbias_a, rel_a, scratch_a, scratch_floats_a, dwx_a, a, na)) return rc;
    if (int rc = apply_rel_args(rows_b, c_b, g_b, ldg_b, y_b, ldy_b, mean_b, invstd_b, gamma_b, beta_b, relu_b, sums_b, dy_b, ldo_b, dgamma_b,
                                dbeta_b, dbias_b, rel_b, scratch_b, scratch_floats_b, dwx_b, b, nb)) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_relu_bwd_apply_rel_pair_kernel, dim3((unsigned)(na + nb)), dim3(kTT), 0, st, a, b, (unsigned)na);
    const unsigned sa = (3 * c_a + 15) / 16, sb = (3 * c_b + 15) / 16;
    hipLaunchKernelGGL(rows_outer3_sum_pair_kernel, dim3(sa + sb), dim3(kTT), 0, st, (int)na, 3 * c_a, scratch_a, dwx_a, (int)nb, 3 * c_b, scratch_b,
                       dwx_b, sa);
    return check_launch();
}

extern "C" int pn2x_bn_relu_max(long groups, int k, int c, const float *y, int ldy, const double *sums, const float *gamma,
                                const float *beta, const float *conv_bias, float eps, float momentum, float *running_mean,
                                float *running_var, long long *num_batches_tracked, float *save_mean, float *save_invstd, float *out,
                                int *arg, void *stream) {
    return pn2x_bn_relu_max_ld(groups, k, c, y, ldy, sums, gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
                               num_batches_tracked, save_mean, save_invstd, out, c, arg, stream);
}

// ... with the output rows ldo floats apart: `out` may be a column block of a wider (groups x ldo) buffer -- the two scales of a
// module write the two halves of ONE tensor instead of being concatenated by a copy launch (round 6)
extern "C" int pn2x_bn_relu_max_ld(long groups, int k, int c, const float *y, int ldy, const double *sums, const float *gamma,
                                   const float *beta, const float *conv_bias, float eps, float momentum, float *running_mean,
                                   float *running_var, long long *num_batches_tracked, float *save_mean, float *save_invstd, float *out,
                                   int ldo, int *arg, void *stream) {
    using namespace pn2;
    if (groups < 1 || k < 1 || bad_c(c) || ldy < c || ldy % 4 || ldo < c || ldo % 4) return PN2_EINVAL;
    if (!y || !sums || !gamma || !beta || !save_mean || !save_invstd || !out || !arg) return PN2_ENULL;
    if (((uintptr_t)y | (uintptr_t)out | (uintptr_t)arg) % 16) return PN2_EINVAL;
    const long items = groups * (c / 4);
    int split = 1;  // rows of a group over `split` lanes while that keeps fewer than ~128k threads busy
    while (split < 64 && 2 * split <= k && items * split < 131072) split *= 2;
    if (split >= 4) {
        const int ipb = kTT / split;
        const MaxSplitArgs a{groups, k, c, split, y, ldy, sums, gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
                             num_batches_tracked, save_mean, save_invstd, out, arg, ldo};
        hipLaunchKernelGGL(bn_relu_max_split_kernel, dim3((unsigned)((items + ipb - 1) / ipb)), dim3(kTT), 0, (hipStream_t)stream, a);
        return check_launch();
    }
    hipLaunchKernelGGL(bn_relu_max_kernel, dim3((unsigned)((items + kTT - 1) / kTT)), dim3(kTT), 0, (hipStream_t)stream, groups, k, c, y, ldy,
                       sums, gamma, beta, conv_bias, eps, momentum, running_mean, running_var, num_batches_tracked, save_mean, save_invstd,
                       out, ldo, arg);
    return check_launch();
}

// pn2x_bn_relu_max for two problems (the two neighbourhood sizes of a module) in ONE launch where both take the split kernel (few
// groups: rows of a group over several lanes); two calls otherwise.  Same results.
extern "C" int pn2x_bn_relu_max_pair(long groups_a, int k_a, int c_a, const float *y_a, int ldy_a, const double *sums_a, const float *gamma_a,
                                     const float *beta_a, const float *conv_bias_a, float eps_a, float momentum_a, float *running_mean_a,
                                     float *running_var_a, long long *nbt_a, float *save_mean_a, float *save_invstd_a, float *out_a, int *arg_a,
                                     long groups_b, int k_b, int c_b, const float *y_b, int ldy_b, const double *sums_b, const float *gamma_b,
                                     const float *beta_b, const float *conv_bias_b, float eps_b, float momentum_b, float *running_mean_b,
                                     float *running_var_b, long long *nbt_b, float *save_mean_b, float *save_invstd_b, float *out_b, int *arg_b,
                                     void *stream) {
    return pn2x_bn_relu_max_pair_ld(groups_a, k_a, c_a, y_a, ldy_a, sums_a, gamma_a, beta_a, conv_bias_a, eps_a, momentum_a, running_mean_a,
                                    running_var_a, nbt_a, save_mean_a, save_invstd_a, out_a, c_a, arg_a, groups_b, k_b, c_b, y_b, ldy_b, sums_b,
                                    gamma_b, beta_b, conv_bias_b, eps_b, momentum_b, running_mean_b, running_var_b, nbt_b, save_mean_b,
                                    save_invstd_b, out_b, c_b, arg_b, stream);
}

extern "C" int pn2x_bn_relu_max_pair_ld(long groups_a, int k_a, int c_a, const float *y_a, int ldy_a, const double *sums_a, const float *gamma_a,
                                        const float *beta_a, const float *conv_bias_a, float eps_a, float momentum_a, float *running_mean_a,
                                        float *running_var_a, long long *nbt_a, float *save_mean_a, float *save_invstd_a, float *out_a, int ldo_a,
                                        int *arg_a, long groups_b, int k_b, int c_b, const float *y_b, int ldy_b, const double *sums_b,
                                        const float *gamma_b, const float *beta_b, const float *conv_bias_b, float eps_b, float momentum_b,
                                        float *running_mean_b, float *running_var_b, long long *nbt_b, float *save_mean_b, float *save_invstd_b,
                                        float *out_b, int ldo_b, int *arg_b, void *stream) {
    using namespace pn2;
    auto split_of = [](long groups, int k, int c) {
        const long items = groups * (c / 4);
        int split = 1;
        while (split < 64 && 2 * split <= k && items * split < 131072) split *= 2;
        return split;
    };
    const bool ok_a = groups_a >= 1 && k_a >= 1 && !bad_c(c_a) && ldy_a >= c_a && ldy_a % 4 == 0 && y_a && sums_a && gamma_a && beta_a &&
                      save_mean_a && save_invstd_a && out_a && arg_a && (((uintptr_t)y_a | (uintptr_t)out_a | (uintptr_t)arg_a) % 16) == 0 && ldo_a >= c_a && ldo_a % 4 == 0;
    const bool ok_b = groups_b >= 1 && k_b >= 1 && !bad_c(c_b) && ldy_b >= c_b && ldy_b % 4 == 0 && y_b && sums_b && gamma_b && beta_b &&
                      save_mean_b && save_invstd_b && out_b && arg_b && (((uintptr_t)y_b | (uintptr_t)out_b | (uintptr_t)arg_b) % 16) == 0 && ldo_b >= c_b && ldo_b % 4 == 0;
    const int sa = ok_a ? split_of(groups_a, k_a, c_a) : 1, sb = ok_b ? split_of(groups_b, k_b, c_b) : 1;
    if (!ok_a || !ok_b || sa < 4 || sb < 4) {
        const int rc = pn2x_bn_relu_max_ld(groups_a, k_a, c_a, y_a, ldy_a, sums_a, gamma_a, beta_a, conv_bias_a, eps_a, momentum_a, running_mean_a,
                                           running_var_a, nbt_a, save_mean_a, save_invstd_a, out_a, ldo_a, arg_a, stream);
        if (rc != PN2_OK) return rc;
        return pn2x_bn_relu_max_ld(groups_b, k_b, c_b, y_b, ldy_b, sums_b, gamma_b, beta_b, conv_bias_b, eps_b, momentum_b, running_mean_b,
                                   running_var_b, nbt_b, save_mean_b, save_invstd_b, out_b, ldo_b, arg_b, stream);
    }
    const MaxSplitArgs a{groups_a, k_a, c_a, sa, y_a, ldy_a, sums_a, gamma_a, beta_a, conv_bias_a, eps_a, momentum_a, running_mean_a,
                         running_var_a, nbt_a, save_mean_a, save_invstd_a, out_a, arg_a, ldo_a};
    const MaxSplitArgs b{groups_b, k_b, c_b, sb, y_b, ldy_b, sums_b, gamma_b, beta_b, conv_bias_b, eps_b, momentum_b, running_mean_b,
                         running_var_b, nbt_b, save_mean_b, save_invstd_b, out_b, arg_b, ldo_b};
    const long ia = groups_a * (c_a / 4), ib = groups_b * (c_b / 4);
    const unsigned na = (unsigned)((ia + kTT / sa - 1) / (kTT / sa)), nb = (unsigned)((ib + kTT / sb - 1) / (kTT / sb));
    hipLaunchKernelGGL(bn_relu_max_split_pair_kernel, dim3(na + nb), dim3(kTT), 0, (hipStream_t)stream, a, b, na);
    return check_launch();
}

extern "C" int pn2x_bn_relu_max_bwd(long groups, int k, int c, const float *dout, const int *arg, const float *y, int ldy,
                                    const float *mean, const float *invstd, const float *gamma, const float *beta, double *sums, float *dy,
                                    int ldo, float *dgamma, float *dbeta, float *dbias, void *stream) {
    using namespace pn2;
    if (groups < 1 || k < 1 || bad_c(c) || ldy < c || ldy % 4 || ldo < c || ldo % 4) return PN2_EINVAL;
    if (!dout || !arg || !y || !mean || !invstd || !gamma || !beta || !sums || !dy || !dgamma || !dbeta) return PN2_ENULL;
    if (((uintptr_t)y | (uintptr_t)dout | (uintptr_t)dy | (uintptr_t)arg) % 16) return PN2_EINVAL;
    const long rows = groups * k;
    const int rpb = rows_per_block_for(rows, c);
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, grid, dim3(kTT), 0, (hipStream_t)stream, rows, c, dout, c, y, ldy, mean, invstd, gamma, beta,
                       rpb, 1, sums, arg, k);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, grid, dim3(kTT), 0, (hipStream_t)stream, rows, c, dout, c, y, ldy, mean, invstd, gamma, beta,
                       sums, rpb, 1, dy, ldo, dgamma, dbeta, dbias, arg, k);
    return check_launch();
}

extern "C" int pn2x_scatter_add_rows(int b, int n, int m, int c, const float *dout, int ldo, const int *idx, float *din, int ldi,
                                     void *stream) {
    using namespace pn2;
    if (b < 0 || n < 1 || m < 0 || c < 0 || c % 4 || ldo < c || ldo % 4 || ldi < c || ldi % 4) return PN2_EINVAL;
    if (b == 0 || m == 0 || c == 0) return PN2_OK;
    if (!dout || !idx || !din) return PN2_ENULL;
    if (((uintptr_t)dout | (uintptr_t)din) % 16) return PN2_EINVAL;
    if (const int cc = slab_channels(b, n, c)) {
        hipLaunchKernelGGL(scatter_rows_lds_kernel<false>, dim3((c + cc - 1) / cc, b), dim3(kTT), (size_t)n * cc * sizeof(float), (hipStream_t)stream,
                           n, m, c, cc, dout, ldo, idx, nullptr, din, ldi);
        return check_launch();
    }
    const int Q = c / 4;
    const long total = (long)m * Q;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((total + kTT - 1) / kTT), b), dim3(kTT), 0, (hipStream_t)stream, n, m, Q,
                       dout, ldo, idx, din, ldi);
    return check_launch();
}

extern "C" int pn2x_three_interpolate_pm_grad(int b, int c, int m, int n, const float *dout, int ldo, const int *idx,
                                              const float *weight, float *dpoints, int ldp, void *stream) {
    using namespace pn2;
    if (b < 0 || m < 1 || n < 0 || c < 0 || c % 4 || ldo < c || ldo % 4 || ldp < c || ldp % 4) return PN2_EINVAL;
    if (b == 0 || n == 0 || c == 0) return PN2_OK;
    if (!dout || !idx || !weight || !dpoints) return PN2_ENULL;
    if (((uintptr_t)dout | (uintptr_t)dpoints) % 16) return PN2_EINVAL;
    if (const int cc = slab_channels(b, m, c)) {
        hipLaunchKernelGGL(scatter_rows_lds_kernel<true>, dim3((c + cc - 1) / cc, b), dim3(kTT), (size_t)m * cc * sizeof(float), (hipStream_t)stream,
                           m, n, c, cc, dout, ldo, idx, weight, dpoints, ldp);
        return check_launch();
    }
    const int Q = c / 4;
    const long total = (long)n * Q;
    hipLaunchKernelGGL(interp_pm_bwd_kernel, dim3((unsigned)((total + kTT - 1) / kTT), b), dim3(kTT), 0, (hipStream_t)stream, m, n, Q, dout,
                       ldo, idx, weight, dpoints, ldp);
    return check_launch();
}

extern "C" int pn2x_sa_layer1(int b, int n, int s, int k, int c1, const float *a1f, int a1f_ld, const float *xyz, const float *cxyz,
                              const float *wx, const float *cadd, int cadd_ld, const int *idx, float *out, float *rel_out,
                              void *stream) {
    return pn2x_sa_layer1_ld(b, n, s, k, c1, a1f, a1f_ld, xyz, cxyz, wx, 3, cadd, cadd_ld, idx, out, rel_out, stream);
}

extern "C" int pn2x_sa_layer1_ld(int b, int n, int s, int k, int c1, const float *a1f, int a1f_ld, const float *xyz, const float *cxyz,
                                 const float *wx, int wx_ld, const float *cadd, int cadd_ld, const int *idx, float *out, float *rel_out,
                                 void *stream) {
    using namespace pn2;
    if (b < 0 || n < 1 || s < 0 || k < 1 || c1 < 4 || c1 % 4 || (xyz && wx_ld < 3)) return PN2_EINVAL;
    if (b == 0 || s == 0) return PN2_OK;
    if (!idx || !out || (!a1f && !xyz)) return PN2_ENULL;
    if (xyz && (!cxyz || !wx)) return PN2_ENULL;
    if ((a1f && (a1f_ld < c1 || a1f_ld % 4)) || (cadd && (cadd_ld < c1 || cadd_ld % 4))) return PN2_EINVAL;
    if (((uintptr_t)a1f | (uintptr_t)cadd | (uintptr_t)out) % 16) return PN2_EINVAL;
    const int Q = c1 / 4;
    const long total = (long)s * k * Q;
    hipLaunchKernelGGL(sa_layer1_kernel, dim3((unsigned)((total + kTT - 1) / kTT), b), dim3(kTT), 0, (hipStream_t)stream, n, s, k, Q, a1f,
                       a1f_ld, xyz, cxyz, wx, wx_ld, cadd, cadd_ld, idx, out, rel_out);
    return check_launch();
}

// pn2x_sa_layer1_ld that also accumulates the BatchNorm statistics of its output into `sums` (pn2x_bn_sums_doubles(c1) doubles, zeroed
// by the caller; the layout pn2x_bn_stats writes): in the same launch when the channel quads divide the workgroup, by a
// pn2x_bn_stats launch behind it otherwise.
namespace pn2 {
// 1: the statistics cannot be taken in the launch (channel quads do not divide the workgroup): the caller runs the two-launch form
static int sa_layer1_stats_args(int b, int n, int s, int k, int c1, const float *a1f, int a1f_ld, const float *xyz, const float *cxyz,
                                const float *wx, int wx_ld, const float *cadd, int cadd_ld, const int *idx, float *out, float *rel_out,
                                double *sums, SaLayer1Args &a, unsigned &nbx) {
    if (!sums) return PN2_ENULL;
    const int Q = c1 >= 4 ? c1 / 4 : 1;
    if (c1 % 4 || c1 < 4 || kTT % Q || (long)b * s * k > 2147483647L) return 1;
    if (b < 0 || n < 1 || s < 0 || k < 1 || (xyz && wx_ld < 3)) return PN2_EINVAL;
    if (b == 0 || s == 0) { nbx = 0; return PN2_OK; }
    if (!idx || !out || (!a1f && !xyz)) return PN2_ENULL;
    if (xyz && (!cxyz || !wx)) return PN2_ENULL;
    if ((a1f && (a1f_ld < c1 || a1f_ld % 4)) || (cadd && (cadd_ld < c1 || cadd_ld % 4))) return PN2_EINVAL;
    if (((uintptr_t)a1f | (uintptr_t)cadd | (uintptr_t)out) % 16) return PN2_EINVAL;
    const int rpp = kTT / Q, sk = s * k;
    constexpr int target_wgs = 2048;  // (a sweep from 128 to 2048 workgroups moved the kernel by < 5 % between 512 and 2048)
    int rpb = (int)(((long)sk * b + target_wgs - 1) / target_wgs);  // ~2048 workgroups on a large problem, at least 8 rows per thread
    if (rpb < 8 * rpp) rpb = 8 * rpp;
    rpb = (rpb + rpp - 1) / rpp * rpp;
    nbx = (unsigned)((sk + rpb - 1) / rpb);
    a = SaLayer1Args{n, s, k, Q, a1f, a1f_ld, xyz, cxyz, wx, wx_ld, cadd, cadd_ld, idx, out, rel_out, rpb, sums};
    return PN2_OK;
}
}  // namespace pn2

// pn2x_sa_layer1_ld that also accumulates the BatchNorm statistics of its output into `sums` (pn2x_bn_sums_doubles(c1) doubles, zeroed
// by the caller; the layout pn2x_bn_stats writes): in the same launch when the channel quads divide the workgroup, by a
// pn2x_bn_stats launch behind it otherwise.
extern "C" int pn2x_sa_layer1_stats(int b, int n, int s, int k, int c1, const float *a1f, int a1f_ld, const float *xyz, const float *cxyz,
                                    const float *wx, int wx_ld, const float *cadd, int cadd_ld, const int *idx, float *out, float *rel_out,
                                    double *sums, void *stream) {
    using namespace pn2;
    SaLayer1Args a;
    unsigned nbx = 0;
    const int rc0 = sa_layer1_stats_args(b, n, s, k, c1, a1f, a1f_ld, xyz, cxyz, wx, wx_ld, cadd, cadd_ld, idx, out, rel_out, sums, a, nbx);
    if (rc0 == 1) {
        const int rc = pn2x_sa_layer1_ld(b, n, s, k, c1, a1f, a1f_ld, xyz, cxyz, wx, wx_ld, cadd, cadd_ld, idx, out, rel_out, stream);
        if (rc != PN2_OK || b == 0 || s == 0) return rc;
        return pn2x_bn_stats((long)b * s * k, c1, out, c1, sums, stream);
    }
    if (rc0 != PN2_OK || nbx == 0) return rc0;
    hipLaunchKernelGGL(sa_layer1_stats_kernel, dim3(nbx, b), dim3(kTT), 0, (hipStream_t)stream, a);
    return check_launch();
}

// ... for the two neighbourhood sizes of one module (same clouds, coordinates and centres; own neighbour lists, weights, outputs and
// statistics) in ONE launch; falls back to two pn2x_sa_layer1_stats calls where a scale cannot take its statistics in the launch.
extern "C" int pn2x_sa_layer1_stats_pair(int b, int n, int s, const float *xyz, const float *cxyz, int k_a, int c1_a, const float *a1f_a,
                                         int a1f_ld_a, const float *wx_a, int wx_ld_a, const float *cadd_a, int cadd_ld_a, const int *idx_a,
                                         float *out_a, float *rel_out_a, double *sums_a, int k_b, int c1_b, const float *a1f_b, int a1f_ld_b,
                                         const float *wx_b, int wx_ld_b, const float *cadd_b, int cadd_ld_b, const int *idx_b, float *out_b,
                                         float *rel_out_b, double *sums_b, void *stream) {
    using namespace pn2;
    SaLayer1Args a, c;
    unsigned na = 0, nb = 0;
    const int ra = sa_layer1_stats_args(b, n, s, k_a, c1_a, a1f_a, a1f_ld_a, xyz, cxyz, wx_a, wx_ld_a, cadd_a, cadd_ld_a, idx_a, out_a, rel_out_a,
                                        sums_a, a, na);
    const int rb = sa_layer1_stats_args(b, n, s, k_b, c1_b, a1f_b, a1f_ld_b, xyz, cxyz, wx_b, wx_ld_b, cadd_b, cadd_ld_b, idx_b, out_b, rel_out_b,
                                        sums_b, c, nb);
    if (ra != PN2_OK || rb != PN2_OK || na == 0 || nb == 0) {
        if ((ra != PN2_OK && ra != 1) || (rb != PN2_OK && rb != 1)) return ra != PN2_OK && ra != 1 ? ra : rb;
        const int rc = pn2x_sa_layer1_stats(b, n, s, k_a, c1_a, a1f_a, a1f_ld_a, xyz, cxyz, wx_a, wx_ld_a, cadd_a, cadd_ld_a, idx_a, out_a,
                                            rel_out_a, sums_a, stream);
        if (rc != PN2_OK) return rc;
        return pn2x_sa_layer1_stats(b, n, s, k_b, c1_b, a1f_b, a1f_ld_b, xyz, cxyz, wx_b, wx_ld_b, cadd_b, cadd_ld_b, idx_b, out_b, rel_out_b,
                                    sums_b, stream);
    }
    hipLaunchKernelGGL(sa_layer1_stats_pair_kernel, dim3(na + nb, b), dim3(kTT), 0, (hipStream_t)stream, a, c, na);
    return check_launch();
}

extern "C" int pn2x_bn_sums_doubles(int c) { return 2 * c * pn2::kRep; }

namespace pn2 {
int inverse_index_launch(int b, int n_dst, int l, const int *idx, int *offsets, int *order, hipStream_t st) {
    const size_t lds = ((size_t)n_dst + 1 + kTT) * sizeof(int);
    if (lds > 64 * 1024) return PN2_ERANGE;
    hipLaunchKernelGGL(inverse_index_kernel, dim3(b), dim3(kTT), lds, st, n_dst, l, idx, offsets, order);
    return check_launch();
}

int inverse_index_chunked_launch(int b, int n_dst, int l, int mt, int nchunks, const int *idx, unsigned short *offsets, int off_stride,
                                 unsigned short *order, hipStream_t st) {
    const size_t lds = ((size_t)n_dst + 1 + 16) * sizeof(int);
    if (lds > 64 * 1024 || mt < 1 || mt > 4 * kInvT || (long)nchunks * mt < l || off_stride < n_dst + 1) return PN2_ERANGE;
    hipLaunchKernelGGL(inverse_index_chunked_kernel, dim3((unsigned)(b * nchunks)), dim3(kInvT), lds, st, n_dst, l, mt, nchunks, off_stride, idx,
                       offsets, order);
    return check_launch();
}
}  // namespace pn2

extern "C" int pn2x_inverse_index(int b, int n_dst, int l, const int *idx, int *offsets, int *order, void *stream) {
    using namespace pn2;
    if (b < 0 || n_dst < 1 || l < 0) return PN2_EINVAL;
    if (b == 0) return PN2_OK;
    if (!idx || !offsets || !order) return PN2_ENULL;
    return inverse_index_launch(b, n_dst, l, idx, offsets, order, (hipStream_t)stream);
}

extern "C" int pn2x_rows_segment_sum(int b, int n_dst, int m_src, int t, int c, const float *dout, int ldo, const int *offsets,
                                     const int *order, const float *weight, float *din, int ldi, int accumulate, void *stream) {
    using namespace pn2;
    if (b < 0 || n_dst < 1 || m_src < 0 || c < 0 || c % 4 || ldo < c || ldo % 4 || ldi < c || ldi % 4 || (t != 1 && t != 3)) return PN2_EINVAL;
    if (b == 0 || c == 0) return PN2_OK;
    if (!dout || !offsets || !order || !din || (t == 3 && !weight)) return PN2_ENULL;
    if (((uintptr_t)dout | (uintptr_t)din) % 16) return PN2_EINVAL;
    const int Q = c / 4;
    // long segments on average (>= 4 entries per destination) and a quad count that divides 256: split every segment over lanes
    int el = 1;
    if ((long)m_src * t >= 4L * n_dst && Q <= 64 && kTT % Q == 0) {
        el = kTT / Q;
        if (el > 16) el = 16;
    }
    if (el > 1) {
        const int dpb = kTT / (Q * el);
        const dim3 sgrid((unsigned)((n_dst + dpb - 1) / dpb), b);
        if (t == 1)
            hipLaunchKernelGGL(rows_segment_sum_split_kernel<1>, sgrid, dim3(kTT), 0, (hipStream_t)stream, n_dst, m_src, Q, el, dout, ldo, offsets, order, weight, din, ldi, accumulate);
        else
            hipLaunchKernelGGL(rows_segment_sum_split_kernel<3>, sgrid, dim3(kTT), 0, (hipStream_t)stream, n_dst, m_src, Q, el, dout, ldo, offsets, order, weight, din, ldi, accumulate);
        return check_launch();
    }
    const long total = (long)n_dst * Q;
    const dim3 grid((unsigned)((total + kTT - 1) / kTT), b);
    const SegSumArgs a{n_dst, m_src, Q, dout, ldo, offsets, order, weight, din, ldi, accumulate};
    if (t == 1)
        hipLaunchKernelGGL(rows_segment_sum_kernel<1>, grid, dim3(kTT), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(rows_segment_sum_kernel<3>, grid, dim3(kTT), 0, (hipStream_t)stream, a);
    return check_launch();
}

// pn2x_rows_segment_sum (t = 1) for two index lists over the same destination rows in ONE launch where both take the
// one-thread-per-(destination, quad) kernel (short segments); two calls otherwise.  Same results.
extern "C" int pn2x_rows_segment_sum_pair(int b, int n_dst, int m_a, int c_a, const float *dout_a, int ldo_a, const int *offsets_a,
                                          const int *order_a, float *din_a, int ldi_a, int m_b, int c_b, const float *dout_b, int ldo_b,
                                          const int *offsets_b, const int *order_b, float *din_b, int ldi_b, int accumulate, void *stream) {
    using namespace pn2;
    auto split = [&](int m, int c) { const int Q = c / 4; return c >= 4 && (long)m >= 4L * n_dst && Q <= 64 && kTT % Q == 0; };
    const bool ok = b > 0 && n_dst >= 1 && c_a >= 4 && c_b >= 4 && c_a % 4 == 0 && c_b % 4 == 0 && !split(m_a, c_a) && !split(m_b, c_b) &&
                    dout_a && dout_b && offsets_a && offsets_b && order_a && order_b && din_a && din_b && ldo_a >= c_a && ldo_b >= c_b &&
                    ldo_a % 4 == 0 && ldo_b % 4 == 0 && ldi_a >= c_a && ldi_b >= c_b && ldi_a % 4 == 0 && ldi_b % 4 == 0 &&
                    (((uintptr_t)dout_a | (uintptr_t)dout_b | (uintptr_t)din_a | (uintptr_t)din_b) % 16) == 0 && m_a >= 0 && m_b >= 0;
    if (!ok) {
        const int rc = pn2x_rows_segment_sum(b, n_dst, m_a, 1, c_a, dout_a, ldo_a, offsets_a, order_a, nullptr, din_a, ldi_a, accumulate, stream);
        if (rc != PN2_OK) return rc;
        return pn2x_rows_segment_sum(b, n_dst, m_b, 1, c_b, dout_b, ldo_b, offsets_b, order_b, nullptr, din_b, ldi_b, accumulate, stream);
    }
    const SegSumArgs a{n_dst, m_a, c_a / 4, dout_a, ldo_a, offsets_a, order_a, nullptr, din_a, ldi_a, accumulate};
    const SegSumArgs c{n_dst, m_b, c_b / 4, dout_b, ldo_b, offsets_b, order_b, nullptr, din_b, ldi_b, accumulate};
    const unsigned na = (unsigned)(((long)n_dst * a.Q + kTT - 1) / kTT), nb = (unsigned)(((long)n_dst * c.Q + kTT - 1) / kTT);
    hipLaunchKernelGGL(rows_segment_sum_pair_kernel, dim3(na + nb, b), dim3(kTT), 0, (hipStream_t)stream, a, c, na);
    return check_launch();
}
