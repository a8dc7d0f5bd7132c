// train_gemm.hip -- train-mode grouped-MLP layers on the fp32 matrix cores with BatchNorm folded INTO the GEMMs (gfx950).
//
// The reference trains every [Conv 1x1 + BatchNorm + ReLU] stack (network/models/pointnet_utils.py:399-403, :460-462,
// :504-506, :577-581) as separate convolution / batch-norm / ReLU kernels.  Round 2 of this repo ran the convolution as a
// library GEMM over point-major rows and everything between two GEMMs as streaming kernels (train_ops.hip): each (R x C)
// activation was then written once and read three times forward, and read / written seven times backward.  Here the
// BatchNorm lives in the GEMMs themselves, for every layer but the first of a stack (whose pre-activations come from a
// gather / a concatenated input) -- v_mfma_f32_32x32x2_f32 tile loops whose operands are transformed on the way into LDS:
//
//   forward   tg_fwd     Y_i = relu(BN_{i-1}(Y_{i-1})) W_i^T        A prologue: normalise + ReLU per input channel (constants
//                                                                    from the fp64 sums of the producer); epilogue: per-channel
//                                                                    sum / sum of squares of Y_i (the next BatchNorm's statistics)
//   backward  tg_dgrad   G_{i-1} = (dY_i W_i) . [H_{i-1} > 0]        A prologue: dY_i = gamma invstd (g - mean(g) - xhat mean(g xhat))
//                                                                    from (G_i, Y_i); epilogue: ReLU mask of layer i-1 from Y_{i-1},
//                                                                    sum(g), sum(g xhat) of layer i-1 (its BatchNorm backward sums)
//             tg_wgrad   dW_i = dY_i^T H_{i-1}                       both operands transformed on load; split over the rows, partial
//                                                                    tiles reduced by tg_reduce (which also emits dgamma / dbeta)
//
// so a hidden activation is written once and read once forward (as pre-activation), and neither the normalised activations
// H nor the pre-activation gradients dY ever exist in HBM.  The top of a stack (max over the K neighbours, or the
// materialised output) and its first layer keep the streaming kernels of train_ops.hip.
//
// Tiling: 256 threads = 4 waves; fwd / dgrad: 128 rows x NT columns per workgroup (a wave owns 32 rows x NT: NT/32
// accumulator blocks of 16 VGPRs), reduction staged through LDS in chunks of 32 as k-minor tiles with an odd row stride (the
// 32 rows a ds_read_b32 lane group touches fall on 32 different banks); operands are fetched into registers one chunk ahead
// of the MFMAs that consume the previous one.  The fp32 MFMA issues at the fp32 vector rate (64 cycles per 32x32x2), so two
// ds_read_b32 per instruction keep the LDS pipe under a tenth of its rate and the kernels are bound by the matrix pipe for
// wide layers (128+ channels) and by HBM for the narrow ones (32 / 64 channels x 262144 rows).
#include <cstdlib>
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
namespace tg {

constexpr int kT = 256;
constexpr int BM = 128;      // rows per tile (fwd / dgrad)
constexpr int KC = 32;       // reduction chunk
constexpr int LDK = KC + 1;  // k-minor LDS tiles: odd stride
constexpr int RC = 32;       // rows per chunk (wgrad: the reduction runs over rows)
constexpr int kMaxC = 512;   // channels of a fused layer (LDS constant tables)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu_nan(float h) { return !(h <= 0.f) ? h : 0.f; }  // propagates NaN like torch

// mean / invstd of one channel from the fp64 sums (kBnRep interleaved copies of [sum | sum of squares]); same arithmetic as
// train_ops.hip's bn_consts, so the fused and the streaming kernels normalise with identical constants
__device__ __forceinline__ void bn_channel(const double *__restrict__ sums, int C, int c, long rows, float eps, float &mean,
                                           float &invstd, double &var) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < kBnRep; ++r) {
        s1 += sums[(size_t)r * 2 * C + c];
        s2 += sums[(size_t)r * 2 * C + C + c];
    }
    const double m = s1 / (double)rows;
    double v = s2 / (double)rows - m * m;
    v = v > 0.0 ? v : 0.0;
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(v + (double)eps));
    var = v;
}

// per-lane partial column sums (lane l and l ^ 32 hold the same column of a 32x32 block) -> fp64 accumulators
template <int NT>
__device__ __forceinline__ void flush_column_sums(const float *cs, const float *cq, float *red, int C, int n0, double *__restrict__ sums) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int nb = 0; nb < NT / 32; ++nb) {
        const float s = cs[nb] + __shfl_xor(cs[nb], 32), q = cq[nb] + __shfl_xor(cq[nb], 32);
        if (lane < 32) {
            red[wave * NT + nb * 32 + lane] = s;
            red[(4 + wave) * NT + nb * 32 + lane] = q;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < NT) {
        const int c = threadIdx.x;
        const double s = (double)red[c] + (double)red[NT + c] + (double)red[2 * NT + c] + (double)red[3 * NT + c];
        const double q = (double)red[4 * NT + c] + (double)red[5 * NT + c] + (double)red[6 * NT + c] + (double)red[7 * NT + c];
        double *dst = sums + (size_t)((blockIdx.x + blockIdx.y) % kBnRep) * 2 * C;
        unsafeAtomicAdd(dst + n0 + c, s);
        unsafeAtomicAdd(dst + C + n0 + c, q);
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------------
struct FwdArgs {
    long R;
    int K, N;
    const float *X; int ldx;   // pre-activations of the previous layer (R x K)
    const float *W; int ldw;   // (N x K)
    float *Y; int ldy;         // (R x N)
    const double *sums_in;     // previous layer's forward sums
    const float *gamma, *beta, *conv_bias;
    float eps, momentum;
    float *running_mean, *running_var;
    long long *nbt;
    float *save_mean, *save_invstd;  // written by workgroup (0, 0): the consumer finalises its producer's statistics
    double *sums_out;                // this layer's forward sums (may be null)
};

template <int NT>
__global__ void __launch_bounds__(kT)
tg_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds;               // [K][4]: mean, invstd, gamma, beta
    float *As = cst + 4 * a.K;      // [BM][LDK]
    float *Bs = As + BM * LDK;      // [NT][LDK]
    float *red = Bs + NT * LDK;     // [8][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * NT;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0;
    for (int k = tid; k < a.K; k += kT) {
        float mean, invstd;
        double var;
        bn_channel(a.sums_in, a.K, k, a.R, a.eps, mean, invstd, var);
        cst[4 * k + 0] = mean; cst[4 * k + 1] = invstd; cst[4 * k + 2] = a.gamma[k]; cst[4 * k + 3] = a.beta[k];
        if (first) {
            a.save_mean[k] = mean;
            a.save_invstd[k] = invstd;
            if (a.running_mean) {  // torch: running = (1 - m) running + m batch, the variance unbiased
                const float bm = mean + (a.conv_bias ? a.conv_bias[k] : 0.f);  // the GEMM output excludes the conv bias (cancels in BN)
                const float bv = (float)(a.R > 1 ? var * ((double)a.R / (double)(a.R - 1)) : var);
                a.running_mean[k] = (1.f - a.momentum) * a.running_mean[k] + a.momentum * bm;
                a.running_var[k] = (1.f - a.momentum) * a.running_var[k] + a.momentum * bv;
            }
        }
    }
    if (first && tid == 0 && a.nbt) *a.nbt += 1;
    __syncthreads();

    const long T = (a.R + BM - 1) / BM;
    const int nchunks = a.K / KC;  // K % KC == 0 (launcher)
    const int kq = tid & 7, rb = tid >> 3;
    float4 pa[4], pb[NT / 32];
    // loads are unconditional (rows clamped to the last valid one): no exec-mask branches between them, all in flight together
    auto prefetch = [&](long tile, int chunk, bool with_b) {
        const int k = chunk * KC + 4 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long row = tile * BM + rb + 32 * i;
            row = row < a.R ? row : a.R - 1;
            pa[i] = *reinterpret_cast<const float4 *>(a.X + row * a.ldx + k);
        }
        if (with_b) {
#pragma unroll
            for (int i = 0; i < NT / 32; ++i) pb[i] = *reinterpret_cast<const float4 *>(a.W + (size_t)(n0 + rb + 32 * i) * a.ldw + k);
        }
    };
    auto commit = [&](long tile, int chunk, bool with_b) {
        const int k = chunk * KC + 4 * kq;
        float4 c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = *reinterpret_cast<const float4 *>(cst + 4 * (k + e));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool v = tile * BM + rb + 32 * i < a.R;  // rows beyond the problem contribute exact zeros
            const float x[4] = {pa[i].x, pa[i].y, pa[i].z, pa[i].w};
            float *dst = As + (rb + 32 * i) * LDK + 4 * kq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = relu_nan(((x[e] - c[e].x) * c[e].y) * c[e].z + c[e].w);
                dst[e] = v ? h : 0.f;
            }
        }
        if (with_b) {
#pragma unroll
            for (int i = 0; i < NT / 32; ++i) {
                float *dst = Bs + (rb + 32 * i) * LDK + 4 * kq;
                dst[0] = pb[i].x; dst[1] = pb[i].y; dst[2] = pb[i].z; dst[3] = pb[i].w;
            }
        }
    };

    f32x16 acc[NT / 32];
    float cs[NT / 32], cq[NT / 32];
#pragma unroll
    for (int nb = 0; nb < NT / 32; ++nb) cs[nb] = cq[nb] = 0.f;
    const int m = wave * 32 + (lane & 31), kh = lane >> 5;
    bool b_resident = false;  // a single-chunk reduction keeps the weight tile in LDS for every row tile
    long tile = blockIdx.x;
    int chunk = 0;
    if (tile < T) prefetch(tile, 0, true);
    while (tile < T) {
        const bool wb = !b_resident;
        commit(tile, chunk, wb);
        if (nchunks == 1) b_resident = true;
        __syncthreads();
        long ntile = tile;
        int nchunk = chunk + 1;
        if (nchunk == nchunks) { nchunk = 0; ntile += gridDim.x; }
        if (ntile < T) prefetch(ntile, nchunk, !b_resident);
        if (chunk == 0) {
#pragma unroll
            for (int nb = 0; nb < NT / 32; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        }
#pragma unroll 4
        for (int s = 0; s < KC / 2; ++s) {
            const float av = As[m * LDK + 2 * s + kh];
#pragma unroll
            for (int nb = 0; nb < NT / 32; ++nb) {
                const float bv = Bs[(nb * 32 + (lane & 31)) * LDK + 2 * s + kh];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
            }
        }
        if (chunk == nchunks - 1) {
            const long row0 = tile * BM + wave * 32 + 4 * kh;
            const bool full = tile * BM + BM <= a.R;  // workgroup-uniform: only the last tile stores under a row guard
#pragma unroll
            for (int nb = 0; nb < NT / 32; ++nb) {
                float s = 0.f, q = 0.f;
                float *dst = a.Y + n0 + nb * 32 + (lane & 31);
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(row0 + (r & 3) + 8 * (r >> 2)) * a.ldy] = acc[nb][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long row = row0 + (r & 3) + 8 * (r >> 2);
                        if (row < a.R) dst[row * a.ldy] = acc[nb][r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[nb][r];
                    s += v;
                    q += v * v;
                }
                cs[nb] += s;
                cq[nb] += q;
            }
        }
        __syncthreads();
        tile = ntile;
        chunk = nchunk;
    }
    if (a.sums_out) flush_column_sums<NT>(cs, cq, red, a.N, n0, a.sums_out);
}

// ---- the pre-activation gradient dY_i, computed on load -----------------------------------------------------------------------
// GMODE 0: G holds g_i = dH_i . [H_i > 0] (written by the dgrad epilogue of layer i+1)
// GMODE 1: G holds dH_i of a materialised top layer: the ReLU mask is recomputed from Y_i
// GMODE 2: G holds d(max over Kmax consecutive rows) (R / Kmax rows): routed to the recorded arg-max row, ReLU-masked
struct DySrc {
    const float *G; int ldg;
    const int *arg; int Kmax;  // GMODE 2; Kmax a power of two is divided by shift (kshift >= 0)
    int kshift;
    const float *Y; int ldy;
    const float *mean, *invstd, *gamma, *beta;
    const double *sums_bwd;  // layer i: sum(g), sum(g xhat)
};
// constants per channel n in LDS: [n][8] = mean, invstd, scale = gamma invstd, m1 = sum(g) / R, m2 = sum(g xhat) / R, gamma, beta, -
__device__ __forceinline__ void dy_constants(const DySrc &d, int C, int c0, int count, long R, float *cst) {
    const float inv_r = (float)(1.0 / (double)R);
    for (int i = threadIdx.x; i < count; i += kT) {
        const int c = c0 + i;
        double sa = 0.0, sb = 0.0;
        for (int r = 0; r < kBnRep; ++r) {
            sa += d.sums_bwd[(size_t)r * 2 * C + c];
            sb += d.sums_bwd[(size_t)r * 2 * C + C + c];
        }
        const float g = d.gamma[c], is = d.invstd[c];
        float *o = cst + 8 * i;
        o[0] = d.mean[c]; o[1] = is; o[2] = g * is; o[3] = (float)sa * inv_r; o[4] = (float)sb * inv_r; o[5] = g; o[6] = d.beta[c]; o[7] = 0.f;
    }
}

// the constants of four consecutive channels, fetched from LDS ONCE per chunk and thread (a thread's channels are the same
// for all its rows of a chunk; re-reading them per row made the commit phase LDS-bound: 32 ds_read_b128 per thread and chunk)
struct DyConst {
    float4 c0[4], c1[4];  // c0 = mean, invstd, scale, m1; c1 = m2, gamma, beta, -
};
__device__ __forceinline__ void dy_load_constants(DyConst &c, const float *cst) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c.c0[e] = *reinterpret_cast<const float4 *>(cst + 8 * e);
        c.c1[e] = *reinterpret_cast<const float4 *>(cst + 8 * e + 4);
    }
}
// one float4 of dY from the fetched registers (g: gradient source, y: pre-activation, ar: arg-max rows (GMODE 2)); kk = row
// within its max group (GMODE 2)
template <int GMODE>
__device__ __forceinline__ float4 dy_value(float4 gv, float4 yv, int4 av, int kk, bool valid, const DyConst &c) {
    const float g[4] = {gv.x, gv.y, gv.z, gv.w}, y[4] = {yv.x, yv.y, yv.z, yv.w};
    const int ar[4] = {av.x, av.y, av.z, av.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float4 c0 = c.c0[e], c1 = c.c1[e];
        const float xhat = (y[e] - c0.x) * c0.y;
        float gg = g[e];
        if constexpr (GMODE == 2) gg = ar[e] == kk ? gg : 0.f;
        if constexpr (GMODE != 0) gg = (xhat * c1.y + c1.z > 0.f) ? gg : 0.f;  // [relu(BN(y)) > 0], torch's evaluation order
        const float d = c0.z * (gg - c0.w - xhat * c1.x);
        o[e] = valid ? d : 0.f;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

// ---- backward, data: G_{i-1} = (dY_i W_i) . mask_{i-1}, plus the BatchNorm-backward sums of layer i-1 -------------------------
struct DgradArgs {
    long R;
    int Kd, N;  // Kd = channels of layer i (the reduction), N = channels of layer i-1
    DySrc dy;
    const float *W; int ldw;   // (Kd x N)
    const float *Yp; int ldyp; // pre-activations of layer i-1
    const float *mean_p, *invstd_p, *gamma_p, *beta_p;
    float *Gp; int ldgp;       // out: g_{i-1}
    double *sums_bwd_p;        // out: sum(g_{i-1}), sum(g_{i-1} xhat_{i-1})
};

template <int NT, int GMODE>
__global__ void __launch_bounds__(kT)
tg_dgrad_kernel(DgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cstA = lds;                 // [Kd][8]
    float *cstE = cstA + 8 * a.Kd;     // [NT][4]: mean, invstd, gamma, beta of layer i-1 (this workgroup's columns)
    float *As = cstE + 4 * NT;         // [BM][LDK]
    float *Bs = As + BM * LDK;         // [KC][NT]
    float *red = Bs + KC * NT;         // [8][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * NT;
    dy_constants(a.dy, a.Kd, 0, a.Kd, a.R, cstA);
    for (int j = tid; j < NT; j += kT) {
        cstE[4 * j + 0] = a.mean_p[n0 + j]; cstE[4 * j + 1] = a.invstd_p[n0 + j];
        cstE[4 * j + 2] = a.gamma_p[n0 + j]; cstE[4 * j + 3] = a.beta_p[n0 + j];
    }
    __syncthreads();

    const long T = (a.R + BM - 1) / BM;
    const int nchunks = a.Kd / KC;  // Kd % KC == 0 (launcher)
    const int kq = tid & 7, rb = tid >> 3;
    constexpr int BQ = NT / 4;        // float4 per weight row
    constexpr int BI = NT / 32;       // weight float4 per thread and chunk
    const int jq = tid % BQ, kr0 = tid / BQ;
    float4 pg[4], py[4];
    f32x4 pb[BI];  // native vector type: a conditionally written array of float4 structs stays in scratch memory (memcpy copies)
    int4 par[4];
    const DySrc &d = a.dy;
    auto prefetch = [&](long tile, int chunk, bool with_b) {
        const int k = chunk * KC + 4 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long row = tile * BM + rb + 32 * i;
            row = row < a.R ? row : a.R - 1;  // unconditional loads: the commit zeroes what lies beyond the problem
            py[i] = *reinterpret_cast<const float4 *>(d.Y + row * d.ldy + k);
            if constexpr (GMODE == 2) {
                const long grp = d.kshift >= 0 ? ((int)row >> d.kshift) : ((int)row / d.Kmax);
                pg[i] = *reinterpret_cast<const float4 *>(d.G + grp * d.ldg + k);
                par[i] = *reinterpret_cast<const int4 *>(d.arg + grp * d.ldg + k);
            } else {
                pg[i] = *reinterpret_cast<const float4 *>(d.G + row * d.ldg + k);
            }
        }
        if (with_b) {
#pragma unroll
            for (int i = 0; i < BI; ++i)
                pb[i] = *reinterpret_cast<const f32x4 *>(a.W + (size_t)(chunk * KC + kr0 + (kT / BQ) * i) * a.ldw + n0 + 4 * jq);
        }
    };
    auto commit = [&](long tile, int chunk, bool with_b) {
        const int k = chunk * KC + 4 * kq;
        DyConst dc;
        dy_load_constants(dc, cstA + 8 * k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = tile * BM + rb + 32 * i;
            int kk = 0;
            int4 av = make_int4(0, 0, 0, 0);
            if constexpr (GMODE == 2) {
                const int rr = (int)(row < a.R ? row : a.R - 1);
                kk = d.kshift >= 0 ? (rr & (d.Kmax - 1)) : (rr % d.Kmax);
                av = par[i];
            }
            const float4 dyv = dy_value<GMODE>(pg[i], py[i], av, kk, row < a.R, dc);
            float *dst = As + (rb + 32 * i) * LDK + 4 * kq;
            dst[0] = dyv.x; dst[1] = dyv.y; dst[2] = dyv.z; dst[3] = dyv.w;
        }
        if (with_b) {
#pragma unroll
            for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4 *>(Bs + (kr0 + (kT / BQ) * i) * NT + 4 * jq) = pb[i];
        }
    };

    f32x16 acc[NT / 32];
    float cs[NT / 32], cq[NT / 32];
#pragma unroll
    for (int nb = 0; nb < NT / 32; ++nb) cs[nb] = cq[nb] = 0.f;
    const int m = wave * 32 + (lane & 31), kh = lane >> 5;
    bool b_resident = false;
    long tile = blockIdx.x;
    int chunk = 0;
    if (tile < T) prefetch(tile, 0, true);
    while (tile < T) {
        commit(tile, chunk, !b_resident);
        if (nchunks == 1) b_resident = true;
        __syncthreads();
        long ntile = tile;
        int nchunk = chunk + 1;
        if (nchunk == nchunks) { nchunk = 0; ntile += gridDim.x; }
        const bool last = chunk == nchunks - 1;
        // the next chunk's operands are fetched behind this chunk's MFMAs; across a tile boundary only after the epilogue, whose
        // own loads and temporaries would otherwise have to live next to the prefetched registers
        if (!last && ntile < T) prefetch(ntile, nchunk, !b_resident);
        if (chunk == 0) {
#pragma unroll
            for (int nb = 0; nb < NT / 32; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        }
#pragma unroll 4
        for (int s = 0; s < KC / 2; ++s) {
            const float av = As[m * LDK + 2 * s + kh];
#pragma unroll
            for (int nb = 0; nb < NT / 32; ++nb) {
                const float bv = Bs[(2 * s + kh) * NT + nb * 32 + (lane & 31)];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
            }
        }
        if (last) {
            const long row0 = tile * BM + wave * 32 + 4 * kh;
            const bool full = tile * BM + BM <= a.R;
#pragma unroll
            for (int nb = 0; nb < NT / 32; ++nb) {
                const int j = nb * 32 + (lane & 31);
                const float4 ce = *reinterpret_cast<const float4 *>(cstE + 4 * j);
                float s = 0.f, q = 0.f;
                const float *yp = a.Yp + n0 + j;
                float *dst = a.Gp + n0 + j;
                float yv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // the sixteen pre-activation loads are independent and unconditional (row clamped)
                    long row = row0 + (r & 3) + 8 * (r >> 2);
                    row = row < a.R ? row : a.R - 1;
                    yv[r] = yp[row * a.ldyp];
                }
                float gv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float xhat = (yv[r] - ce.x) * ce.y;
                    gv[r] = (xhat * ce.z + ce.w > 0.f) ? acc[nb][r] : 0.f;  // rows beyond R have acc == 0
                    s += gv[r];
                    q += gv[r] * xhat;
                }
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(row0 + (r & 3) + 8 * (r >> 2)) * a.ldgp] = gv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long row = row0 + (r & 3) + 8 * (r >> 2);
                        if (row < a.R) dst[row * a.ldgp] = gv[r];
                    }
                }
                cs[nb] += s;
                cq[nb] += q;
                __builtin_amdgcn_sched_barrier(0);  // one column block at a time: 16 loads in flight, not 64 registers of them
            }
            if (ntile < T) prefetch(ntile, nchunk, !b_resident);
        }
        __syncthreads();
        tile = ntile;
        chunk = nchunk;
    }
    flush_column_sums<NT>(cs, cq, red, a.N, n0, a.sums_bwd_p);
}

// ---- backward, weights: dW_i (N x K) = dY_i^T H_{i-1}, reduction over the rows split across workgroups -----------------------
struct WgradArgs {
    long R;
    int N, K;  // N = channels of layer i (rows of dW), K = channels of layer i-1 (columns of dW)
    DySrc dy;
    const float *Yp; int ldyp;  // pre-activations of layer i-1 (H_{i-1} = relu(BN(Yp)) on load)
    const float *mean_p, *invstd_p, *gamma_p, *beta_p;
    float *partial;             // [gridDim.x * KG][N][K]
    long rows_per_split;        // multiple of RC
    float *dW;                  // zeroed here (row split 0) for tg_reduce's atomic accumulation
};

template <int MT, int NT, int GMODE>
__global__ void __launch_bounds__(kT)
tg_wgrad_kernel(WgradArgs a) {
    constexpr int STRIPS = MT / 32, KG = 4 / STRIPS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cstA = lds;               // [MT][8]
    float *cstH = cstA + 8 * MT;     // [NT][4]
    float *Ad = cstH + 4 * NT;       // [RC][MT]
    float *Hd = Ad + RC * MT;        // [RC][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * MT, j0 = blockIdx.z * NT;
    dy_constants(a.dy, a.N, m0, MT, a.R, cstA);
    for (int j = tid; j < NT; j += kT) {
        cstH[4 * j + 0] = a.mean_p[j0 + j]; cstH[4 * j + 1] = a.invstd_p[j0 + j];
        cstH[4 * j + 2] = a.gamma_p[j0 + j]; cstH[4 * j + 3] = a.beta_p[j0 + j];
    }
    __syncthreads();
    const long r_begin = (long)blockIdx.x * a.rows_per_split;
    const long r_end = (r_begin + a.rows_per_split) < a.R ? (r_begin + a.rows_per_split) : a.R;
    constexpr int AQ = MT / 4, AI = MT / 32, HQ = NT / 4, HI = NT / 32;
    const int anq = tid % AQ, ar0 = tid / AQ, hjq = tid % HQ, hr0 = tid / HQ;
    float4 pg[AI], py[AI], ph[HI];
    int4 par[AI];
    const DySrc &d = a.dy;
    const long r_last = r_end - 1;
    auto prefetch = [&](long rbase) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            long row = rbase + ar0 + (kT / AQ) * i;
            row = row < r_end ? row : r_last;  // unconditional loads; the commit zeroes rows beyond this split
            py[i] = *reinterpret_cast<const float4 *>(d.Y + row * d.ldy + m0 + 4 * anq);
            if constexpr (GMODE == 2) {
                const long grp = d.kshift >= 0 ? ((int)row >> d.kshift) : ((int)row / d.Kmax);
                pg[i] = *reinterpret_cast<const float4 *>(d.G + grp * d.ldg + m0 + 4 * anq);
                par[i] = *reinterpret_cast<const int4 *>(d.arg + grp * d.ldg + m0 + 4 * anq);
            } else {
                pg[i] = *reinterpret_cast<const float4 *>(d.G + row * d.ldg + m0 + 4 * anq);
            }
        }
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            long row = rbase + hr0 + (kT / HQ) * i;
            row = row < r_end ? row : r_last;
            ph[i] = *reinterpret_cast<const float4 *>(a.Yp + row * a.ldyp + j0 + 4 * hjq);
        }
    };
    DyConst dc;  // this thread's four dY channels and four H channels are the same for every chunk of the kernel
    dy_load_constants(dc, cstA + 8 * 4 * anq);
    float4 c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = *reinterpret_cast<const float4 *>(cstH + 4 * (4 * hjq + e));
    auto commit = [&](long rbase) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int rr = ar0 + (kT / AQ) * i;
            const long row = rbase + rr;
            int kk = 0;
            int4 av = make_int4(0, 0, 0, 0);
            if constexpr (GMODE == 2) {
                const int rr_ = (int)(row < r_end ? row : r_last);
                kk = d.kshift >= 0 ? (rr_ & (d.Kmax - 1)) : (rr_ % d.Kmax);
                av = par[i];
            }
            *reinterpret_cast<float4 *>(Ad + rr * MT + 4 * anq) = dy_value<GMODE>(pg[i], py[i], av, kk, row < r_end, dc);
        }
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            const int rr = hr0 + (kT / HQ) * i;
            const bool v = rbase + rr < r_end;
            const float x[4] = {ph[i].x, ph[i].y, ph[i].z, ph[i].w};
            float h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = relu_nan(((x[e] - c[e].x) * c[e].y) * c[e].z + c[e].w);
                h[e] = v ? t : 0.f;
            }
            *reinterpret_cast<float4 *>(Hd + rr * NT + 4 * hjq) = make_float4(h[0], h[1], h[2], h[3]);
        }
    };

    f32x16 acc[NT / 32];
#pragma unroll
    for (int nb = 0; nb < NT / 32; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int strip = wave % STRIPS, kg = wave / STRIPS, kh = lane >> 5;
    long rbase = r_begin;
    if (rbase < r_end) prefetch(rbase);
    while (rbase < r_end) {
        commit(rbase);
        __syncthreads();
        const long nbase = rbase + RC;
        if (nbase < r_end) prefetch(nbase);
        for (int s = kg; s < RC / 2; s += KG) {
            const float av = Ad[(2 * s + kh) * MT + strip * 32 + (lane & 31)];
#pragma unroll
            for (int nb = 0; nb < NT / 32; ++nb) {
                const float bv = Hd[(2 * s + kh) * NT + nb * 32 + (lane & 31)];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
            }
        }
        __syncthreads();
        rbase = nbase;
    }
    float *out = a.partial + ((size_t)blockIdx.x * KG + kg) * (size_t)a.N * a.K;
#pragma unroll
    for (int nb = 0; nb < NT / 32; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = m0 + strip * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            out[(size_t)n * a.K + j0 + nb * 32 + (lane & 31)] = acc[nb][r];
        }
    if (blockIdx.x == 0) {  // this (m, j) tile of dW starts from zero: tg_reduce accumulates its slices of the partials into it
        for (int e = tid; e < MT * NT; e += kT) a.dW[(size_t)(m0 + e / NT) * a.K + j0 + e % NT] = 0.f;
    }
}

// dW += sum of a slice of the partial tiles (grid.y slices, 64 elements x 4 partial lanes per workgroup: the partials of a
// narrow layer are few elements deep and thousands of tiles long, so the sum is spread over the partial axis as well);
// dgamma / dbeta of the layer from its backward sums; the conv bias in front of a BatchNorm has an identically zero gradient
__global__ void __launch_bounds__(kT)
tg_reduce_kernel(const float *__restrict__ partial, int P, int numel, float *__restrict__ dW, const double *__restrict__ sums_bwd, int N,
                 float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dbias) {
    __shared__ float red[4][64];
    const int el = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = (p0 + per) < P ? (p0 + per) : P;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < numel) {
        int p = p0 + pl;
        for (; p + 12 < p1; p += 16) {
            s0 += partial[(size_t)p * numel + e];
            s1 += partial[(size_t)(p + 4) * numel + e];
            s2 += partial[(size_t)(p + 8) * numel + e];
            s3 += partial[(size_t)(p + 12) * numel + e];
        }
        for (; p < p1; p += 4) s0 += partial[(size_t)p * numel + e];
    }
    red[pl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (pl == 0 && e < numel) atomicAdd(dW + e, (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]));
    const int c = blockIdx.x * kT + threadIdx.x;
    if (blockIdx.y == 0 && c < N && sums_bwd) {
        double sa = 0.0, sb = 0.0;
        for (int r = 0; r < kBnRep; ++r) {
            sa += sums_bwd[(size_t)r * 2 * N + c];
            sb += sums_bwd[(size_t)r * 2 * N + N + c];
        }
        dbeta[c] = (float)sa;
        dgamma[c] = (float)sb;
        if (dbias) dbias[c] = 0.f;
    }
}

// The reductions of SEVERAL layers in one launch: dW of a layer is not needed before the optimiser, so the backward only
// writes partial tiles and one launch at the end of the pass sums them all (18 launches of ~5 us per training step otherwise).
constexpr int kRmMax = 24;
struct ReduceMulti {
    const float *partial[kRmMax];
    float *dW[kRmMax];
    const double *sums[kRmMax];
    float *dgamma[kRmMax], *dbeta[kRmMax], *dbias[kRmMax];
    int P[kRmMax], numel[kRmMax], N[kRmMax], ps[kRmMax], sld[kRmMax];  // sld: channels of the whole layer (N: of this column slice)
    int block_start[kRmMax + 1];
    int n;
};

// 16-byte loads: a thread sums four consecutive elements of its slice of the partial tiles (numel is a multiple of 4: the tiles
// are C_i x C_{i-1} with both multiples of 32), four independent accumulator quads in flight.  With 4-byte loads the launch
// moved the ~200 MB of partial tiles of a training step at 2.8 TB/s.
__global__ void __launch_bounds__(kT)
tg_reduce_multi_kernel(ReduceMulti a) {
    __shared__ float4 red[4][64];
    int t = 0;
    while (t + 1 < a.n && a.block_start[t + 1] <= (int)blockIdx.x) ++t;
    const int local = blockIdx.x - a.block_start[t];
    const int numel = a.numel[t], P = a.P[t], N = a.N[t];
    const int nq = numel / 4, nbx = (nq + 63) / 64;
    const int bx = local % nbx, by = local / nbx;
    const float4 *__restrict__ partial = reinterpret_cast<const float4 *>(a.partial[t]);
    const int el = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int e = bx * 64 + el;
    const int per = (P + a.ps[t] - 1) / a.ps[t];
    const int p0 = by * per, p1 = (p0 + per) < P ? (p0 + per) : P;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto add = [](float4 &s, const float4 v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
    if (e < nq) {
        int p = p0 + pl;
        for (; p + 12 < p1; p += 16) {
            const float4 v0 = partial[(size_t)p * nq + e], v1 = partial[(size_t)(p + 4) * nq + e];
            const float4 v2 = partial[(size_t)(p + 8) * nq + e], v3 = partial[(size_t)(p + 12) * nq + e];
            add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3);
        }
        for (; p < p1; p += 4) add(s0, partial[(size_t)p * nq + e]);
    }
    add(s0, s1); add(s2, s3); add(s0, s2);
    red[pl][el] = s0;
    __syncthreads();
    if (pl == 0 && e < nq) {
        float4 r = red[0][el];
        add(r, red[1][el]);
        float4 q = red[2][el];
        add(q, red[3][el]);
        add(r, q);
        float *dst = a.dW[t] + 4 * (size_t)e;
        // hardware fp32 atomics (global_atomic_add_f32): the plain atomicAdd compiles to a compare-and-swap loop
        unsafeAtomicAdd(dst + 0, r.x); unsafeAtomicAdd(dst + 1, r.y); unsafeAtomicAdd(dst + 2, r.z); unsafeAtomicAdd(dst + 3, r.w);
    }
    const int c = bx * kT + threadIdx.x;
    if (by == 0 && c < N && a.sums[t]) {
        double sa = 0.0, sb = 0.0;
        const int sld = a.sld[t];
        for (int r = 0; r < kBnRep; ++r) {
            sa += a.sums[t][(size_t)r * 2 * sld + c];
            sb += a.sums[t][(size_t)r * 2 * sld + sld + c];
        }
        a.dbeta[t][c] = (float)sa;
        a.dgamma[t][c] = (float)sb;
        if (a.dbias[t]) a.dbias[t][c] = 0.f;
    }
}

static int kshift_of(int k) {
    if (k < 1 || (k & (k - 1))) return -1;
    int s = 0;
    while ((1 << s) < k) ++s;
    return s;
}
static int col_tile(int n) { return n % 128 == 0 ? 128 : (n % 64 == 0 ? 64 : (n % 32 == 0 ? 32 : 0)); }
static bool bad_ld(int ld, int c) { return ld < c || ld % 4; }

}  // namespace tg
}  // namespace pn2

using namespace pn2;
using namespace pn2::tg;

extern "C" int pn2x_tg_supported(int c_in, int c_out) {
    return (c_in >= KC && c_in % KC == 0 && c_in <= kMaxC && c_out <= 4096 && col_tile(c_out) != 0) ? 1 : 0;
}

extern "C" int pn2x_tg_fwd(long rows, int k, int n, const float *x, int ldx, const float *w, int ldw, float *y, int ldy,
                           const double *sums_in, const float *gamma, const float *beta, const float *conv_bias, float eps,
                           float momentum, float *running_mean, float *running_var, long long *num_batches_tracked,
                           float *save_mean, float *save_invstd, double *sums_out, void *stream) {
    if (rows < 1 || !pn2x_tg_supported(k, n) || bad_ld(ldx, k) || bad_ld(ldw, k) || bad_ld(ldy, n)) return PN2_EINVAL;
    if (!x || !w || !y || !sums_in || !gamma || !beta || !save_mean || !save_invstd) return PN2_ENULL;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) % 16) return PN2_EINVAL;
    FwdArgs a{rows, k, n, x, ldx, w, ldw, y, ldy, sums_in, gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
              num_batches_tracked, save_mean, save_invstd, sums_out};
    const int nt = col_tile(n);
    const long tiles = (rows + BM - 1) / BM;
    const int ny = n / nt;
    long gx = tiles;
    const long cap = (long)num_compute_units() * 4 / ny;  // a few workgroups per CU: statistics stay in registers across row tiles
    if (gx > cap) gx = cap < 1 ? 1 : cap;
    const dim3 grid((unsigned)gx, ny);
    const size_t lds = (size_t)(4 * k + BM * LDK + nt * LDK + 8 * nt) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (nt == 128) hipLaunchKernelGGL(tg_fwd_kernel<128>, grid, dim3(kT), lds, st, a);
    else if (nt == 64) hipLaunchKernelGGL(tg_fwd_kernel<64>, grid, dim3(kT), lds, st, a);
    else hipLaunchKernelGGL(tg_fwd_kernel<32>, grid, dim3(kT), lds, st, a);
    return check_launch();
}

static int check_dy(int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi, int ldyi, int c, const float *mean,
                    const float *invstd, const float *gamma, const float *beta, const double *sums, long rows) {
    if (gmode < 0 || gmode > 2 || bad_ld(ldg, c) || bad_ld(ldyi, c)) return PN2_EINVAL;
    if (rows > 0x7fffffffL || (gmode == 2 && (kmax < 1 || rows % kmax))) return PN2_EINVAL;
    if (!g || !yi || !mean || !invstd || !gamma || !beta || !sums || (gmode == 2 && !arg)) return PN2_ENULL;
    if (((uintptr_t)g | (uintptr_t)yi | (uintptr_t)arg) % 16) return PN2_EINVAL;
    return PN2_OK;
}

extern "C" int pn2x_tg_dgrad(long rows, int kd, int n, int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi,
                             int ldyi, const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i,
                             const double *sums_bwd_i, const float *w, int ldw, const float *yp, int ldyp, const float *mean_p,
                             const float *invstd_p, const float *gamma_p, const float *beta_p, float *gp, int ldgp,
                             double *sums_bwd_p, void *stream) {
    if (rows < 1 || !pn2x_tg_supported(kd, n) || bad_ld(ldw, n) || bad_ld(ldyp, n) || bad_ld(ldgp, n)) return PN2_EINVAL;
    if (int rc = check_dy(gmode, g, ldg, arg, kmax, yi, ldyi, kd, mean_i, invstd_i, gamma_i, beta_i, sums_bwd_i, rows)) return rc;
    if (!w || !yp || !mean_p || !invstd_p || !gamma_p || !beta_p || !gp || !sums_bwd_p) return PN2_ENULL;
    if (((uintptr_t)w | (uintptr_t)yp | (uintptr_t)gp) % 16) return PN2_EINVAL;
    DgradArgs a{rows, kd, n, DySrc{g, ldg, arg, kmax, kshift_of(kmax), yi, ldyi, mean_i, invstd_i, gamma_i, beta_i, sums_bwd_i}, w, ldw, yp, ldyp,
                mean_p, invstd_p, gamma_p, beta_p, gp, ldgp, sums_bwd_p};
    // column tiles of at most 64: with the 128-wide tile the kernel needs 256 registers (prefetched operands + accumulators + the
    // epilogue's loads) and ONE workgroup per CU then serialises its HBM round trips; two 64-wide tiles recompute dY twice but
    // keep two to three workgroups resident
    int nt = col_tile(n);
    if (nt > 64) nt = 64;
    const long tiles = (rows + BM - 1) / BM;
    const int ny = n / nt;
    long gx = tiles;
    const long cap = (long)num_compute_units() * 6 / ny;
    if (gx > cap) gx = cap < 1 ? 1 : cap;
    const dim3 grid((unsigned)gx, ny);
    const size_t lds = (size_t)(8 * kd + 4 * nt + BM * LDK + KC * nt + 8 * nt) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define PN2_TG_D(NT_, GM_) hipLaunchKernelGGL((tg_dgrad_kernel<NT_, GM_>), grid, dim3(kT), lds, st, a)
#define PN2_TG_DN(GM_) do { if (nt == 64) PN2_TG_D(64, GM_); else PN2_TG_D(32, GM_); } while (0)
    if (gmode == 0) PN2_TG_DN(0);
    else if (gmode == 1) PN2_TG_DN(1);
    else PN2_TG_DN(2);
#undef PN2_TG_DN
#undef PN2_TG_D
    return check_launch();
}

// partial-buffer planning: number of row splits and the floats the partial tiles need
static void wgrad_plan(long rows, int n, int k, int &mt, int &nt, long &rows_per_split, int &splits, int &kg) {
    mt = col_tile(n);
    nt = col_tile(k);
    kg = 4 / (mt / 32);
    const long budget = 4L << 20;  // floats of partial tiles
    long s = budget / ((long)n * k * kg);
    const long tiles = (long)(n / mt) * (k / nt);
    const long want = (4L * num_compute_units() + tiles - 1) / tiles;  // ~4 workgroups per CU
    if (s > want) s = want;
    if (s < 1) s = 1;
    long rps = (rows + s - 1) / s;
    rps = (rps + RC - 1) / RC * RC;
    rows_per_split = rps;
    splits = (int)((rows + rps - 1) / rps);
}

extern "C" long pn2x_tg_wgrad_partial_floats(long rows, int n, int k) {
    if (rows < 1 || !pn2x_tg_supported(k, n) || col_tile(k) == 0) return -1;
    int mt, nt, splits, kg;
    long rps;
    wgrad_plan(rows, n, k, mt, nt, rps, splits, kg);
    return (long)splits * kg * n * k;
}

extern "C" int pn2x_tg_wgrad(long rows, int n, int k, int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi,
                             int ldyi, const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i,
                             const double *sums_bwd_i, const float *yp, int ldyp, const float *mean_p, const float *invstd_p,
                             const float *gamma_p, const float *beta_p, float *partial, long partial_floats, float *dw,
                             float *dgamma, float *dbeta, float *dbias, void *stream) {
    return pn2x_tg_wgrad2(rows, n, k, gmode, g, ldg, arg, kmax, yi, ldyi, mean_i, invstd_i, gamma_i, beta_i, sums_bwd_i, yp, ldyp, mean_p,
                          invstd_p, gamma_p, beta_p, partial, partial_floats, dw, dgamma, dbeta, dbias, nullptr, stream);
}

// n_partials != NULL: only the partial tiles are written (and dw zeroed); *n_partials receives their count for a later
// pn2x_tg_reduce_multi over several layers.
extern "C" int pn2x_tg_wgrad2(long rows, int n, int k, int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi,
                              int ldyi, const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i,
                              const double *sums_bwd_i, const float *yp, int ldyp, const float *mean_p, const float *invstd_p,
                              const float *gamma_p, const float *beta_p, float *partial, long partial_floats, float *dw,
                              float *dgamma, float *dbeta, float *dbias, int *n_partials, void *stream) {
    if (rows < 1 || !pn2x_tg_supported(k, n) || col_tile(k) == 0 || bad_ld(ldyp, k)) return PN2_EINVAL;
    if (int rc = check_dy(gmode, g, ldg, arg, kmax, yi, ldyi, n, mean_i, invstd_i, gamma_i, beta_i, sums_bwd_i, rows)) return rc;
    if (!yp || !mean_p || !invstd_p || !gamma_p || !beta_p || !partial || !dw || !dgamma || !dbeta) return PN2_ENULL;
    if (((uintptr_t)yp | (uintptr_t)partial) % 16) return PN2_EINVAL;
    int mt, nt, splits, kg;
    long rps;
    wgrad_plan(rows, n, k, mt, nt, rps, splits, kg);
    if (partial_floats < (long)splits * kg * n * k) return PN2_ESCRATCH;
    WgradArgs a{rows, n, k, DySrc{g, ldg, arg, kmax, kshift_of(kmax), yi, ldyi, mean_i, invstd_i, gamma_i, beta_i, sums_bwd_i}, yp, ldyp,
                mean_p, invstd_p, gamma_p, beta_p, partial, rps, dw};
    const dim3 grid(splits, n / mt, k / nt);
    const size_t lds = (size_t)(8 * mt + 4 * nt + RC * mt + RC * nt) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define PN2_TG_W(MT_, NT_, GM_) hipLaunchKernelGGL((tg_wgrad_kernel<MT_, NT_, GM_>), grid, dim3(kT), lds, st, a)
#define PN2_TG_WN(MT_, GM_) do { if (nt == 128) PN2_TG_W(MT_, 128, GM_); else if (nt == 64) PN2_TG_W(MT_, 64, GM_); else PN2_TG_W(MT_, 32, GM_); } while (0)
#define PN2_TG_WM(GM_) do { if (mt == 128) PN2_TG_WN(128, GM_); else if (mt == 64) PN2_TG_WN(64, GM_); else PN2_TG_WN(32, GM_); } while (0)
    if (gmode == 0) PN2_TG_WM(0);
    else if (gmode == 1) PN2_TG_WM(1);
    else PN2_TG_WM(2);
#undef PN2_TG_WM
#undef PN2_TG_WN
#undef PN2_TG_W
    if (int rc = check_launch()) return rc;
    const int numel = n * k, P = splits * kg;
    if (n_partials) { *n_partials = P; return PN2_OK; }
    int ps = P / 32;  // >= 32 partial tiles per slice (8 per lane)
    if (ps < 1) ps = 1;
    if (ps > 64) ps = 64;
    // (numel + 63) / 64 >= (n + 255) / 256 workgroups along x: the dgamma / dbeta tail covers every channel
    hipLaunchKernelGGL(tg_reduce_kernel, dim3((numel + 63) / 64, ps), dim3(kT), 0, st, partial, P, numel, dw, sums_bwd_i, n,
                       dgamma, dbeta, dbias);
    return check_launch();
}

extern "C" int pn2x_tg_reduce_multi(int count, const float *const *partial, const int *n_partials, const int *numel, float *const *dw,
                                    const double *const *sums_bwd, const int *channels, float *const *dgamma, float *const *dbeta,
                                    float *const *dbias, void *stream) {
    return pn2x_tg_reduce_multi2(count, partial, n_partials, numel, dw, sums_bwd, channels, nullptr, dgamma, dbeta, dbias, stream);
}

// sums_ld (may be NULL: = channels): entry j is a column slice of a layer with sums_ld[j] channels (sums_bwd[j], dgamma[j], dbeta[j],
// dbias[j], dw[j] already offset to the slice)
extern "C" int pn2x_tg_reduce_multi2(int count, const float *const *partial, const int *n_partials, const int *numel, float *const *dw,
                                     const double *const *sums_bwd, const int *channels, const int *sums_ld, float *const *dgamma,
                                     float *const *dbeta, float *const *dbias, void *stream) {
    if (count < 0) return PN2_EINVAL;
    if (count == 0) return PN2_OK;
    if (!partial || !n_partials || !numel || !dw || !sums_bwd || !channels || !dgamma || !dbeta || !dbias) return PN2_ENULL;
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < count; t0 += kRmMax) {
        ReduceMulti a;
        a.n = (count - t0) < kRmMax ? (count - t0) : kRmMax;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const int j = t0 + i;
            if (!partial[j] || !dw[j] || !dgamma[j] || !dbeta[j] || n_partials[j] < 1 || numel[j] < 1 || channels[j] < 1) return PN2_EINVAL;
            a.partial[i] = partial[j]; a.dW[i] = dw[j]; a.sums[i] = sums_bwd[j];
            a.dgamma[i] = dgamma[j]; a.dbeta[i] = dbeta[j]; a.dbias[i] = dbias[j];
            a.P[i] = n_partials[j]; a.numel[i] = numel[j]; a.N[i] = channels[j];
            a.sld[i] = sums_ld ? sums_ld[j] : channels[j];
            if (a.sld[i] < channels[j]) return PN2_EINVAL;
            constexpr int per_slice = 128;  // (32 in round 4: 59.6 -> 50.0 us, a quarter of the atomics)
            // partial tiles per slice (PN2_TGR_SLICE: probes).  32 per slice = 8 slices x numel atomics per layer: 59.6 us for the step's
            // 200 MB of partial tiles; 128 per slice: 50.0 us (fewer atomics, still thousands of workgroups)
            int ps = a.P[i] / per_slice;
            if (ps < 1) ps = 1;
            if (ps > 64) ps = 64;
            a.ps[i] = ps;
            a.block_start[i] = blocks;
            if (numel[j] % 4 || (uintptr_t)partial[j] % 16) return PN2_EINVAL;
            blocks += ((numel[j] / 4 + 63) / 64) * ps;
        }
        a.block_start[a.n] = blocks;
        hipLaunchKernelGGL(tg_reduce_multi_kernel, dim3(blocks), dim3(kT), 0, st, a);
    }
    return check_launch();
}
