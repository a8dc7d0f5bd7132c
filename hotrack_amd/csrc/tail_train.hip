// tail_train.hip -- the 21-token tail of HandTrackNet in TRAINING mode (hand_network.py:139-147 with attn=False,
// transformer.py:65-67): LayerNorms, two FFNs with dropout and residuals.  In a captured training step the torch composition
// is ~70 element-wise / reduce / fill launches of 4-5 us (forward + backward) around seven small GEMMs; here every run between
// two GEMMs is one launch per direction:
//
//   tail_ln_fwd / _bwd          u = x + dropout(y + bias)   [y, bias optional]   out = LN_b(LN_a(u))   [LN_b optional]
//                               backward: dx, dy, dbias, dgamma / dbeta of both LayerNorms (column sums by atomics onto a zeroed buffer)
//   tail_relu_drop_fwd / _bwd   h = dropout(relu(z + bias));  backward: dz, dbias
//
// Dropout masks are a hash of (seed, site, element index) -- regenerated in the backward, never stored; the seed is a device
// counter advanced by the first kernel of a forward (graph-capture safe) and handed to the later kernels through a per-forward
// tensor.  torch semantics: kept elements scaled by 1 / (1 - p); LayerNorm with biased variance and eps inside the square root.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
namespace tt {

__device__ __forceinline__ float wsum(float v) { return wave_sum_f32(v); }

// keep-mask of element i at dropout site `site`: a 32-bit mix of (seed, site, i) compared with p * 2^32
__device__ __forceinline__ bool keep(unsigned long long seed, unsigned site, unsigned long long i, unsigned thresh) {
    unsigned long long z = seed * 0x9E3779B97F4A7C15ull + ((unsigned long long)site << 40) + i;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32) >= thresh;
}
__host__ __device__ inline unsigned drop_threshold(float p) { return p <= 0.f ? 0u : (p >= 1.f ? 0xFFFFFFFFu : (unsigned)((double)p * 4294967296.0)); }

struct LnArgs {
    long rows; int c;
    const float *x, *y, *bias;                 // u = x + drop(y + bias)
    float p; unsigned site;
    const long long *seed_in; long long *seed_dev, *seed_out;  // seed_dev != null: this launch advances the counter (no dropout in it)
    const float *ga, *ba; float eps_a;
    const float *gb, *bb; float eps_b;         // gb == null: one LayerNorm
    float *out;
    float *stats;                              // (rows, 4): mean_a, rstd_a, mean_b, rstd_b
};

template <int EPL>
__global__ void __launch_bounds__(256)
tail_ln_fwd_kernel(LnArgs a) {
    if (a.seed_dev && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long s = a.seed_dev[0] + 1;
        a.seed_dev[0] = s;
        a.seed_out[0] = s;
    }
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int lane = threadIdx.x & 63, c = a.c;
    const unsigned thresh = drop_threshold(a.p);
    const unsigned long long seed = (a.y && thresh) ? (unsigned long long)a.seed_in[0] : 0ull;
    const float scale = a.p < 1.f ? 1.f / (1.f - a.p) : 0.f;
    float v[EPL], s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int ch = lane + 64 * e;
        float t = 0.f;
        if (ch < c) {
            t = a.x[row * c + ch];
            if (a.y) {
                float d = a.y[row * c + ch] + (a.bias ? a.bias[ch] : 0.f);
                if (thresh) d = keep(seed, a.site, (unsigned long long)row * c + ch, thresh) ? d * scale : 0.f;
                t += d;
            }
        }
        v[e] = t;
        s += t;
    }
    const float inv_c = 1.f / (float)c;
    float mean = wsum(s) * inv_c, q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { const float d = (lane + 64 * e < c) ? v[e] - mean : 0.f; q += d * d; }
    float rstd = rsqrtf(wsum(q) * inv_c + a.eps_a);
    if (lane == 0) { a.stats[row * 4 + 0] = mean; a.stats[row * 4 + 1] = rstd; }
    s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int ch = lane + 64 * e;
        v[e] = ch < c ? (v[e] - mean) * rstd * a.ga[ch] + a.ba[ch] : 0.f;
        s += v[e];
    }
    if (a.gb) {
        mean = wsum(s) * inv_c;
        q = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { const float d = (lane + 64 * e < c) ? v[e] - mean : 0.f; q += d * d; }
        rstd = rsqrtf(wsum(q) * inv_c + a.eps_b);
        if (lane == 0) { a.stats[row * 4 + 2] = mean; a.stats[row * 4 + 3] = rstd; }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int ch = lane + 64 * e;
            if (ch < c) v[e] = (v[e] - mean) * rstd * a.gb[ch] + a.bb[ch];
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int ch = lane + 64 * e;
        if (ch < c) a.out[row * c + ch] = v[e];
    }
}

struct LnBwdArgs {
    LnArgs f;              // the forward's arguments (out unused)
    const float *dout;
    float *dx, *dy;        // dy: null when the forward had no y
    float *dga, *dba, *dgb, *dbb, *dbias;  // zero-initialised accumulators (c floats each; dgb / dbb / dbias may be null)
    int rows_per_wave;
};

template <int EPL>
__global__ void __launch_bounds__(256)
tail_ln_bwd_kernel(LnBwdArgs b) {
    const LnArgs &a = b.f;
    const int lane = threadIdx.x & 63, c = a.c;
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long r0 = wave_id * b.rows_per_wave;
    const long r1 = r0 + b.rows_per_wave < a.rows ? r0 + b.rows_per_wave : a.rows;
    const unsigned thresh = drop_threshold(a.p);
    const unsigned long long seed = (a.y && thresh) ? (unsigned long long)a.seed_in[0] : 0ull;
    const float scale = a.p < 1.f ? 1.f / (1.f - a.p) : 0.f;
    const float inv_c = 1.f / (float)c;
    float ga[EPL], gb[EPL], ba[EPL];
    float sga[EPL], sba[EPL], sgb[EPL], sbb[EPL], sbias[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int ch = lane + 64 * e;
        ga[e] = ch < c ? a.ga[ch] : 0.f;
        ba[e] = ch < c ? a.ba[ch] : 0.f;
        gb[e] = (a.gb && ch < c) ? a.gb[ch] : 0.f;
        sga[e] = sba[e] = sgb[e] = sbb[e] = sbias[e] = 0.f;
    }
    for (long row = r0; row < r1; ++row) {
        const float mean_a = a.stats[row * 4 + 0], rstd_a = a.stats[row * 4 + 1];
        float xh[EPL], g[EPL];  // xhat of LN_a, incoming gradient
        bool kp[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int ch = lane + 64 * e;
            float u = 0.f;
            kp[e] = true;
            if (ch < c) {
                u = a.x[row * c + ch];
                if (a.y) {
                    float d = a.y[row * c + ch] + (a.bias ? a.bias[ch] : 0.f);
                    if (thresh) { kp[e] = keep(seed, a.site, (unsigned long long)row * c + ch, thresh); d = kp[e] ? d * scale : 0.f; }
                    u += d;
                }
            }
            xh[e] = ch < c ? (u - mean_a) * rstd_a : 0.f;
            g[e] = ch < c ? b.dout[row * c + ch] : 0.f;
        }
        if (a.gb) {  // through LN_b: its input is n1 = xhat_a * ga + ba
            const float mean_b = a.stats[row * 4 + 2], rstd_b = a.stats[row * 4 + 3];
            float s1 = 0.f, s2 = 0.f, xb[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const bool in = lane + 64 * e < c;
                xb[e] = in ? ((xh[e] * ga[e] + ba[e]) - mean_b) * rstd_b : 0.f;
                sgb[e] += g[e] * xb[e];
                sbb[e] += g[e];
                g[e] *= gb[e];                 // d xhat_b
                s1 += g[e];
                s2 += g[e] * xb[e];
            }
            s1 = wsum(s1) * inv_c;
            s2 = wsum(s2) * inv_c;
#pragma unroll
            for (int e = 0; e < EPL; ++e) g[e] = (lane + 64 * e < c) ? rstd_b * (g[e] - s1 - xb[e] * s2) : 0.f;  // d n1
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            sga[e] += g[e] * xh[e];
            sba[e] += g[e];
            g[e] *= ga[e];                     // d xhat_a
            s1 += g[e];
            s2 += g[e] * xh[e];
        }
        s1 = wsum(s1) * inv_c;
        s2 = wsum(s2) * inv_c;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int ch = lane + 64 * e;
            if (ch >= c) continue;
            const float du = rstd_a * (g[e] - s1 - xh[e] * s2);
            b.dx[row * c + ch] = du;
            if (b.dy) {
                const float dyb = (thresh ? (kp[e] ? du * scale : 0.f) : du);
                b.dy[row * c + ch] = dyb;
                sbias[e] += dyb;
            }
        }
    }
    // the four waves' column sums meet in LDS: one atomic per column, accumulator and workgroup
    __shared__ float red[4][5][64 * EPL];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int ch = lane + 64 * e;
        red[w][0][ch] = sga[e]; red[w][1][ch] = sba[e]; red[w][2][ch] = sgb[e]; red[w][3][ch] = sbb[e]; red[w][4][ch] = sbias[e];
    }
    __syncthreads();
    float *const dst[5] = {b.dga, b.dba, b.dgb, b.dbb, b.dbias};
    for (int i = threadIdx.x; i < 5 * c; i += 256) {
        const int k = i / c, ch = i - k * c;
        if (dst[k]) atomicAdd(dst[k] + ch, (red[0][k][ch] + red[1][k][ch]) + (red[2][k][ch] + red[3][k][ch]));
    }
}

// h = dropout(relu(z + bias)); element-wise over (rows, c)
__global__ void __launch_bounds__(256)
tail_relu_drop_fwd_kernel(long n, int c, const float *__restrict__ z, const float *__restrict__ bias, float p, unsigned site,
                          const long long *__restrict__ seed_in, float *__restrict__ out) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const unsigned thresh = drop_threshold(p);
    const unsigned long long seed = thresh ? (unsigned long long)seed_in[0] : 0ull;
    const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
    const float4 v = *reinterpret_cast<const float4 *>(z + i);
    const int ch = (int)(i % c);  // c % 4 == 0: the four elements share a row
    const float4 bv = bias ? *reinterpret_cast<const float4 *>(bias + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    float h[4] = {v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = !(h[e] <= 0.f) ? h[e] : 0.f;  // NaN propagates like torch's relu
        if (thresh) h[e] = keep(seed, site, (unsigned long long)(i + e), thresh) ? h[e] * scale : 0.f;
    }
    *reinterpret_cast<float4 *>(out + i) = make_float4(h[0], h[1], h[2], h[3]);
}

// dz = dh * mask * [z + bias > 0]; dbias += column sums.  A workgroup owns 16 rows x 256 columns: wave w takes rows 4w .. 4w+3,
// lane l the column quad 4l of the tile (four independent 16-byte load pairs in flight); the four waves' column sums meet in LDS
// and leave as ONE atomic per column and workgroup.
__global__ void __launch_bounds__(256)
tail_relu_drop_bwd_kernel(long rows, int c, const float *__restrict__ z, const float *__restrict__ bias, float p, unsigned site,
                          const long long *__restrict__ seed_in, const float *__restrict__ dh, float *__restrict__ dz,
                          float *__restrict__ dbias) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q0 = blockIdx.y * 256 + lane * 4;
    const long r0 = (long)blockIdx.x * 16 + w * 4;
    const unsigned thresh = drop_threshold(p);
    const unsigned long long seed = thresh ? (unsigned long long)seed_in[0] : 0ull;
    const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (q0 < c) {
        const float4 bv = bias ? *reinterpret_cast<const float4 *>(bias + q0) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
        float4 zv[4], gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = r0 + r < rows ? r0 + r : rows - 1;
            zv[r] = *reinterpret_cast<const float4 *>(z + row * c + q0);
            gv[r] = *reinterpret_cast<const float4 *>(dh + row * c + q0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r0 + r >= rows) break;
            const long i = (r0 + r) * c + q0;
            const float zz[4] = {zv[r].x, zv[r].y, zv[r].z, zv[r].w}, gg[4] = {gv[r].x, gv[r].y, gv[r].z, gv[r].w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g = (zz[e] + bb[e] > 0.f) ? gg[e] : 0.f;
                if (thresh) g = keep(seed, site, (unsigned long long)(i + e), thresh) ? g * scale : 0.f;
                o[e] = g;
                acc[e] += g;
            }
            *reinterpret_cast<float4 *>(dz + i) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (!dbias) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) red[w][lane * 4 + e] = acc[e];
    __syncthreads();
    const int col = blockIdx.y * 256 + threadIdx.x;
    if (col < c) atomicAdd(dbias + col, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

static int epl_of(int c) { return c <= 128 ? 2 : (c <= 256 ? 4 : (c <= 512 ? 8 : (c <= 1024 ? 16 : 0))); }

// ---- the last Conv1d of final_mlp + residual on the initial keypoints + de-canonicalisation (hand_network.py:141-147) ---------
// forward: one wave per token (b, j).  kp_hand[b, :, j] = h[tok] . w^T + bias + xyz1[b, :, j]  (B, 3, J), channel-major as the
// reference's tensors; kp_cam[b, j, :] = scale[b] R_b kp_hand[b, :, j] + t_b  (B, J, 3).  Replaces a GEMM with a 3-wide output, the
// residual add, a batched 3 x 3 matmul, scale / translate and two layout copies (8 launches); the same for its backward (6).
__global__ void __launch_bounds__(256)
pose_head_fwd_kernel(int tokens, int j, int c, const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ bias,
                     const float *__restrict__ xyz1, const float *__restrict__ R, const float *__restrict__ t,
                     const float *__restrict__ scale, int sstride, float *__restrict__ kp_hand, float *__restrict__ kp_cam) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= tokens) return;
    const int lane = threadIdx.x & 63;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int ch = lane; ch < c; ch += 64) {
        const float hv = h[(size_t)tok * c + ch];
        a0 = fmaf(hv, w[ch], a0);
        a1 = fmaf(hv, w[c + ch], a1);
        a2 = fmaf(hv, w[2 * c + ch], a2);
    }
    a0 = wsum(a0); a1 = wsum(a1); a2 = wsum(a2);
    if (lane == 0) {
        const int b = tok / j, jj = tok - b * j;
        const float *x = xyz1 + (size_t)b * 3 * j + jj;
        float *o = kp_hand + (size_t)b * 3 * j + jj;
        const float px = a0 + bias[0] + x[0], py = a1 + bias[1] + x[j], pz = a2 + bias[2] + x[2 * j];
        o[0] = px; o[j] = py; o[2 * j] = pz;
        const float *Rb = R + 9 * b, *tb = t + 3 * b;
        const float sc = scale[(size_t)b * sstride];  // sstride 0: one scale for every cloud
        kp_cam[3 * tok] = sc * (Rb[0] * px + Rb[1] * py + Rb[2] * pz) + tb[0];  // scale * (R p) + t, decanonicalize's order
        kp_cam[3 * tok + 1] = sc * (Rb[3] * px + Rb[4] * py + Rb[5] * pz) + tb[1];
        kp_cam[3 * tok + 2] = sc * (Rb[6] * px + Rb[7] * py + Rb[8] * pz) + tb[2];
    }
}

// backward: g = d kp_hand (B, 3, J) [+ scale R^T d kp_cam].  A workgroup takes 32 tokens, a thread one input channel:
// dh[tok, ch] = sum_i g_i w[i, ch];  dw[i, ch] += sum_tok g_i h[tok, ch], dbias[i] += sum_tok g_i  (zero-filled accumulators).
__global__ void __launch_bounds__(256)
pose_head_bwd_kernel(int tokens, int j, int c, const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ g_hand,
                     const float *__restrict__ g_cam, const float *__restrict__ R, const float *__restrict__ scale, int sstride,
                     float *__restrict__ dh, float *__restrict__ dw, float *__restrict__ dbias) {
    __shared__ float gs[32][3];
    const int tok0 = blockIdx.x * 32;
    if (threadIdx.x < 96) {
        const int tl = threadIdx.x / 3, i = threadIdx.x % 3, tok = tok0 + tl;
        float g = 0.f;
        if (tok < tokens) {
            const int b = tok / j, jj = tok - b * j;
            g = g_hand ? g_hand[(size_t)b * 3 * j + (size_t)i * j + jj] : 0.f;
            if (g_cam) {
                const float *Rb = R + 9 * b, *gc = g_cam + 3 * (size_t)tok;
                g += scale[(size_t)b * sstride] * (Rb[i] * gc[0] + Rb[3 + i] * gc[1] + Rb[6 + i] * gc[2]);
            }
        }
        gs[tl][i] = g;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        const float w0 = w[ch], w1 = w[c + ch], w2 = w[2 * c + ch];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int tl = 0; tl < 32 && tok0 + tl < tokens; ++tl) {
            const float g0 = gs[tl][0], g1 = gs[tl][1], g2 = gs[tl][2];
            const size_t o = (size_t)(tok0 + tl) * c + ch;
            dh[o] = g0 * w0 + g1 * w1 + g2 * w2;
            const float hv = h[o];
            a0 += g0 * hv; a1 += g1 * hv; a2 += g2 * hv;
        }
        atomicAdd(dw + ch, a0);
        atomicAdd(dw + c + ch, a1);
        atomicAdd(dw + 2 * c + ch, a2);
    }
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (int tl = 0; tl < 32; ++tl) s += gs[tl][threadIdx.x];
        atomicAdd(dbias + threadIdx.x, s);
    }
}

}  // namespace tt
}  // namespace pn2

using namespace pn2;
using namespace pn2::tt;

extern "C" int pn2x_tail_ln_fwd(long rows, int c, const float *x, const float *y, const float *bias, float p, int site,
                                const long long *seed_in, long long *seed_dev, long long *seed_out, const float *ga, const float *ba,
                                float eps_a, const float *gb, const float *bb, float eps_b, float *out, float *stats, void *stream) {
    if (rows < 0 || c < 1 || epl_of(c) == 0 || !(p >= 0.f && p < 1.f)) return PN2_EINVAL;
    if (rows == 0) return PN2_OK;
    if (!x || !ga || !ba || !out || !stats || ((gb == nullptr) != (bb == nullptr))) return PN2_ENULL;
    if ((y && p > 0.f && !seed_in) || (seed_dev && !seed_out)) return PN2_ENULL;
    LnArgs a{rows, c, x, y, bias, p, (unsigned)site, seed_in, seed_dev, seed_out, ga, ba, eps_a, gb, bb, eps_b, out, stats};
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    switch (epl_of(c)) {
        case 2: hipLaunchKernelGGL(tail_ln_fwd_kernel<2>, dim3(blocks), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(tail_ln_fwd_kernel<4>, dim3(blocks), dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL(tail_ln_fwd_kernel<8>, dim3(blocks), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(tail_ln_fwd_kernel<16>, dim3(blocks), dim3(256), 0, st, a); break;
    }
    return check_launch();
}

extern "C" int pn2x_tail_ln_bwd(long rows, int c, const float *x, const float *y, const float *bias, float p, int site,
                                const long long *seed_in, const float *ga, const float *ba, float eps_a, const float *gb,
                                const float *bb, float eps_b, const float *stats, const float *dout, float *dx, float *dy, float *dga,
                                float *dba, float *dgb, float *dbb, float *dbias, void *stream) {
    if (rows < 0 || c < 1 || epl_of(c) == 0 || epl_of(c) > 8 || !(p >= 0.f && p < 1.f)) return PN2_EINVAL;
    if (rows == 0) return PN2_OK;
    if (!x || !ga || !ba || !stats || !dout || !dx || !dga || !dba || (gb && (!dgb || !dbb)) || (y && !dy)) return PN2_ENULL;
    if (y && p > 0.f && !seed_in) return PN2_ENULL;
    LnBwdArgs b{LnArgs{rows, c, x, y, bias, p, (unsigned)site, seed_in, nullptr, nullptr, ga, ba, eps_a, gb, bb, eps_b, nullptr,
                       const_cast<float *>(stats)},
                dout, dx, dy, dga, dba, gb ? dgb : nullptr, gb ? dbb : nullptr, (y && bias) ? dbias : nullptr, 2};
    const long waves = (rows + b.rows_per_wave - 1) / b.rows_per_wave;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    switch (epl_of(c)) {
        case 2: hipLaunchKernelGGL(tail_ln_bwd_kernel<2>, dim3(blocks), dim3(256), 0, st, b); break;
        case 4: hipLaunchKernelGGL(tail_ln_bwd_kernel<4>, dim3(blocks), dim3(256), 0, st, b); break;
        default: hipLaunchKernelGGL(tail_ln_bwd_kernel<8>, dim3(blocks), dim3(256), 0, st, b); break;
    }
    return check_launch();
}

extern "C" int pn2x_tail_relu_drop_fwd(long rows, int c, const float *z, const float *bias, float p, int site, const long long *seed_in,
                                       float *out, void *stream) {
    if (rows < 0 || c < 4 || c % 4 || !(p >= 0.f && p < 1.f)) return PN2_EINVAL;
    if (rows == 0) return PN2_OK;
    if (!z || !out || (p > 0.f && !seed_in)) return PN2_ENULL;
    if (((uintptr_t)z | (uintptr_t)out | (uintptr_t)bias) % 16) return PN2_EINVAL;
    const long n = rows * c;
    hipLaunchKernelGGL(tail_relu_drop_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, c, z, bias, p,
                       (unsigned)site, seed_in, out);
    return check_launch();
}

extern "C" int pn2x_tail_relu_drop_bwd(long rows, int c, const float *z, const float *bias, float p, int site, const long long *seed_in,
                                       const float *dh, float *dz, float *dbias, void *stream) {
    if (rows < 0 || c < 4 || c % 4 || !(p >= 0.f && p < 1.f)) return PN2_EINVAL;
    if (rows == 0) return PN2_OK;
    if (!z || !dh || !dz || (p > 0.f && !seed_in)) return PN2_ENULL;
    if (((uintptr_t)z | (uintptr_t)dh | (uintptr_t)dz | (uintptr_t)bias) % 16) return PN2_EINVAL;
    hipLaunchKernelGGL(tail_relu_drop_bwd_kernel, dim3((unsigned)((rows + 15) / 16), (unsigned)((c + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rows, c, z, bias, p, (unsigned)site, seed_in, dh, dz, dbias);
    return check_launch();
}

extern "C" int pn2x_tail_pose_head_fwd(int b, int j, int c, const float *h, const float *w, const float *bias, const float *xyz1,
                                       const float *R, const float *t, const float *scale, int scale_stride, float *kp_hand,
                                       float *kp_cam, void *stream) {
    if (b < 0 || j < 1 || c < 1) return PN2_EINVAL;
    if (b == 0) return PN2_OK;
    if (!h || !w || !bias || !xyz1 || !R || !t || !scale || !kp_hand || !kp_cam) return PN2_ENULL;
    const int tokens = b * j;
    hipLaunchKernelGGL(pose_head_fwd_kernel, dim3((tokens + 3) / 4), dim3(256), 0, (hipStream_t)stream, tokens, j, c, h, w, bias, xyz1, R,
                       t, scale, scale_stride ? 1 : 0, kp_hand, kp_cam);
    return check_launch();
}

extern "C" int pn2x_tail_pose_head_bwd(int b, int j, int c, const float *h, const float *w, const float *g_hand, const float *g_cam,
                                       const float *R, const float *scale, int scale_stride, float *dh, float *dw, float *dbias,
                                       void *stream) {
    if (b < 0 || j < 1 || c < 1) return PN2_EINVAL;
    if (b == 0) return PN2_OK;
    if (!h || !w || !dh || !dw || !dbias || (!g_hand && !g_cam) || (g_cam && (!R || !scale))) return PN2_ENULL;
    const int tokens = b * j;
    hipLaunchKernelGGL(pose_head_bwd_kernel, dim3((tokens + 31) / 32), dim3(256), 0, (hipStream_t)stream, tokens, j, c, h, w, g_hand, g_cam,
                       R, scale, scale_stride ? 1 : 0, dh, dw, dbias);
    return check_launch();
}
