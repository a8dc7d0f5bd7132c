// tail.hip -- the 21-token tail of HandTrackNet (hand_network.py:139-147 with attn=False): after the keypoint
// branches only LayerNorms, two small FFNs and the 3-channel head remain.  On the GPU each torch op there is a
// launch-bound ~5 us kernel over a few KB; these kernels fuse the element-wise runs between the GEMMs:
//   add_layernorm : out = LN2( LN1( x + y + bias ) )      (residual add + bias + one or two LayerNorms, 1 launch)
//   pose_head     : delta = h W^T + b ; kp_hand = delta + xyz1 ; kp_cam = (kp_hand R^T) * scale + t   (1 launch for
//                   the reference's Conv1d(…,3,1) + add + matmul + mul + add, hand_network.py:141-147)
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_f32(v); }  // DPP ladder, no ds_bpermute round trips

// One wave per row; C <= 64 * EPL.  torch.nn.functional.layer_norm semantics: biased variance, eps inside the sqrt.
template <int EPL>
__global__ void __launch_bounds__(256)
add_layernorm_kernel(long rows, int c, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ bias,
                     const float *__restrict__ g1, const float *__restrict__ b1, float eps1, const float *__restrict__ g2,
                     const float *__restrict__ b2, float eps2, float *__restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float v[EPL];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int ch = lane + 64 * e;
        float t = 0.f;
        if (ch < c) {
            t = x[row * c + ch];
            if (y) t += y[row * c + ch];
            if (bias) t += bias[ch];
        }
        v[e] = t;
        s += t;
    }
    const float inv_c = 1.0f / (float)c;
    float mean = wave_sum(s) * inv_c, q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float d = (lane + 64 * e < c) ? v[e] - mean : 0.f;
        q += d * d;
    }
    float rstd = rsqrtf(wave_sum(q) * inv_c + eps1);
    s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int ch = lane + 64 * e;
        v[e] = ch < c ? (v[e] - mean) * rstd * g1[ch] + b1[ch] : 0.f;
        s += v[e];
    }
    if (g2) {  // a second LayerNorm straight after the first (TransT.s11 -> c11.norm1)
        mean = wave_sum(s) * inv_c;
        q = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float d = (lane + 64 * e < c) ? v[e] - mean : 0.f;
            q += d * d;
        }
        rstd = rsqrtf(wave_sum(q) * inv_c + eps2);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int ch = lane + 64 * e;
            if (ch < c) v[e] = (v[e] - mean) * rstd * g2[ch] + b2[ch];
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int ch = lane + 64 * e;
        if (ch < c) out[row * c + ch] = v[e];
    }
}

// One wave per token (b, j): three dot products of length c, then the rigid transform back to the camera frame.
__global__ void __launch_bounds__(256)
pose_head_kernel(int tokens, int j, int c, const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ bias,
                 const float *__restrict__ xyz1, const float *__restrict__ R, const float *__restrict__ t, float scale,
                 float *__restrict__ kp_hand, float *__restrict__ kp_cam, const int *__restrict__ nonfinite) {
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
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) {
        const int b = tok / j;
        float px = a0 + bias[0] + xyz1[3 * tok], py = a1 + bias[1] + xyz1[3 * tok + 1], pz = a2 + bias[2] + xyz1[3 * tok + 2];
        if (nonfinite && nonfinite[b]) px = py = pz = __builtin_nanf("");  // flagged by pn2x_hand_frame3: non-finite input frame
        kp_hand[3 * tok] = px; kp_hand[3 * tok + 1] = py; kp_hand[3 * tok + 2] = pz;
        const float *Rb = R + 9 * b, *tb = t + 3 * b;
        // row vector times R^T: out_i = sum_k p_k R[i][k]
        kp_cam[3 * tok] = (px * Rb[0] + py * Rb[1] + pz * Rb[2]) * scale + tb[0];
        kp_cam[3 * tok + 1] = (px * Rb[3] + py * Rb[4] + pz * Rb[5]) * scale + tb[1];
        kp_cam[3 * tok + 2] = (px * Rb[6] + py * Rb[7] + pz * Rb[8]) * scale + tb[2];
    }
}

}  // namespace pn2

using namespace pn2;

extern "C" int pn2x_add_layernorm(long rows, int c, const float *x, const float *y, const float *bias, const float *g1,
                                  const float *b1, float eps1, const float *g2, const float *b2, float eps2, float *out,
                                  void *stream) {
    if (rows < 0 || c < 1 || c > 1024) return PN2_EINVAL;
    if (rows == 0) return PN2_OK;
    if (!x || !g1 || !b1 || !out || ((g2 == nullptr) != (b2 == nullptr))) return PN2_ENULL;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define PN2_LN(E) hipLaunchKernelGGL(add_layernorm_kernel<E>, dim3(blocks), dim3(256), 0, st, rows, c, x, y, bias, g1, b1, eps1, g2, b2, eps2, out)
    if (c <= 128) PN2_LN(2);
    else if (c <= 256) PN2_LN(4);
    else if (c <= 512) PN2_LN(8);
    else PN2_LN(16);
#undef PN2_LN
    return check_launch();
}

extern "C" int pn2x_pose_head(int b, int j, int c, const float *h, const float *w, const float *bias, const float *xyz1,
                              const float *R, const float *t, float scale, float *kp_hand, float *kp_cam, void *stream) {
    return pn2x_pose_head2(b, j, c, h, w, bias, xyz1, R, t, scale, kp_hand, kp_cam, nullptr, stream);
}

extern "C" int pn2x_pose_head2(int b, int j, int c, const float *h, const float *w, const float *bias, const float *xyz1,
                               const float *R, const float *t, float scale, float *kp_hand, float *kp_cam, const int *nonfinite,
                               void *stream) {
    if (b < 0 || j < 1 || c < 1) return PN2_EINVAL;
    if (b == 0) return PN2_OK;
    if (!h || !w || !bias || !xyz1 || !R || !t || !kp_hand || !kp_cam) return PN2_ENULL;
    const int tokens = b * j;
    hipLaunchKernelGGL(pose_head_kernel, dim3((tokens + 3) / 4), dim3(256), 0, (hipStream_t)stream, tokens, j, c, h, w, bias, xyz1,
                       R, t, scale, kp_hand, kp_cam, nonfinite);
    return check_launch();
}
