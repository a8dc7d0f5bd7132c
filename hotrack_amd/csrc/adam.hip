// adam.hip -- multi-tensor Adam step for the training entry point (network/train.py, reference trainer.py:49-52:
// torch.optim.Adam(lr, betas=(0.9, 0.999), eps, weight_decay) -- L2 weight decay folded into the gradient, no amsgrad).
//
// torch's own fused Adam (multi_tensor_apply) moves the 4.17 M parameters that receive a gradient in this network at
// 0.66 TB/s (four launches, 178 us of a 4.8 ms training step); the update is a pure stream -- read p, g, m, v, write p, m, v:
// 28 bytes per parameter -- so here every workgroup takes one 8192-element chunk of one tensor with 16-byte accesses and the
// tensor table travels in the kernel arguments (no device-side table to keep coherent, capture-safe).  The arithmetic is the
// one of torch's kernel (fused_adam_utils.cuh: lerp for the first moment, sqrt(v) / sqrt(bias_correction2) + eps), the
// per-parameter `step` tensors keep torch's state_dict layout and are advanced by a second, tiny launch.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
namespace adam {

constexpr int kMaxT = 64;      // tensors per launch (kernel-argument budget)
constexpr int kChunk = 2048;   // elements per workgroup (round 5: 8192 gave 115 - 384 workgroups per launch on 256 CUs -- ragged rounds)
constexpr int kT = 256;

struct Pack {
    float *p[kMaxT];
    const float *g[kMaxT];
    float *m[kMaxT];
    float *v[kMaxT];
    float *step[kMaxT];
    int chunk_start[kMaxT + 1];  // first workgroup of every tensor
    long numel[kMaxT];
    int n;
};

__global__ void __launch_bounds__(kT)
adam_kernel(Pack a, double lr_d, double beta1_d, double beta2_d, double eps_d, double wd_d) {
    int t = 0;
    const int wg = blockIdx.x;
    while (t + 1 < a.n && a.chunk_start[t + 1] <= wg) ++t;  // <= 64 wave-uniform compares
    const long off = (long)(wg - a.chunk_start[t]) * kChunk;
    const long n = a.numel[t];
    const long end = off + kChunk < n ? off + kChunk : n;
    // scalar constants in double like torch's kernel (its hyper-parameters are doubles: 1 - 0.999f is off by 1.3e-5 relative)
    // the bias corrections need two fp64 pow() and a sqrt(): ONE thread of the workgroup computes them (every thread doing so cost
    // more than the 8 elements it then updates: a workgroup moves 57 KB and spent its time in the fp64 library code)
    __shared__ float bias_corr[2];
    if (threadIdx.x == 0) {
        const double step = (double)a.step[t][0] + 1.0;  // the counters are advanced after all update kernels (adam_step_kernel)
        const double bc1 = 1.0 - pow(beta1_d, step), bc2 = 1.0 - pow(beta2_d, step);
        bias_corr[0] = (float)(lr_d / bc1);
        bias_corr[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = bias_corr[0], bc2_sqrt = bias_corr[1];
    const float beta2 = (float)beta2_d, omb1 = (float)(1.0 - beta1_d), omb2 = (float)(1.0 - beta2_d), eps = (float)eps_d, wd = (float)wd_d;
    float *__restrict__ p = a.p[t];
    const float *__restrict__ g = a.g[t];
    float *__restrict__ m = a.m[t];
    float *__restrict__ v = a.v[t];
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        if (wd != 0.f) gg += pp * wd;
        mm = mm + (gg - mm) * omb1;                    // torch: lerp(exp_avg, grad, 1 - beta1)
        vv = beta2 * vv + omb2 * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp -= step_size * mm / denom;
    };
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    long i = off + 4L * threadIdx.x;
    if (vec) {
        // two quads per thread and step: eight 16-byte loads in flight before the first dependent store
        for (; i + 4L * kT + 3 < end; i += 8L * kT) {
            const long k = i + 4L * kT;
            float4 p0 = *reinterpret_cast<float4 *>(p + i), m0 = *reinterpret_cast<float4 *>(m + i), v0 = *reinterpret_cast<float4 *>(v + i);
            const float4 g0 = *reinterpret_cast<const float4 *>(g + i);
            float4 p1 = *reinterpret_cast<float4 *>(p + k), m1 = *reinterpret_cast<float4 *>(m + k), v1 = *reinterpret_cast<float4 *>(v + k);
            const float4 g1 = *reinterpret_cast<const float4 *>(g + k);
            upd(p0.x, g0.x, m0.x, v0.x); upd(p0.y, g0.y, m0.y, v0.y); upd(p0.z, g0.z, m0.z, v0.z); upd(p0.w, g0.w, m0.w, v0.w);
            upd(p1.x, g1.x, m1.x, v1.x); upd(p1.y, g1.y, m1.y, v1.y); upd(p1.z, g1.z, m1.z, v1.z); upd(p1.w, g1.w, m1.w, v1.w);
            *reinterpret_cast<float4 *>(p + i) = p0;
            *reinterpret_cast<float4 *>(m + i) = m0;
            *reinterpret_cast<float4 *>(v + i) = v0;
            *reinterpret_cast<float4 *>(p + k) = p1;
            *reinterpret_cast<float4 *>(m + k) = m1;
            *reinterpret_cast<float4 *>(v + k) = v1;
        }
        for (; i + 3 < end; i += 4L * kT) {
            float4 pp = *reinterpret_cast<float4 *>(p + i), mm = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
            const float4 gg = *reinterpret_cast<const float4 *>(g + i);
            upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
            *reinterpret_cast<float4 *>(p + i) = pp;
            *reinterpret_cast<float4 *>(m + i) = mm;
            *reinterpret_cast<float4 *>(v + i) = vv;
        }
        // tail of the tensor (< 4 elements): the thread whose quad crosses `end`
        if (i < end && i + 3 >= end)
            for (long j = i; j < end; ++j) upd(p[j], g[j], m[j], v[j]);
    } else {
        for (long j = off + threadIdx.x; j < end; j += kT) upd(p[j], g[j], m[j], v[j]);
    }
}

__global__ void adam_step_kernel(Pack a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < a.n) a.step[t][0] += 1.f;
}

__global__ void adam_advance_kernel(float *__restrict__ steps, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) steps[t] += 1.f;
}

}  // namespace adam
}  // namespace pn2

// One Adam step over n tensors (fp32, contiguous).  p / g / m / v / step: host arrays of n device pointers (step: one fp32
// counter per tensor, torch's `state['step']` of a capturable optimiser); numel: host array of n element counts.
extern "C" int pn2x_adam_multi(int n, void *const *p, const void *const *g, void *const *m, void *const *v, void *const *step,
                               const long *numel, double lr, double beta1, double beta2, double eps, double weight_decay, void *stream) {
    return pn2x_adam_multi2(n, p, g, m, v, step, numel, lr, beta1, beta2, eps, weight_decay, 1, stream);
}

// All step counters of an optimiser in ONE contiguous buffer (views of it as the per-parameter `step` tensors): advanced by one
// launch after the update kernels (pn2x_adam_multi2 with advance = 0) instead of one launch per 64 tensors.
extern "C" int pn2x_adam_advance(float *steps, int n, void *stream) {
    if (n < 0) return PN2_EINVAL;
    if (n == 0) return PN2_OK;
    if (!steps) return PN2_ENULL;
    hipLaunchKernelGGL(pn2::adam::adam_advance_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, steps, n);
    return pn2::check_launch();
}

extern "C" int pn2x_adam_multi2(int n, void *const *p, const void *const *g, void *const *m, void *const *v, void *const *step,
                                const long *numel, double lr, double beta1, double beta2, double eps, double weight_decay, int advance,
                                void *stream) {
    using namespace pn2;
    using namespace pn2::adam;
    if (n < 0 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return PN2_EINVAL;
    if (n == 0) return PN2_OK;
    if (!p || !g || !m || !v || !step || !numel) return PN2_ENULL;
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < n; t0 += kMaxT) {
        Pack a;
        a.n = (n - t0) < kMaxT ? (n - t0) : kMaxT;
        int chunks = 0;
        for (int i = 0; i < a.n; ++i) {
            if (!p[t0 + i] || !g[t0 + i] || !m[t0 + i] || !v[t0 + i] || !step[t0 + i] || numel[t0 + i] < 0) return PN2_ENULL;
            a.p[i] = (float *)p[t0 + i]; a.g[i] = (const float *)g[t0 + i]; a.m[i] = (float *)m[t0 + i]; a.v[i] = (float *)v[t0 + i];
            a.step[i] = (float *)step[t0 + i];
            a.numel[i] = numel[t0 + i];
            a.chunk_start[i] = chunks;
            chunks += (int)((numel[t0 + i] + kChunk - 1) / kChunk);
        }
        a.chunk_start[a.n] = chunks;
        if (chunks > 0) hipLaunchKernelGGL(adam_kernel, dim3(chunks), dim3(kT), 0, st, a, lr, beta1, beta2, eps, weight_decay);
    }
    for (int t0 = 0; advance && t0 < n; t0 += kMaxT) {  // after ALL update kernels: they read the old counters
        Pack a;
        a.n = (n - t0) < kMaxT ? (n - t0) : kMaxT;
        for (int i = 0; i < a.n; ++i) a.step[i] = (float *)step[t0 + i];
        hipLaunchKernelGGL(adam_step_kernel, dim3(1), dim3(kMaxT), 0, st, a);
    }
    return check_launch();
}
