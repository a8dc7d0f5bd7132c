// epilogue.hip -- y[b,c,n] = act(y[b,c,n] + bias[c]) in place: the bias/ReLU half of an
// eval-mode (BatchNorm-folded) 1x1 convolution whose GEMM half is a plain library GEMM.
// Pure HBM streaming: 16-byte loads/stores, one pass.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {

template <bool RELU, bool VEC>
__global__ void __launch_bounds__(256)
bias_act_kernel(float *__restrict__ y, const float *__restrict__ bias, int C, long n, long total_rows) {
    // rows = b*C + c, each of n contiguous floats
    const long per_row = VEC ? (n >> 2) : n;
    const long total = total_rows * per_row;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / per_row;
        const float bv = bias[row % C];
        if constexpr (VEC) {
            float4 *p = reinterpret_cast<float4 *>(y) + e;
            float4 v = *p;
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *p = v;
        } else {
            float v = y[e] + bv;
            y[e] = RELU ? fmaxf(v, 0.f) : v;
        }
    }
}

}  // namespace pn2

extern "C" int pn2x_bias_act(int b, int c, int n, float *y, const float *bias, int relu, void *stream) {
    using namespace pn2;
    if (b < 0 || c < 0 || n < 0) return PN2_EINVAL;
    if (b == 0 || c == 0 || n == 0) return PN2_OK;
    if (!y || !bias) return PN2_ENULL;
    const bool vec = (n % 4 == 0) && ((uintptr_t)y % 16 == 0);
    const long rows = (long)b * c;
    const long work = rows * (vec ? n / 4 : n);
    long blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (vec) {
        if (relu) hipLaunchKernelGGL((bias_act_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, st, y, bias, c, (long)n, rows);
        else hipLaunchKernelGGL((bias_act_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, y, bias, c, (long)n, rows);
    } else {
        if (relu) hipLaunchKernelGGL((bias_act_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, st, y, bias, c, (long)n, rows);
        else hipLaunchKernelGGL((bias_act_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, st, y, bias, c, (long)n, rows);
    }
    return check_launch();
}

namespace pn2 {
__global__ void __launch_bounds__(256)
bias_act_pm_kernel(float *__restrict__ y, int ldy, const float *__restrict__ bias, int c, long rows, long rows_per_bias, int relu) {
    const long total = rows * c;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / c;
        const int ch = (int)(e - r * c);
        float v = y[r * ldy + ch] + bias[(r / rows_per_bias) * c + ch];
        y[r * ldy + ch] = relu ? fmaxf(v, 0.f) : v;
    }
}

__global__ void __launch_bounds__(256)
gather_rows_kernel(int n, int m, int c, const float *__restrict__ src, const int *__restrict__ idx, float *__restrict__ out, long total) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / c;  // b*m + j
        const int ch = (int)(e - row * c);
        const long b = row / m;
        out[e] = src[((size_t)b * n + idx[row]) * c + ch];
    }
}
}  // namespace pn2

extern "C" int pn2x_bias_act_pm(long rows, int c, float *y, int ldy, const float *bias, long rows_per_bias, int relu, void *stream) {
    using namespace pn2;
    if (rows < 0 || c < 0 || ldy < c || rows_per_bias < 1) return PN2_EINVAL;
    if (rows == 0 || c == 0) return PN2_OK;
    if (!y || !bias) return PN2_ENULL;
    long blocks = (rows * c + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bias_act_pm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, ldy, bias, c, rows, rows_per_bias, relu);
    return check_launch();
}

extern "C" int pn2x_gather_rows(int b, int n, int m, int c, const float *src, const int *idx, float *out, void *stream) {
    using namespace pn2;
    if (b < 0 || n < 1 || m < 0 || c < 0) return PN2_EINVAL;
    if (b == 0 || m == 0 || c == 0) return PN2_OK;
    if (!src || !idx || !out) return PN2_ENULL;
    const long total = (long)b * m * c;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, m, c, src, idx, out, total);
    return check_launch();
}

namespace pn2 {
// out[b, ch] = max_r x[b, r, ch]   (x (b, r, c) point-major: lanes walk channels -> coalesced rows).
// Block = 64 channels x 4 row groups; every thread keeps 8 independent loads in flight (one thread per channel
// walking all r rows was a latency chain: 11.8 us for one 128 x 512 cloud), the four partial maxima meet in LDS.
__global__ void __launch_bounds__(256)
max_rows_kernel(int r, int c, const float *__restrict__ x, float *__restrict__ out) {
    __shared__ float part[4][64];
    const int chunks = (c + 63) / 64;
    const int b = blockIdx.x / chunks, ch = (blockIdx.x % chunks) * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float m = -__builtin_inff();
    if (ch < c) {
        const float *p = x + (size_t)b * r * c + ch;
        int i = rg;
        for (; i + 28 < r; i += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(i + 4 * u) * c];
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
        }
        for (; i < r; i += 4) m = fmaxf(m, p[(size_t)i * c]);
    }
    part[rg][threadIdx.x & 63] = m;
    __syncthreads();
    if (rg == 0 && ch < c)
        out[(size_t)b * c + ch] = fmaxf(fmaxf(part[0][threadIdx.x], part[1][threadIdx.x]), fmaxf(part[2][threadIdx.x], part[3][threadIdx.x]));
}
}  // namespace pn2

extern "C" int pn2x_max_rows(int b, int r, int c, const float *x, float *out, void *stream) {
    using namespace pn2;
    if (b < 0 || r < 1 || c < 0) return PN2_EINVAL;
    if (b == 0 || c == 0) return PN2_OK;
    if (!x || !out) return PN2_ENULL;
    const long blocks = (long)b * ((c + 63) / 64);
    if (blocks > 2147483647L) return PN2_ERANGE;
    hipLaunchKernelGGL(max_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, r, c, x, out);
    return check_launch();
}

// ---- several device-to-device copies as ONE launch --------------------------------------------------------------------------------
// The batch hand-over of a graph-captured step (network/trainer.py: a handful of batch leaves + the geometry pack into the captured
// step's static buffers) was five runtime copy launches in round 5 and one torch multi-tensor launch of 17 us in round 6; the
// tensors are a few hundred KB.  Table in the kernel arguments (capture-safe, nothing to keep coherent), a workgroup moves 16 KB.
namespace pn2 {
constexpr int kCopyMax = 24;
constexpr long kCopyChunk = 16384;  // bytes per workgroup
struct CopyPack {
    char *dst[kCopyMax];
    const char *src[kCopyMax];
    long bytes[kCopyMax];
    int first[kCopyMax + 1];  // first workgroup of every copy
    int n;
};
__global__ void __launch_bounds__(256) copy_multi_kernel(CopyPack a) {
    int t = 0;
    const int wg = blockIdx.x;
    while (t + 1 < a.n && a.first[t + 1] <= wg) ++t;
    const long off = (long)(wg - a.first[t]) * kCopyChunk;
    const long n = a.bytes[t];
    const long end = off + kCopyChunk < n ? off + kCopyChunk : n;
    char *__restrict__ d = a.dst[t];
    const char *__restrict__ s = a.src[t];
    if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
        long i = off + 16L * threadIdx.x;
        for (; i + 16 <= end; i += 16L * 256) *reinterpret_cast<float4 *>(d + i) = *reinterpret_cast<const float4 *>(s + i);
        for (; i < end; ++i) d[i] = s[i];  // (the one thread that meets the last, partial quad)
    } else {
        for (long i = off + threadIdx.x; i < end; i += 256) d[i] = s[i];
    }
}
}  // namespace pn2

extern "C" int pn2x_copy_multi_max(void) { return pn2::kCopyMax; }

extern "C" int pn2x_copy_multi(int n, void *const *dst, const void *const *src, const long *bytes, void *stream) {
    using namespace pn2;
    if (n < 0 || n > kCopyMax) return PN2_EINVAL;
    if (n == 0) return PN2_OK;
    if (!dst || !src || !bytes) return PN2_ENULL;
    CopyPack a;
    a.n = n;
    long wgs = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] < 0) return PN2_EINVAL;
        if (bytes[i] > 0 && (!dst[i] || !src[i])) return PN2_ENULL;
        a.dst[i] = (char *)dst[i];
        a.src[i] = (const char *)src[i];
        a.bytes[i] = bytes[i];
        a.first[i] = (int)wgs;
        wgs += (bytes[i] + kCopyChunk - 1) / kCopyChunk;
    }
    a.first[n] = (int)wgs;
    if (wgs > 2147483647L) return PN2_ERANGE;
    if (wgs == 0) return PN2_OK;
    hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}
