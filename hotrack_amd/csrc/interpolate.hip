// interpolate.hip -- three_interpolate forward / backward for gfx950.
//
// Replaces three_interpolate_kernel_fast / _grad_ (reference interpolate_gpu.cu:149-233).
// Same streaming structure as gather_group.hip: a thread owns one output point, loads its
// three (index, weight) pairs once and walks a channel range with coalesced stores; the
// backward accumulates into an LDS-private [CC][M] slab instead of 3 global atomics/element.
#include "pn2_common.h"

namespace pn2 {

constexpr int kIpThreads = 256;

__global__ void __launch_bounds__(kIpThreads)
interp_fwd_kernel(int c, int m, int n, int c_per_block, const float *__restrict__ points_all,
                  const int *__restrict__ idx_all, const float *__restrict__ weight_all,
                  float *__restrict__ out_all) {
    const int b = blockIdx.z;
    const int j = blockIdx.x * kIpThreads + threadIdx.x;
    if (j >= n) return;
    const int c0 = blockIdx.y * c_per_block;
    const int c1 = (c0 + c_per_block) < c ? (c0 + c_per_block) : c;
    const int *__restrict__ id = idx_all + ((size_t)b * n + j) * 3;
    const float *__restrict__ w = weight_all + ((size_t)b * n + j) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float *__restrict__ src = points_all + ((size_t)b * c + c0) * m;
    float *__restrict__ dst = out_all + ((size_t)b * c + c0) * n + j;
#pragma unroll 4
    for (int ch = c0; ch < c1; ++ch) {
        // w0*p0 + w1*p1 + w2*p2 (interpolate_gpu.cu:168) in the oracle's contraction order
        dst[0] = __builtin_fmaf(w2, src[i2], __builtin_fmaf(w0, src[i0], w1 * src[i1]));
        src += m;
        dst += n;
    }
}

// LDS-staged variant: the three gathers per output element are the kernel's cost when they are vector-memory instructions
// (12 global loads per 16 bytes stored); the rows they read are short (m floats per channel), so a workgroup stages CC rows of
// its cloud in LDS once and every gather becomes a ds_read.  A thread owns 4 consecutive queries: three 16-byte loads each for
// their indices and weights, one 16-byte coalesced store per channel.
__global__ void __launch_bounds__(kIpThreads)
interp_fwd_lds_kernel(int c, int m, int n, int cc, int j_per_block, const float *__restrict__ points_all,
                      const int *__restrict__ idx_all, const float *__restrict__ weight_all, float *__restrict__ out_all) {
    extern __shared__ __attribute__((aligned(16))) float rows[];  // [cc][m]
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * cc;
    const int nc = (c - c0) < cc ? (c - c0) : cc;
    const float *__restrict__ src = points_all + ((size_t)b * c + c0) * m;
    for (int i = threadIdx.x; i < nc * m; i += kIpThreads) rows[i] = src[i];
    __syncthreads();
    const int j_begin = blockIdx.x * j_per_block;
    const int j_end = (j_begin + j_per_block) < n ? (j_begin + j_per_block) : n;
    for (int j0 = j_begin + threadIdx.x * 4; j0 < j_end; j0 += kIpThreads * 4) {  // n % 4 == 0, j_per_block % 4 == 0
        const int4 *ip = reinterpret_cast<const int4 *>(idx_all + ((size_t)b * n + j0) * 3);
        const float4 *wp = reinterpret_cast<const float4 *>(weight_all + ((size_t)b * n + j0) * 3);
        int4 ia = ip[0], ib = ip[1], ic = ip[2];          // (i0 i1 i2 | i0) (i1 i2 | i0 i1) (i2 | i0 i1 i2) of queries j0 .. j0+3
        ia.x = lds_index(ia.x, m); ia.y = lds_index(ia.y, m); ia.z = lds_index(ia.z, m); ia.w = lds_index(ia.w, m);
        ib.x = lds_index(ib.x, m); ib.y = lds_index(ib.y, m); ib.z = lds_index(ib.z, m); ib.w = lds_index(ib.w, m);
        ic.x = lds_index(ic.x, m); ic.y = lds_index(ic.y, m); ic.z = lds_index(ic.z, m); ic.w = lds_index(ic.w, m);
        const float4 wa = wp[0], wb = wp[1], wc = wp[2];
        const float *r = rows;
        float *dst = out_all + ((size_t)b * c + c0) * n + j0;
#pragma unroll 4
        for (int ch = 0; ch < nc; ++ch) {
            // w0*p0 + w1*p1 + w2*p2 (interpolate_gpu.cu:168) in the oracle's contraction order
            float4 o;
            o.x = __builtin_fmaf(wa.z, r[ia.z], __builtin_fmaf(wa.x, r[ia.x], wa.y * r[ia.y]));
            o.y = __builtin_fmaf(wb.y, r[ib.y], __builtin_fmaf(wa.w, r[ia.w], wb.x * r[ib.x]));
            o.z = __builtin_fmaf(wc.x, r[ic.x], __builtin_fmaf(wb.z, r[ib.z], wb.w * r[ib.w]));
            o.w = __builtin_fmaf(wc.w, r[ic.w], __builtin_fmaf(wc.y, r[ic.y], wc.z * r[ic.z]));
            *reinterpret_cast<float4 *>(dst) = o;
            r += m;
            dst += n;
        }
    }
}

int interp_fwd_dispatch(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                        float *out, hipStream_t st) {
    if (b == 0 || c == 0 || n == 0) return PN2_OK;
    if (n % 4 == 0 && (((uintptr_t)idx | (uintptr_t)weight | (uintptr_t)out) % 16 == 0) && (long)m * 4 * 4 <= 64 * 1024 && n >= 2 * m) {
        int cc = (int)((64L * 1024) / ((long)m * 4));
        if (cc > 32) cc = 32;
        if (cc > c) cc = c;
        // enough workgroups to fill the chip, but every workgroup writes at least twice the elements it stages
        while (cc > 4 && (long)b * ((c + cc - 1) / cc) < 512) cc = (cc + 1) / 2;
        const int ych = (c + cc - 1) / cc;
        int xb = (int)((1024 + (long)ych * b - 1) / ((long)ych * b));
        const int xb_max = n / (2 * m) > 0 ? n / (2 * m) : 1;
        if (xb > xb_max) xb = xb_max;
        if (xb < 1) xb = 1;
        const int j_per_block = ((n + xb - 1) / xb + 3) / 4 * 4;
        xb = (n + j_per_block - 1) / j_per_block;
        hipLaunchKernelGGL(interp_fwd_lds_kernel, dim3(xb, ych, b), dim3(kIpThreads), (size_t)cc * m * sizeof(float), st, c, m, n, cc,
                           j_per_block, points, idx, weight, out);
        return check_launch();
    }
    const int xb = (n + kIpThreads - 1) / kIpThreads;
    int ysplit = (int)((2048 + (long)xb * b - 1) / ((long)xb * b));
    if (ysplit < 1) ysplit = 1;
    if (ysplit > c) ysplit = c;
    const int c_per_block = (c + ysplit - 1) / ysplit;
    ysplit = (c + c_per_block - 1) / c_per_block;
    dim3 grid(xb, ysplit, b);
    hipLaunchKernelGGL(interp_fwd_kernel, grid, dim3(kIpThreads), 0, st, c, m, n, c_per_block, points, idx, weight, out);
    return check_launch();
}

// A work item = (query j, chunk of kIpU channels): its kIpU gradient loads are independent and issued together, then the
// 3 x kIpU LDS adds (see group_bwd_lds_kernel: one load at a time was a latency chain, 0.04 of the HBM roofline).
constexpr int kIpU = 8;
__global__ void __launch_bounds__(kIpThreads)
interp_bwd_lds_kernel(int c, int n, int m, int cc, const float *__restrict__ grad_out_all,
                      const int *__restrict__ idx_all, const float *__restrict__ weight_all,
                      float *__restrict__ grad_points_all) {
    extern __shared__ __attribute__((aligned(16))) float acc[];  // [cc][m]
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * cc;
    const int nc = (c - c0) < cc ? (c - c0) : cc;
    for (int i = threadIdx.x; i < nc * m; i += kIpThreads) acc[i] = 0.f;
    __syncthreads();
    const float *__restrict__ g = grad_out_all + ((size_t)b * c + c0) * n;
    const int chunks = (nc + kIpU - 1) / kIpU;
    for (int item = threadIdx.x; item < n * chunks; item += kIpThreads) {
        const int chunk = item / n, j = item - chunk * n;
        const int ch0 = chunk * kIpU;
        const int *__restrict__ id = idx_all + ((size_t)b * n + j) * 3;
        const float *__restrict__ w = weight_all + ((size_t)b * n + j) * 3;
        const int i0 = lds_index(id[0], m), i1 = lds_index(id[1], m), i2 = lds_index(id[2], m);
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        float v[kIpU];
#pragma unroll
        for (int u = 0; u < kIpU; ++u) v[u] = (ch0 + u < nc) ? g[(size_t)(ch0 + u) * n + j] : 0.f;
#pragma unroll
        for (int u = 0; u < kIpU; ++u) {
            if (ch0 + u < nc) {
                float *a = acc + (ch0 + u) * m;
                atomicAdd(a + i0, v[u] * w0);
                atomicAdd(a + i1, v[u] * w1);
                atomicAdd(a + i2, v[u] * w2);
            }
        }
    }
    __syncthreads();
    float *__restrict__ dst = grad_points_all + ((size_t)b * c + c0) * m;
    const int total = nc * m;
    int i = threadIdx.x;
    for (; i + 3 * kIpThreads < total; i += 4 * kIpThreads) {
        const float d0 = dst[i], d1 = dst[i + kIpThreads], d2 = dst[i + 2 * kIpThreads], d3 = dst[i + 3 * kIpThreads];
        dst[i] = d0 + acc[i];
        dst[i + kIpThreads] = d1 + acc[i + kIpThreads];
        dst[i + 2 * kIpThreads] = d2 + acc[i + 2 * kIpThreads];
        dst[i + 3 * kIpThreads] = d3 + acc[i + 3 * kIpThreads];
    }
    for (; i < total; i += kIpThreads) dst[i] += acc[i];
}

__global__ void __launch_bounds__(kIpThreads)
interp_bwd_atomic_kernel(int c, int n, int m, const float *__restrict__ grad_out_all,
                         const int *__restrict__ idx_all, const float *__restrict__ weight_all,
                         float *__restrict__ grad_points_all) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int j = blockIdx.x * kIpThreads + threadIdx.x;
    if (j >= n) return;
    const int *__restrict__ id = idx_all + ((size_t)b * n + j) * 3;
    const float *__restrict__ w = weight_all + ((size_t)b * n + j) * 3;
    const float go = grad_out_all[((size_t)b * c + ch) * n + j];
    float *__restrict__ dst = grad_points_all + ((size_t)b * c + ch) * m;
    unsafeAtomicAdd(dst + id[0], go * w[0]);
    unsafeAtomicAdd(dst + id[1], go * w[1]);
    unsafeAtomicAdd(dst + id[2], go * w[2]);
}

int interp_bwd_dispatch(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                        float *grad_points, hipStream_t st) {
    if (b == 0 || c == 0 || n == 0) return PN2_OK;
    if (const int rc = scatter_cm_dispatch(3, b, c, m, n, grad_out, idx, weight, grad_points, st); rc != PN2_ERANGE) return rc;  // see scatter_cm.hip; only "shape not covered" falls through
    int cc = (64 * 1024) / (int)(sizeof(float) * (size_t)m);
    if (cc >= 1) {
        if (cc > 16) cc = 16;
        if (cc > c) cc = c;
        while (cc > 1 && (long)b * ((c + cc - 1) / cc) < 1024) cc = (cc + 1) / 2;
        dim3 grid((c + cc - 1) / cc, b);
        hipLaunchKernelGGL(interp_bwd_lds_kernel, grid, dim3(kIpThreads), (size_t)cc * m * sizeof(float), st, c, n, m, cc,
                           grad_out, idx, weight, grad_points);
    } else {
        dim3 grid((n + kIpThreads - 1) / kIpThreads, c, b);
        hipLaunchKernelGGL(interp_bwd_atomic_kernel, grid, dim3(kIpThreads), 0, st, c, n, m, grad_out, idx, weight, grad_points);
    }
    return check_launch();
}

// ---- point-major variant (pn2x_three_interpolate_pm) --------------------------------------------------
// points (b, m, ldp) and out (b, n, ldo) hold one point per ROW: the three source rows are read and the
// destination row written as contiguous 16-byte segments (better coalescing than the channel-major
// reference layout), and `out` may be a column block of a wider buffer (the reference's torch.cat of
// [skip features | interpolated] then costs nothing).
template <bool VEC>
__global__ void __launch_bounds__(256)
interp_pm_kernel(int c, int m, int n, const float *__restrict__ points_all, int ldp, const int *__restrict__ idx_all,
                 const float *__restrict__ weight_all, float *__restrict__ out_all, int ldo, long total) {
    const int per_row = VEC ? (c >> 2) : c;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / per_row;  // = b*n + j
        const int col = (int)(e - row * per_row);
        const long b = row / n;
        const int *__restrict__ id = idx_all + row * 3;
        const float *__restrict__ w = weight_all + row * 3;
        const float *__restrict__ src = points_all + (size_t)b * m * ldp;
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        if constexpr (VEC) {
            const float4 p0 = *reinterpret_cast<const float4 *>(src + (size_t)id[0] * ldp + 4 * col);
            const float4 p1 = *reinterpret_cast<const float4 *>(src + (size_t)id[1] * ldp + 4 * col);
            const float4 p2 = *reinterpret_cast<const float4 *>(src + (size_t)id[2] * ldp + 4 * col);
            float4 o;
            o.x = __builtin_fmaf(w2, p2.x, __builtin_fmaf(w0, p0.x, w1 * p1.x));
            o.y = __builtin_fmaf(w2, p2.y, __builtin_fmaf(w0, p0.y, w1 * p1.y));
            o.z = __builtin_fmaf(w2, p2.z, __builtin_fmaf(w0, p0.z, w1 * p1.z));
            o.w = __builtin_fmaf(w2, p2.w, __builtin_fmaf(w0, p0.w, w1 * p1.w));
            *reinterpret_cast<float4 *>(out_all + (size_t)row * ldo + 4 * col) = o;
        } else {
            const float p0 = src[(size_t)id[0] * ldp + col], p1 = src[(size_t)id[1] * ldp + col], p2 = src[(size_t)id[2] * ldp + col];
            out_all[(size_t)row * ldo + col] = __builtin_fmaf(w2, p2, __builtin_fmaf(w0, p0, w1 * p1));
        }
    }
}

int interp_pm_dispatch(int b, int c, int m, int n, const float *points, int ldp, const int *idx, const float *weight,
                       float *out, int ldo, hipStream_t st) {
    const bool vec = (c % 4 == 0) && (ldp % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)points | (uintptr_t)out) % 16 == 0);
    const long total = (long)b * n * (vec ? c / 4 : c);
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (vec) hipLaunchKernelGGL(interp_pm_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, c, m, n, points, ldp, idx, weight, out, ldo, total);
    else hipLaunchKernelGGL(interp_pm_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, c, m, n, points, ldp, idx, weight, out, ldo, total);
    return check_launch();
}

}  // namespace pn2
